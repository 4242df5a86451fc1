set -e
OUT=gpurun_out/${1:-h2}
mkdir -p $OUT
for gg in 0 2 3 4; do
for cfg in "1 96 96" "1 32 32" "2 96 96" "2 64 64" "4 128 128" "8 128 128"; do
  set -- $cfg
  echo "== G $gg" >> $OUT/probe.txt
  TASEG_CLASS_H_G=$gg python tools/class_probe.py --stride $1 --cin $2 --cout $3 --half >> $OUT/probe.txt 2>> $OUT/probe.err
done; done
