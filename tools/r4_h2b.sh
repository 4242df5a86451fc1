set -e
OUT=gpurun_out/${1:-h2}
mkdir -p $OUT
for cfg in "1 96 96" "2 96 96" "1 32 32" "2 64 64" "4 64 64"; do
  set -- $cfg
  python tools/class_probe.py --stride $1 --cin $2 --cout $3 --half >> $OUT/probe.txt 2>> $OUT/probe.err
done
