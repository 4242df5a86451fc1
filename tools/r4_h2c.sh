set -e
OUT=gpurun_out/${1:-h2}
mkdir -p $OUT
for d in 0 1 2 4 3 7 5; do
  echo "== dbg $d" >> $OUT/probe.txt
  TASEG_DBG_H2=$d python tools/class_probe.py --stride 1 --cin 96 --cout 96 --half >> $OUT/probe.txt 2>> $OUT/probe.err
done
