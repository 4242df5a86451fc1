set -e
OUT=gpurun_out/${1:-order}
mkdir -p $OUT
for o in "" "--ms" "--morton 1" "--morton 3" "--morton 5"; do
  for h in "" "--half"; do
    echo "== order [$o] $h" >> $OUT/order.txt
    python tools/class_probe.py --stride 1 --cin 96 --cout 96 $o $h >> $OUT/order.txt 2>> $OUT/order.err
    python tools/class_probe.py --stride 2 --cin 96 --cout 96 $o $h >> $OUT/order.txt 2>> $OUT/order.err
  done
done
