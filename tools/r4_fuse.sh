set -e
OUT=gpurun_out/${1:-fuse}
mkdir -p $OUT
python -m pytest tests/test_gpu_conv_class.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
for h in "" "--half"; do
for cfg in "1 96 96" "1 128 96" "1 32 32" "2 96 96" "2 64 64" "2 32 32" "4 64 64" "4 128 128"; do
  set -- $cfg
  python tools/class_probe.py --stride $1 --cin $2 --cout $3 $h >> $OUT/probe.txt 2>> $OUT/probe.err
done; done
