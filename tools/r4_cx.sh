set -e
OUT=gpurun_out/${1:-cx}
mkdir -p $OUT
for cfg in "1 96 96" "1 128 96" "2 96 96" "1 96 128"; do
  set -- $cfg
  python tools/experiments/class_x_probe.py --stride $1 --cin $2 --cout $3 >> $OUT/probe.txt 2>> $OUT/probe.err
done
python tools/experiments/class_x_probe.py --stride 1 --cin 96 --cout 96 --scale 1e-4 >> $OUT/probe.txt 2>> $OUT/probe.err
