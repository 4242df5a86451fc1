# A/B of one environment switch on the AMP lines: bash tools/r4_ab3.sh <out> <ENV_NAME>
set -e
OUT=gpurun_out/${1:-ab3}
VAR=${2:-TASEG_CLASS_H_SLICES}
mkdir -p $OUT
python -m pytest tests/test_gpu_conv_class.py tests/test_gpu_amp.py tests/test_gpu_class_model.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
for rep in 1 2; do
  for w in "--amp" "--workload nuscenes_ms --amp" "--workload minkunet_ms --amp"; do
    tag=$(echo $w | tr -d ' -')
    python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/new_${tag}_$rep.json 2> /dev/null
    env $VAR=1 python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/old_${tag}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
