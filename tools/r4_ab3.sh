# A/B of one environment setting on the AMP lines: bash tools/r4_ab3.sh <out> <NAME=VALUE of the 'old' arm> [pytest: 1]
set -e
OUT=gpurun_out/${1:-ab3}
VAR=${2:-TASEG_CLASS_H_SLICES=1}
mkdir -p $OUT
if [ "${3:-0}" = "1" ]; then python -m pytest tests/test_gpu_conv_class.py tests/test_gpu_amp.py tests/test_gpu_class_model.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }; fi
for rep in 1 2; do
  for w in "--amp" "--workload nuscenes_ms --amp" "--workload minkunet_ms --amp"; do
    tag=$(echo $w | tr -d ' -')
    python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/new_${tag}_$rep.json 2> /dev/null
    env $VAR python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/old_${tag}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
