# A/B: the finish inside the product (opt-in thresholds) with the weight gradients on the second stream in both arms
set -e
OUT=gpurun_out/${1:-ab6}
mkdir -p $OUT
export TASEG_WGRAD_STREAM=1
for rep in 1 2 3; do
  python bench.py --workload nuscenes_ms --amp --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/old_nusamp_$rep.json 2> /dev/null
  TASEG_CLASS_FINISH_ROWS_HALF=60000 python bench.py --workload nuscenes_ms --amp --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/new_nusamp_$rep.json 2> /dev/null
  python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/old_f32_$rep.json 2> /dev/null
  TASEG_CLASS_FINISH_ROWS=150000 python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/new_f32_$rep.json 2> /dev/null
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
