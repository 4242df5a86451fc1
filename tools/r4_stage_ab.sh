set -e
OUT=gpurun_out/${1:-stab}
mkdir -p $OUT
export TASEG_WGRAD_STREAM=1
for rep in 1 2; do
  python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/base_$rep.json 2> /dev/null
  TASEG_STAGE_THREAD=1 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/thread_early_$rep.json 2> /dev/null
  TASEG_STAGE_THREAD=1 TASEG_STAGE_EARLY=0 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/thread_late_$rep.json 2> /dev/null
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
