set -e
OUT=gpurun_out/${1:-evx}
mkdir -p $OUT
for rep in 1 2 3; do
  python bench.py --eval --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/old_eval_$rep.json 2> /dev/null
  TASEG_CLASS_X=1 python bench.py --eval --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/new_eval_$rep.json 2> /dev/null
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
