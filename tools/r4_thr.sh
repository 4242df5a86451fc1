set -e
OUT=gpurun_out/${1:-thr}
mkdir -p $OUT
export TASEG_WGRAD_STREAM=1
for rep in 1 2; do
  for cfg in "48000 60000" "16000 60000" "48000 20000" "16000 20000" "30000 40000"; do
    set -- $cfg
    tag=a$1_b$2
    TASEG_CLASS_MIN_ROWS_96=$1 TASEG_CLASS_MIN_ROWS_128=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/${tag}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
