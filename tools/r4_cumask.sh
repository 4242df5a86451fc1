set -e
OUT=gpurun_out/${1:-cumask}
mkdir -p $OUT
export TASEG_WGRAD_STREAM=1
for rep in 1 2; do
  for c in 0 64 96 128 160 192; do
    TASEG_WGRAD_SIDE_CUS=$c timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/c${c}_$rep.json 2> $OUT/c${c}_$rep.err || { tail -5 $OUT/c${c}_$rep.err; exit 1; }
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1), d["loss"])
PY
