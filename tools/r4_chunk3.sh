set -e
OUT=gpurun_out/${1:-chunk4}
mkdir -p $OUT
for rep in 1 2 3 4 5 6; do
  for cap in 0 1024; do
    TASEG_WGRAD_MAXCHUNK=$cap TASEG_WGRAD_STREAM=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/side_c${cap}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
r = {}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r.setdefault(os.path.basename(f).split("_")[1], []).append(round(d["ms_per_step"], 3))
for k, v in r.items():
    print(k, sorted(v))
PY
