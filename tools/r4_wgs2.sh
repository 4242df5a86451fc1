set -e
OUT=gpurun_out/${1:-wgs3}
mkdir -p $OUT
for rep in 1 2; do
  for s in 0 1; do
  for w in 256 320 384 512; do
    TASEG_WGRAD_STREAM=$s TASEG_WGRAD_WGS=$w python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/s${s}_w${w}_$rep.json 2> /dev/null
  done; done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
