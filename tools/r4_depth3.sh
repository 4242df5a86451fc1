set -e
OUT=gpurun_out/${1:-depth3}
mkdir -p $OUT
export TASEG_BENCH_HOST_PHASES=1
for rep in 1 2 3 4; do
  for d in 1 2; do
    TASEG_STAGE_DEPTH=$d timeout -k 10 120 python bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 80 --warmup 10 > $OUT/amp_d${d}_$rep.json 2> $OUT/amp_d${d}_$rep.err
  done
done
grep -h "host issue" $OUT/amp_d1_*.err | head -2; grep -h "host issue" $OUT/amp_d2_*.err | head -2
python - <<'PY' $OUT
import json, glob, sys, os
r = {}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = "_".join(os.path.basename(f).split("_")[:2])
    r.setdefault(k, []).append(round(d["ms_per_step"], 3))
for k, v in r.items():
    print(k, sorted(v))
PY
