set -e
OUT=gpurun_out/${1:-ws}
mkdir -p $OUT
for rep in 1 2 3; do
  for w in "--amp" "--batch 8 --amp" "--workload minkunet_ms --amp" "--workload nuscenes_ms"; do
    tag=$(echo $w | tr -d ' -')
    TASEG_WGRAD_STREAM=1 python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/new_${tag}_$rep.json 2> $OUT/new_${tag}_$rep.err
    TASEG_WGRAD_STREAM=0 python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/old_${tag}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
