# soak: 300-step runs with the weight gradients on one stream and on the second stream - the final loss must be the same bits
set -e
OUT=gpurun_out/${1:-soak}
mkdir -p $OUT
for w in "" "--amp" "--workload nuscenes_ms --amp"; do
  tag=f32$(echo $w | tr -d ' -')
  for m in 0 1; do
    TASEG_WGRAD_STREAM=$m python bench.py $w --no-cpu-baseline --no-secondary --no-kernel-events --steps 300 --warmup 8 > $OUT/soak_${tag}_$m.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/soak_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1), repr(d["loss"]))
PY
