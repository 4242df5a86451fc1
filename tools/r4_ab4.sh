# A/B of one environment setting on the fp32 default line and the nuScenes autocast line: bash tools/r4_ab4.sh <out> <NAME=VALUE of the 'old' arm>
set -e
OUT=gpurun_out/${1:-ab4}
VAR=${2:-TASEG_CLASS_GEMM=0}
mkdir -p $OUT
for rep in 1 2 3; do
  for w in "" "--workload nuscenes_ms --amp"; do
    tag=f32$(echo $w | tr -d ' -')
    python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/new_${tag}_$rep.json 2> /dev/null
    env $VAR python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/old_${tag}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
