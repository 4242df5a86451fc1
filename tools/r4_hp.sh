set -e
OUT=gpurun_out/${1:-hp}
mkdir -p $OUT
export TASEG_BENCH_HOST_PHASES=1
python bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/amp.json 2> $OUT/amp.err
python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/fp32.json 2> $OUT/fp32.err
python bench.py --amp --workload nuscenes_ms --no-cpu-baseline --no-secondary --no-kernel-events --steps 40 --warmup 10 > $OUT/nusc.json 2> $OUT/nusc.err
grep -h "host issue\|second stream" $OUT/*.err
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
