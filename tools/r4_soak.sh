set -e
OUT=gpurun_out/${1:-soak2}
mkdir -p $OUT
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary --steps 300 --warmup 10 > $OUT/fp32.json 2> $OUT/fp32.err
timeout -k 10 200 python bench.py --amp --no-cpu-baseline --no-secondary --steps 300 --warmup 10 > $OUT/amp.json 2> $OUT/amp.err
timeout -k 10 200 python bench.py --workload nuscenes_ms --amp --no-cpu-baseline --no-secondary --steps 150 --warmup 10 > $OUT/nusc.json 2> $OUT/nusc.err
timeout -k 10 200 python bench.py --eval --no-cpu-baseline --no-secondary --steps 300 --warmup 10 > $OUT/eval.json 2> $OUT/eval.err
timeout -k 10 200 python bench.py --eval --amp --no-cpu-baseline --no-secondary --steps 300 --warmup 10 > $OUT/evalamp.json 2> $OUT/evalamp.err
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), d["steps"], "steps", round(d["ms_per_step"], 3), "ms", round(d["value"], 1), "scans/s loss", d.get("loss"), "side", (d.get("config") or {}).get("wgrad_on_second_stream"))
PY
