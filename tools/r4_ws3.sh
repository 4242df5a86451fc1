set -e
OUT=gpurun_out/${1:-ws}
mkdir -p $OUT
python -m pytest tests/test_gpu_model.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
for w in "" "--amp" "--workload nuscenes_ms --amp" "--workload minkunet_ms" "--workload minkunet_ms --amp" "--batch 8 --amp"; do
  tag=f32$(echo $w | tr -d ' -')
  python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/auto_${tag}.json 2> $OUT/auto_${tag}.err
  grep "second stream" $OUT/auto_${tag}.err || true
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1), d["config"].get("wgrad_on_second_stream"))
PY
