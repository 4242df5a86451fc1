set -e
OUT=gpurun_out/${1:-ws}
mkdir -p $OUT
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "second_stream" > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
for rep in 1 2; do
  for w in "" "--amp" "--workload nuscenes_ms --amp" "--workload minkunet_ms --amp"; do
    tag=f32$(echo $w | tr -d ' -')
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
