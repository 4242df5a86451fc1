# A/B of the three-product class kernels on the fp32 lines: new = default, old = TASEG_CLASS_X=0; weight gradients pinned to the second stream
set -e
OUT=gpurun_out/${1:-ab5}
mkdir -p $OUT
python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py tests/test_gpu_class_model.py tests/test_gpu_bench_size.py -q -m gpu > $OUT/pytest.txt 2>&1 || true
export TASEG_WGRAD_STREAM=1
for rep in 1 2 3; do
  for w in "" "--workload minkunet_ms" "--workload nuscenes_ms"; do
    tag=f32$(echo $w | tr -d ' -')
    python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/new_${tag}_$rep.json 2> /dev/null
    TASEG_CLASS_X=0 python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 8 > $OUT/old_${tag}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1), d["loss"])
PY
