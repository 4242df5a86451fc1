set -e
OUT=gpurun_out/${1:-chunk}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bench_size.py tests/test_gpu_parity_r2.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -30 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for rep in 1 2 3; do
  TASEG_WGRAD_STREAM=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/side_$rep.json 2> /dev/null
  TASEG_WGRAD_STREAM=0 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/one_$rep.json 2> /dev/null
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1), d["loss"])
PY
