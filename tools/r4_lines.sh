set -e
OUT=gpurun_out/${1:-lines}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_model.py tests/test_gpu_class_model.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
export TASEG_BENCH_HOST_PHASES=1
for rep in 1 2 3; do
  timeout -k 10 120 python bench.py --eval --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > $OUT/eval_$rep.json 2> /dev/null
  timeout -k 10 120 python bench.py --eval --amp --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > $OUT/evalamp_$rep.json 2> /dev/null
  timeout -k 10 120 python bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 80 --warmup 10 > $OUT/amp_$rep.json 2> $OUT/amp_$rep.err
done
timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/fp32_1.json 2> $OUT/fp32_1.err
grep -h "host issue" $OUT/*.err
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
