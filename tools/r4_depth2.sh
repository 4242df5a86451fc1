set -e
OUT=gpurun_out/${1:-depth2}
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for rep in 1 2; do
  timeout -k 10 120 python bench.py --eval --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > $OUT/eval_$rep.json 2> /dev/null
  timeout -k 10 120 python bench.py --eval --amp --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > $OUT/evalamp_$rep.json 2> /dev/null
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
