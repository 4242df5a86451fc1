set -e
OUT=gpurun_out/${1:-evt}
mkdir -p $OUT
python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
for rep in 1 2 3; do
  for w in "--eval" "--eval --amp"; do
    tag=$(echo $w | tr -d ' -')
    python bench.py $w --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/new_${tag}_$rep.json 2> /dev/null
    TASEG_EVAL_TAIL_IN_PASS2=0 python bench.py $w --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/old_${tag}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
