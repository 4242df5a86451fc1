set -e
OUT=gpurun_out/${1:-depth}
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "prefetch" > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for rep in 1 2 3 4; do
  for d in 1 2; do
    export TASEG_EVAL_STAGE_DEPTH=$d
    timeout -k 10 120 python bench.py --eval --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > $OUT/eval_d${d}_$rep.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
    timeout -k 10 120 python bench.py --eval --amp --no-cpu-baseline --no-secondary --steps 100 --warmup 10 > $OUT/evalamp_d${d}_$rep.json 2> $OUT/err.txt || { tail -5 $OUT/err.txt; exit 1; }
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
r = {}
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    k = "_".join(os.path.basename(f).split("_")[:2])
    r.setdefault(k, []).append(round(d["ms_per_step"], 3))
for k, v in r.items():
    print(k, sorted(v))
PY
