set -e
OUT=gpurun_out/${1:-ev3}
mkdir -p $OUT
python -m pytest tests/test_gpu_model.py tests/test_gpu_parity_r2.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -60 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for rep in 1 2 3; do
  for mode in new old; do
    if [ $mode = old ]; then export TASEG_POINTWISE_BLOCK=0; else export TASEG_POINTWISE_BLOCK=1; fi
    python bench.py --eval --no-cpu-baseline --no-secondary --steps 80 --warmup 10 > $OUT/${mode}_eval_$rep.json 2> /dev/null
    python bench.py --eval --amp --no-cpu-baseline --no-secondary --steps 80 --warmup 10 > $OUT/${mode}_evalamp_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
