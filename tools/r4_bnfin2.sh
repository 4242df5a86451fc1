set -e
OUT=gpurun_out/${1:-bnfin2}
mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_amp.py -x -q -m gpu -k "bn or batch or norm or amp" > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for rep in 1 2; do
  for mode in 1 0; do
    export TASEG_BN_FINISH_IN_PARTIAL=$mode
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/fp32_m${mode}_$rep.json 2> /dev/null
    timeout -k 10 120 python bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/amp_m${mode}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1), d.get("loss"))
PY
