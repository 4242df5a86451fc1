set -e
OUT=gpurun_out/${1:-pw}
mkdir -p $OUT
python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "pointwise or residual_block or eval_block_call or conv_block_paths or second_stream" > $OUT/pytest1.txt 2>&1 || { tail -60 $OUT/pytest1.txt; exit 1; }
tail -2 $OUT/pytest1.txt
export TASEG_BENCH_HOST_PHASES=1
for rep in 1 2; do
python bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/amp_$rep.json 2> $OUT/amp_$rep.err
python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/fp32_$rep.json 2> $OUT/fp32_$rep.err
done
python bench.py --eval --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/eval.json 2> /dev/null
python bench.py --eval --amp --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/evalamp.json 2> /dev/null
grep -h "host issue\|second stream" $OUT/*.err
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
