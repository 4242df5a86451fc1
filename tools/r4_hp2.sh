set -e
OUT=gpurun_out/${1:-hp2}
mkdir -p $OUT
python -m pytest tests/test_gpu_model.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -60 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
export TASEG_BENCH_HOST_PHASES=1
for rep in 1 2 3; do
python bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/amp_$rep.json 2> $OUT/amp_$rep.err
done
python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/fp32_1.json 2> $OUT/fp32_1.err
grep -h "host issue\|second stream" $OUT/*.err
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
