set -e
OUT=gpurun_out/${1:-xd}
mkdir -p $OUT
timeout -k 10 500 python -m pytest tests/test_gpu_class_model.py tests/test_gpu_model.py tests/test_gpu_conv_class.py -x -q -m gpu > $OUT/pytest.txt 2>&1 || { tail -40 $OUT/pytest.txt; exit 1; }
tail -1 $OUT/pytest.txt
for rep in 1 2 3; do
  for mode in dgrad 0; do
    TASEG_CLASS_X=$mode TASEG_WGRAD_STREAM=1 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/side_${mode}_$rep.json 2> /dev/null
    TASEG_CLASS_X=$mode TASEG_WGRAD_STREAM=0 timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/one_${mode}_$rep.json 2> /dev/null
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1), d["loss"])
PY
