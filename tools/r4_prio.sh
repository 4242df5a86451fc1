set -e
OUT=gpurun_out/${1:-prio}
mkdir -p $OUT
export TASEG_WGRAD_STREAM=1
python -c "import torch; print(torch.cuda.Stream.priority_range())"
for rep in 1 2; do
  for p in none -1 0 1; do
    if [ $p = none ]; then unset TASEG_WGRAD_SIDE_PRIORITY; else export TASEG_WGRAD_SIDE_PRIORITY=$p; fi
    timeout -k 10 120 python bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 60 --warmup 10 > $OUT/p${p}_$rep.json 2> $OUT/p${p}_$rep.err || { tail -3 $OUT/p${p}_$rep.err; }
  done
done
python - <<'PY' $OUT
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(os.path.basename(f), "failed"); continue
    print(os.path.basename(f), round(d["ms_per_step"], 3), round(d["value"], 1))
PY
