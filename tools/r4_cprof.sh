set -e
OUT=gpurun_out/${1:-cprof}
mkdir -p $OUT
export TASEG_BENCH_CPROFILE=$OUT/amp.prof
python bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --no-wgrad-tune --steps 40 --warmup 10 > $OUT/amp.json 2> $OUT/amp.err
python - <<'PY' $OUT > $OUT/amp_prof.txt
import pstats, sys
p = pstats.Stats(sys.argv[1] + "/amp.prof")
p.sort_stats("cumulative").print_stats(70)
p.sort_stats("tottime").print_stats(45)
PY
rm -f $OUT/amp.prof
