set -e
OUT=gpurun_out/${1:-cprof2}
mkdir -p $OUT
for mode in new old; do
  if [ $mode = old ]; then export TASEG_POINTWISE_BLOCK=0; else export TASEG_POINTWISE_BLOCK=1; fi
  export TASEG_BENCH_CPROFILE=$OUT/$mode.prof
  python bench.py --eval --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/$mode.json 2> $OUT/$mode.err
  python - <<'PY' $OUT/$mode > $OUT/${mode}_prof.txt
import pstats, sys
p = pstats.Stats(sys.argv[1] + ".prof")
p.sort_stats("cumulative").print_stats(45)
p.sort_stats("tottime").print_stats(25)
PY
  rm -f $OUT/$mode.prof
done
