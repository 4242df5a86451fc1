set -e
OUT=gpurun_out/${1:-cprof3}
mkdir -p $OUT
export TASEG_BENCH_CPROFILE=$OUT/e.prof
python bench.py --eval --amp --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/e.json 2> $OUT/e.err
python - <<'PY' $OUT/e > $OUT/e_prof.txt
import pstats, sys
p = pstats.Stats(sys.argv[1] + ".prof")
p.sort_stats("tottime").print_stats(22)
PY
rm -f $OUT/e.prof
export TASEG_BENCH_CPROFILE=$OUT/a.prof
python bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --no-wgrad-tune --steps 40 --warmup 10 > $OUT/a.json 2> $OUT/a.err
python - <<'PY' $OUT/a > $OUT/a_prof.txt
import pstats, sys
p = pstats.Stats(sys.argv[1] + ".prof")
p.sort_stats("tottime").print_stats(22)
PY
rm -f $OUT/a.prof
