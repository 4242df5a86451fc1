set -e
OUT=gpurun_out/${1:-pamp}
mkdir -p $OUT
export TMPDIR=/tmp
RP="rocprofv3 --kernel-trace --stats --output-format csv"
$RP -d $OUT/trace_amp -- python3 bench.py --amp --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench_amp_under_rocprof.json 2> $OUT/trace_amp.err
$RP -d $OUT/trace_nus_amp -- python3 bench.py --workload nuscenes_ms --amp --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench_nus_amp_under_rocprof.json 2> $OUT/trace_nus_amp.err
keep_stats() { f=$(find $OUT/$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$2_kernel_stats.csv; rm -rf $OUT/$1; }
keep_stats trace_amp bench_amp; keep_stats trace_nus_amp bench_nuscenes_amp
