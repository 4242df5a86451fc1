# Round profile set (run on the GPU box through gpurun): default bench line, rocprofv3 kernel traces of the default, --amp and
# nuScenes --amp runs, the two PMC passes behind profiles/traffic.json.   bash tools/collect_profiles.sh <out dir under gpurun_out/>
set -e
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
export TMPDIR=/tmp
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
RP="rocprofv3 --kernel-trace --stats --output-format csv"
# kernel statistics with every kernel alone on the device (weight gradients on the caller's stream: what the roofline figures of the
# bench line are measured on), then the same command with them on the second stream (what the timed steps run where that is faster)
export TASEG_WGRAD_STREAM=0
$RP -d $OUT/trace_default -- python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench_under_rocprof.json 2> $OUT/trace_default.err
$RP -d $OUT/trace_amp -- python3 bench.py --amp --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench_amp_under_rocprof.json 2> $OUT/trace_amp.err
$RP -d $OUT/trace_nus_amp -- python3 bench.py --workload nuscenes_ms --amp --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench_nus_amp_under_rocprof.json 2> $OUT/trace_nus_amp.err
export TASEG_WGRAD_STREAM=1
$RP -d $OUT/trace_default_side -- python3 bench.py --no-cpu-baseline --no-secondary --steps 20 --warmup 5 > $OUT/bench_side_under_rocprof.json 2> $OUT/trace_default_side.err
unset TASEG_WGRAD_STREAM
export TASEG_WGRAD_STREAM=0
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 2 --warmup 1 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 2 --warmup 1 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_amp -o f -- python3 bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 2 --warmup 1 > $OUT/pmc_fetch_amp.json 2> $OUT/pmc_fetch_amp.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_amp -o w -- python3 bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 2 --warmup 1 > $OUT/pmc_write_amp.json 2> $OUT/pmc_write_amp.err
unset TASEG_WGRAD_STREAM
# keep the summaries (kernel_stats.csv per run), drop the per-dispatch traces: gpurun copies back at most 64 MiB
keep_stats() { f=$(find $OUT/$1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/$2_kernel_stats.csv; rm -rf $OUT/$1; }
for t in default amp nus_amp default_side; do python tools/kstats.py $OUT/trace_$t 25 45 > $OUT/kstats_$t.txt; done
keep_stats trace_default bench; keep_stats trace_amp bench_amp; keep_stats trace_nus_amp bench_nuscenes_amp; keep_stats trace_default_side bench_side
python profiles/parse_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $(find $OUT/pmc_fetch_amp -name "*counter_collection.csv") $(find $OUT/pmc_write_amp -name "*counter_collection.csv") > $OUT/traffic.txt
cp profiles/traffic.json $OUT/traffic.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch_amp $OUT/pmc_write_amp
# the data stage alone under the profiler: launches and GPU time per batch = totals / 23 batches (3 warm-up + 20 timed) per form
for wl in minkunet_ms nuscenes_ms; do for form in batched per_sample; do
  $RP -d $OUT/trace_stage_${wl}_$form -- python3 tools/stage_probe.py --workload $wl --only $form --reps 20 > $OUT/stage_${wl}_$form.txt 2> /dev/null
  python tools/kstats.py $OUT/trace_stage_${wl}_$form 23 12 > $OUT/kstats_stage_${wl}_$form.txt
  keep_stats trace_stage_${wl}_$form stage_${wl}_$form; done; done
python tools/stage_probe.py > $OUT/stage_probe.txt 2> /dev/null
for w in "--eval" "--eval --amp"; do tag=$(echo $w | tr -d ' -'); python bench.py $w --no-cpu-baseline --no-secondary --steps 40 --warmup 8 > $OUT/bench_$tag.json 2> /dev/null; done
python tools/eval_probe.py > $OUT/eval_probe.txt 2> /dev/null; python tools/eval_probe.py --amp >> $OUT/eval_probe.txt 2> /dev/null
for w in "--workload minkunet_ms" "--amp" "--workload nuscenes_ms --amp" "--workload nuscenes_ms" "--batch 8" "--batch 8 --amp" "--force-dist"; do
  tag=$(echo $w | tr -d ' -'); python bench.py $w --no-cpu-baseline --no-secondary --steps 30 --warmup 5 > $OUT/bench_$tag.json 2> /dev/null; done
tail -c 400 $OUT/bench_default.json
