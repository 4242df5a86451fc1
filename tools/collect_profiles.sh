# Round profile set (run on the GPU box through gpurun): default bench line, rocprofv3 kernel traces of the default, --amp, nuScenes --amp and
# evaluation runs (kernel statistics + per-queue busy time + launches per step), the two PMC passes behind profiles/traffic.json.
#    bash tools/collect_profiles.sh <out dir under gpurun_out/> [main|side|all]   (writes progress lines: a silent run is taken to be hung;
#    "main" = default line + traces + PMC passes, "side" = the other lines by hand + host phases: each fits one 20-minute gpurun call)
set -e
OUT=gpurun_out/${1:-prof}
PART=${2:-all}
mkdir -p $OUT
export TMPDIR=/tmp
say() { echo "[collect $(date +%H:%M:%S)] $*"; }
if [ "$PART" != side ]; then
say "default bench line"
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
RP="rocprofv3 --kernel-trace --stats --output-format csv"
echo "{" > $OUT/launches.json
# one traced run: kernel statistics, per-queue busy time inside the timed steps, launches per step
trace() {   # tag, launches key, marker regex, steps, bench arguments...
  tag=$1; key=$2; marker=$3; steps=$4; shift 4
  say "trace $tag"
  $RP -d $OUT/trace_$tag -- python3 bench.py "$@" --no-cpu-baseline --no-secondary > $OUT/bench_${tag}_under_rocprof.json 2> $OUT/trace_$tag.err
  python tools/kstats.py $OUT/trace_$tag $steps 45 > $OUT/kstats_$tag.txt
  python tools/stream_busy.py $OUT/trace_$tag $steps "$marker" > $OUT/busy_$tag.txt
  python tools/fill_sources.py $OUT/trace_$tag $steps "$marker" > $OUT/fills_$tag.txt 2>&1 || true
  echo "  \"$key\": $(grep LAUNCHES_PER_STEP $OUT/busy_$tag.txt | cut -d' ' -f2)," >> $OUT/launches.json
  f=$(find $OUT/trace_$tag -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${tag}_kernel_stats.csv
  rm -rf $OUT/trace_$tag
}
# every kernel alone on the device (weight gradients on the caller's stream: what the roofline figures of the bench line are measured on),
# then the same command with them on the second stream (what the timed steps run where that is faster)
export TASEG_WGRAD_STREAM=0
trace default "minkunet" sgd_decide_kernel 20 --steps 20 --warmup 5
trace amp "minkunet amp" sgd_decide_kernel 20 --amp --steps 20 --warmup 5
trace nus_amp "nuscenes_ms amp" sgd_decide_kernel 20 --workload nuscenes_ms --amp --steps 20 --warmup 5
trace ms "minkunet_ms" sgd_decide_kernel 20 --workload minkunet_ms --steps 20 --warmup 5
export TASEG_WGRAD_STREAM=1
trace default_side "minkunet second-stream" sgd_decide_kernel 20 --steps 20 --warmup 5
unset TASEG_WGRAD_STREAM
unset TASEG_WGRAD_STREAM
# the reference's recipe batch sizes, mask distillation and TIAF (round 6)
trace bs12_amp "minkunet amp bs12" sgd_decide_kernel 12 --batch 12 --amp --steps 12 --warmup 4
trace fsa16_bs6_amp "minkunet_ms amp bs6 history16" sgd_decide_kernel 10 --workload minkunet_ms --history 16 --batch 6 --amp --steps 10 --warmup 3
trace kd "kd" sgd_decide_kernel 10 --workload kd --steps 10 --warmup 3
python bench.py --workload tiaf --amp --steps 2 --warmup 1 > /dev/null 2>&1      # (MIOpen's kernels of this process tree compiled before the trace)
trace tiaf_amp "tiaf amp" sgd_decide_kernel 6 --workload tiaf --amp --steps 6 --warmup 2 --no-wgrad-tune
python tools/tiaf_families.py $OUT/tiaf_amp_kernel_stats.csv 13 > $OUT/tiaf_amp_families.txt     # 2 + 6 timed + 5 phase-table steps in the trace
trace eval "eval minkunet" "unvoxelise_kernel" 40 --eval --steps 40 --warmup 8
trace evalamp "eval minkunet amp" "unvoxelise_kernel" 40 --eval --amp --steps 40 --warmup 8
echo "  \"_source\": \"rocprofv3 --kernel-trace of the tree, kernels of all streams inside the timed steps (tools/stream_busy.py)\"" >> $OUT/launches.json
echo "}" >> $OUT/launches.json
export TASEG_WGRAD_STREAM=0
say "PMC passes"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 2 --warmup 1 > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 2 --warmup 1 > $OUT/pmc_write.json 2> $OUT/pmc_write.err
say "PMC passes (amp)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_amp -o f -- python3 bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 2 --warmup 1 > $OUT/pmc_fetch_amp.json 2> $OUT/pmc_fetch_amp.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_amp -o w -- python3 bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 2 --warmup 1 > $OUT/pmc_write_amp.json 2> $OUT/pmc_write_amp.err
unset TASEG_WGRAD_STREAM
python profiles/parse_traffic.py $(find $OUT/pmc_fetch -name "*counter_collection.csv") $(find $OUT/pmc_write -name "*counter_collection.csv") $(find $OUT/pmc_fetch_amp -name "*counter_collection.csv") $(find $OUT/pmc_write_amp -name "*counter_collection.csv") > $OUT/traffic.txt
cp profiles/traffic.json $OUT/traffic.json
rm -rf $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_fetch_amp $OUT/pmc_write_amp
fi
if [ "$PART" = main ]; then exit 0; fi
say "side lines"
for w in "--eval" "--eval --amp" "--workload minkunet_ms" "--amp" "--workload nuscenes_ms --amp" "--workload nuscenes_ms" "--batch 8" "--batch 8 --amp" "--force-dist" "--workload kd" "--workload tiaf --amp" "--workload tiaf" "--batch 12 --amp" "--workload minkunet_ms --history 16 --batch 6 --amp"; do
  tag=$(echo $w | tr -d ' -'); say "bench $w"; python bench.py --steps 30 --warmup 5 $w --no-cpu-baseline --no-secondary > $OUT/bench_$tag.json 2> /dev/null; done
say "2-D convolution kernels: probe, per-layer table, SQ counters"
python tools/conv2d_probe.py > $OUT/conv2d_probe.txt 2> /dev/null
python tools/unet2d_layers.py > $OUT/unet2d_layers.txt 2> /dev/null || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_conv2d -o p -- python3 tools/conv2d_probe.py general > /dev/null 2> $OUT/pmc_conv2d.err
python tools/pmc_conv2d.py $(find $OUT/pmc_conv2d -name "*counter_collection.csv" | head -1) > $OUT/conv3x3_rows_pmc.txt
rm -rf $OUT/pmc_conv2d
say "dense-loss probe"
python tools/loss_probe.py > $OUT/loss_probe.txt 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_loss -- python3 tools/loss_probe.py > /dev/null 2> $OUT/trace_loss.err
python tools/kstats.py $OUT/trace_loss 7 12 >> $OUT/loss_probe.txt
rm -rf $OUT/trace_loss
say "host phases"
for w in "" "--amp"; do TASEG_BENCH_HOST_PHASES=1 python bench.py --steps 30 --warmup 5 $w --no-cpu-baseline --no-secondary 2>&1 > /dev/null | grep -E "host issue|native nodes|second stream" | sed "s/^/[bench.py $w] /" >> $OUT/host_phases.txt; done
for t in 0 1; do TASEG_STAGE_THREAD=$t TASEG_BENCH_HOST_PHASES=1 python bench.py --amp --steps 30 --warmup 5 --no-cpu-baseline --no-secondary 2>&1 > /dev/null | grep -E "host issue|native nodes" | sed "s/^/[bench.py --amp, TASEG_STAGE_THREAD=$t] /" >> $OUT/host_phases.txt; done
tail -c 400 $OUT/bench_default.json
