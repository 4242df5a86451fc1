# Round-4 A/B, second set (one box): direct 2x2x2 plans and the batched data stage on / off, four workloads, two rounds.
OUT=gpurun_out/${1:-ab2}
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-secondary --steps 40 --warmup 8"
for rep in 1 2; do
for w in "" "--amp" "--workload minkunet_ms" "--workload nuscenes_ms --amp"; do
  for sw in "TASEG_DIRECT_CONV=1 TASEG_STAGE_BATCHED=1" "TASEG_DIRECT_CONV=0 TASEG_STAGE_BATCHED=1" "TASEG_DIRECT_CONV=1 TASEG_STAGE_BATCHED=0"; do
    env $sw $B $w 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('round $rep | $sw | bench.py $w |', round(d['value'],2), 'scans/s', round(d['ms_per_step'],3), 'ms')"
  done
done
done | tee $OUT/ab2.txt
