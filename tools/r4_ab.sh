# Round-4 A/B on one box: the default bench line with the round's switches on / off.   bash tools/r4_ab.sh <out dir under gpurun_out/>
OUT=gpurun_out/${1:-ab}
mkdir -p $OUT
B="python bench.py --no-cpu-baseline --no-secondary --steps 30 --warmup 8"
run() { tag=$1; shift; env "$@" $B $EXTRA > $OUT/$tag.json 2> $OUT/$tag.err; python - "$OUT/$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    fam = {f["family"]: round(f["ms_per_step"], 3) for f in d.get("roofline", {}).get("families", [])}
    print(f"{sys.argv[2]:28s} {d['value']:8.2f} {d['unit']}  {d['ms_per_step']:7.3f} ms/step  {fam}", flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
}
EXTRA=""
run base_all_off TASEG_BN_ONE_LAUNCH=0 TASEG_DIRECT_CONV=0
run bn_one_launch TASEG_BN_ONE_LAUNCH=1 TASEG_DIRECT_CONV=0
run direct_conv TASEG_BN_ONE_LAUNCH=0 TASEG_DIRECT_CONV=1
run both_on TASEG_BN_ONE_LAUNCH=1 TASEG_DIRECT_CONV=1
run bn_rows_12k TASEG_BN_ONE_LAUNCH_ROWS=12288 TASEG_DIRECT_CONV=1
run bn_rows_4k TASEG_BN_ONE_LAUNCH_ROWS=4096 TASEG_DIRECT_CONV=1
run bn_stream_only TASEG_BN_COL_REGS=0 TASEG_DIRECT_CONV=1
EXTRA="--amp"
run amp_all_off TASEG_BN_ONE_LAUNCH=0 TASEG_DIRECT_CONV=0
run amp_bn TASEG_BN_ONE_LAUNCH=1 TASEG_DIRECT_CONV=0
run amp_both_on TASEG_BN_ONE_LAUNCH=1 TASEG_DIRECT_CONV=1
