# Round-4 diagnostic: per-layer table of the fp32 step, per-queue timelines of the fp32 and AMP steps.
set -e
OUT=gpurun_out/${1:-diag}
mkdir -p $OUT
export TMPDIR=/tmp
python tools/layer_table.py > $OUT/layer_table.txt 2> $OUT/layer_table.err
RP="rocprofv3 --kernel-trace --output-format csv"
$RP -d $OUT/tl_f32 -- python3 bench.py --no-cpu-baseline --no-secondary --no-kernel-events --steps 6 --warmup 4 > $OUT/tl_f32.json 2> $OUT/tl_f32.err
python tools/timeline.py $OUT/tl_f32 60 > $OUT/timeline_f32.txt
$RP -d $OUT/tl_amp -- python3 bench.py --amp --no-cpu-baseline --no-secondary --no-kernel-events --steps 6 --warmup 4 > $OUT/tl_amp.json 2> $OUT/tl_amp.err
python tools/timeline.py $OUT/tl_amp 40 > $OUT/timeline_amp.txt
python - <<'PY' $OUT
import csv, glob, sys, collections
out = sys.argv[1]
for tag, win in (("f32", 47e6), ("amp", 30e6)):
    f = sorted(glob.glob(f"{out}/tl_{tag}/**/*kernel_trace.csv", recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey], r["Kernel_Name"]) for r in rows)
    t_end = ev[-1][1]
    ev = [e for e in ev if e[0] >= t_end - win]
    byq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
    for s, e, q, n in ev:
        byq[q][n[:90]][0] += 1
        byq[q][n[:90]][1] += e - s
    with open(f"{out}/queues_{tag}.txt", "w") as fh:
        for q, tab in byq.items():
            fh.write(f"queue {q}: {sum(v[1] for v in tab.values()) / 1e6:.2f} ms in {sum(v[0] for v in tab.values())} launches over the last {win / 1e6:.0f} ms\n")
            for n, (c, t) in sorted(tab.items(), key=lambda kv: -kv[1][1])[:40]:
                fh.write(f"   {t / 1e6:7.3f} ms {c:5d}  {n}\n")
PY
rm -rf $OUT/tl_f32 $OUT/tl_amp
