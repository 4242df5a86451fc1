"""Diagnostic: per-queue busy time and idle gaps from a rocprofv3 --kernel-trace CSV (last N steps of bench.py)."""
import csv, glob, sys, collections
d = sys.argv[1]
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
qkey = "Queue_Id" if "Queue_Id" in rows[0] else "Stream_Id"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qkey], r["Kernel_Name"]) for r in rows]
ev.sort()
t_end = ev[-1][1]
window = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 200e6          # analyse the last <ms> of the trace
ev = [e for e in ev if e[0] >= t_end - window]
span = (ev[-1][1] - ev[0][0]) / 1e6
print(f"window {span:.1f} ms, {len(ev)} kernels")
byq = collections.defaultdict(list)
for e in ev:
    byq[e[2]].append(e)
for q, es in byq.items():
    busy = sum(e[1] - e[0] for e in es) / 1e6
    gaps = [(es[i + 1][0] - es[i][1]) / 1e3 for i in range(len(es) - 1)]
    big = sorted(((g, es[i][3][:50], es[i + 1][3][:50]) for i, g in enumerate(gaps) if g > 20), reverse=True)[:12]
    print(f"queue {q}: {len(es)} kernels, busy {busy:.1f} ms ({100 * busy / span:.0f} %), gaps>5us: "
          f"{sum(1 for g in gaps if g > 5)}, sum of gaps {sum(g for g in gaps if g > 0) / 1e3:.1f} ms")
    for g, a, b in big:
        print(f"     {g:8.1f} us between [{a}] and [{b}]")
# union busy over all queues
merged, cur_s, cur_e = 0, None, None
for s, e, _, _ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            merged += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
merged += cur_e - cur_s
print(f"device busy (union of queues) {merged / 1e6:.1f} ms = {100 * merged / 1e6 / span:.0f} % of the window")
