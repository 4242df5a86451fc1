"""Per-queue busy time of a rocprofv3 --kernel-trace run: how much device time does each HIP stream (HSA queue) carry per step?

    python tools/stream_busy.py <trace dir> <steps> [marker regex] [names]

(a fourth argument lists, per queue, the kernels behind its launches: launches and busy ms per step by kernel name)

Reads *_kernel_trace.csv.  The timed steps are the last <steps> occurrences of the marker kernel (default: sgd_decide_kernel, launched
once per training step; an evaluation loop has e.g. `argmax|ArgMax`); the window runs from the end of the marker before them to the end
of the last one.  Per queue: kernels per step, busy ms per step (sum of durations - what the queue would need with a host that is
never late), coverage of the window (under the profiler the host is slower than without it: low coverage here does not prove a
host-bound line, busy ms per step close to the UNPROFILED step time proves a device-bound one), then the union over all queues."""
import csv, glob, re, sys
from collections import defaultdict

d, steps = sys.argv[1], int(sys.argv[2])
marker = re.compile(sys.argv[3] if len(sys.argv) > 3 else "sgd_decide_kernel")
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [e for s, e, q, n in rows if marker.search(n)]
if len(marks) <= steps:
    raise SystemExit(f"only {len(marks)} marker launches for {steps} steps")
# markers may come several to a step (one per bucket): take every (len(marks) // total_steps)-th; the caller gives the TIMED steps
lo, hi = marks[-steps - 1], marks[-1]
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
window = (hi - lo) / 1e6
print(f"{steps} steps, window {window:.2f} ms = {window / steps:.3f} ms per step (under the profiler), {len(rows) / steps:.0f} launches per step")
print(f"LAUNCHES_PER_STEP {len(rows) / steps:.1f}")
qs, names = defaultdict(list), defaultdict(lambda: defaultdict(lambda: [0, 0]))
for s, e, q, n in rows:
    qs[q].append((s, e))
    short = re.sub(r"^void |rocprim::ROCPRIM_\d+_NS::detail::|\(.*$", "", n)[:70]
    names[q][short][0] += 1
    names[q][short][1] += e - s


def union(iv):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(iv):
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            tot += cur_e - cur_s
            cur_s, cur_e = s, e
    return tot + (cur_e - cur_s if cur_e is not None else 0)


for q, iv in sorted(qs.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e in iv)
    print(f"queue {q}: {len(iv) / steps:7.1f} kernels / step, busy {busy / 1e6 / steps:6.3f} ms / step, covers {100 * union(iv) / 1e6 / window:5.1f} % of the window")
allq = [x for iv in qs.values() for x in iv]
print(f"all queues: busy {sum(e - s for s, e in allq) / 1e6 / steps:.3f} ms / step, union {union(allq) / 1e6 / steps:.3f} ms / step "
      f"({100 * union(allq) / 1e6 / window:.1f} % of the window)")
if len(sys.argv) > 4:
    for q, iv in sorted(qs.items(), key=lambda kv: -len(kv[1])):
        print(f"queue {q}:")
        for n, (cnt, ns) in sorted(names[q].items(), key=lambda kv: -kv[1][0]):
            print(f"  {cnt / steps:7.1f} launches / step  {ns / 1e6 / steps:7.3f} ms / step  {n}")
