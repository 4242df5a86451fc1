"""Idle gaps of the launch queue in a rocprofv3 --kernel-trace run: where does the stream that carries the step wait, and for what?

    python tools/queue_gaps.py <trace dir> <steps> [marker regex] [top]

The timed steps are the last <steps> occurrences of the marker kernel (default sgd_decide_kernel).  The launch queue = the queue with the
most kernels.  Prints the gaps summed by (kernel before, kernel after), largest first, with how much of each gap another queue was
busy (a gap covered by another queue is a JOIN - the launch stream waits for that work; an uncovered gap is the HOST being late).
Under the profiler the host is 30-40 % slower than without it (every launch is intercepted): a line that is device-bound on its own
shows host gaps here, above all at the step boundary - read the covered gaps, not the uncovered ones."""
import csv, glob, re, sys
from collections import defaultdict

d, steps = sys.argv[1], int(sys.argv[2])
marker = re.compile(sys.argv[3] if len(sys.argv) > 3 else "sgd_decide_kernel")
top = int(sys.argv[4]) if len(sys.argv) > 4 else 25
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [e for s, e, q, n in rows if marker.search(n)]
lo, hi = marks[-steps - 1], marks[-1]
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
count = defaultdict(int)
for r in rows:
    count[r[2]] += 1
main = max(count, key=count.get)
short = lambda n: re.sub(r"^void |rocprim::ROCPRIM_\d+_NS::detail::|\(.*$", "", n)[:48]
mq = [r for r in rows if r[2] == main]
others = sorted((s, e) for s, e, q, n in rows if q != main)


def covered(a, b):
    tot = 0
    for s, e in others:
        if e <= a:
            continue
        if s >= b:
            break
        tot += min(e, b) - max(s, a)
    return min(tot, b - a)


gaps = defaultdict(lambda: [0, 0, 0])
total = cov_total = 0
for (s0, e0, _, n0), (s1, e1, _, n1) in zip(mq, mq[1:]):
    g = s1 - e0
    if g <= 0:
        continue
    c = covered(e0, s1) if g > 3000 else 0
    k = (short(n0), short(n1))
    gaps[k][0] += g
    gaps[k][1] += 1
    gaps[k][2] += c
    total += g
    cov_total += c
print(f"launch queue {main}: {len(mq) / steps:.0f} kernels / step, idle {total / 1e6 / steps:.3f} ms / step in gaps "
      f"({cov_total / 1e6 / steps:.3f} ms of it while another queue was busy, gaps over 3 us only)")
for (a, b), (g, n, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:top]:
    print(f"  {g / 1e3 / steps:8.1f} us / step  in {n / steps:6.1f} gaps / step (avg {g / n / 1e3:6.1f} us, {100 * c / max(g, 1):3.0f} % covered)  {a}  ->  {b}")
