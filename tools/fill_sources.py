"""Where do the runtime's fill / copy launches of a step come from?  For every `__amd_rocclr_fillBufferAligned` / `copyBuffer` of the
timed steps of a rocprofv3 --kernel-trace run: the queue it ran on and the kernels launched right before and after it on that queue,
counted by (previous, next) pair - the neighbours name the call site (a fill in front of `voxelize_kernel` is that entry point's
hipMemsetAsync, one in front of a `CatArrayBatchedCopy` is a torch.zeros ...).

    python tools/fill_sources.py <trace dir> <steps> [marker regex]
"""
import csv, glob, re, sys
from collections import Counter

d, steps = sys.argv[1], int(sys.argv[2])
marker = re.compile(sys.argv[3] if len(sys.argv) > 3 else "sgd_decide_kernel")
f = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
marks = [e for s, e, q, n in rows if marker.search(n)]
lo, hi = marks[-steps - 1], marks[-1]
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
short = lambda n: re.sub(r"^void |rocprim::ROCPRIM_\d+_NS::detail::|at::native::|\(anonymous namespace\)::|\(.*$|<.*$", "", n)[:48]   # noqa: E731
byq = {}
for r in rows:
    byq.setdefault(r[2], []).append(r)
pairs = Counter()
for q, rs in byq.items():
    for i, (s, e, _, n) in enumerate(rs):
        if "__amd_rocclr" in n:
            kind = "fill" if "fill" in n else "copy"
            prev = next((short(x[3]) for x in reversed(rs[:i]) if "__amd_rocclr" not in x[3]), "-")
            nxt = next((short(x[3]) for x in rs[i + 1:] if "__amd_rocclr" not in x[3]), "-")
            pairs[(kind, q, prev, nxt)] += 1
print(f"{sum(pairs.values()) / steps:.1f} runtime fills / copies per step")
for (kind, q, prev, nxt), c in pairs.most_common(60):
    print(f"{c / steps:6.1f} / step  {kind}  queue {q}:  after {prev:48s} before {nxt}")
