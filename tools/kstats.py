"""Summarise a rocprofv3 --kernel-trace --stats run: per-kernel ms/step from *_kernel_stats.csv."""
import csv, glob, sys
d, steps = sys.argv[1], float(sys.argv[2])
f = sorted(glob.glob(d + "/**/*kernel_stats.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms/step %.2f   launches/step %.0f" % (tot / 1e6 / steps, sum(int(r["Calls"]) for r in rows) / steps))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%7.3f ms/step  %6.1f calls/step  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, int(r["Calls"]) / steps,
          float(r["AverageNs"]) / 1e3, r["Name"][:110]))
