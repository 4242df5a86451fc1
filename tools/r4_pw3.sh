set -e
OUT=$PWD/gpurun_out/${1:-pw3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TASEG_POINTWISE_BLOCK=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/new -o new -- python3 $GRAFT_REPO_ROOT/bench.py --eval --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/new.json 2>$OUT/err.txt
export TASEG_POINTWISE_BLOCK=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/old -o old -- python3 $GRAFT_REPO_ROOT/bench.py --eval --no-cpu-baseline --no-secondary --steps 60 --warmup 10 > $OUT/old.json 2>$OUT/err.txt
cd $OUT
for t in new old; do
  f=$(find $t -name "*kernel_stats.csv" | head -1)
  python3 - $f $t <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
print(sys.argv[2], "kernel ms total", round(tot / 1e6, 2), "launches", calls)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:14]:
    print("   ", r["Name"][:90], r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 2))
PY
  find $t -name "*.csv" ! -name "*kernel_stats.csv" -delete; find $t -name "*.db" -delete
done
