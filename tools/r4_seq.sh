set -e
OUT=$PWD/gpurun_out/${1:-seq}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-secondary --no-kernel-events --no-wgrad-tune --steps 2 --warmup 2 > $OUT/b.json 2>$OUT/err.txt
cd $OUT
f=$(find t -name "*kernel_trace.csv" | head -1)
python3 - $f <<'PY' > $OUT/seq.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = len(rows)
# last quarter ~ the last step
t0 = int(rows[0]["Start_Timestamp"])
for r in rows[-1500:]:
    print(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"][:100], r.get("Stream_Id", ""))
PY
rm -rf t
