#!/bin/bash
# Runs on the GPU box: kernel statistics of the bf16 mode at configs[3] shapes (rocprofv3 --kernel-trace --stats).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03_cfg4_trace
rm -rf "$O" && mkdir -p "$O/trace"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace" -o t -- python3 "$R/bench.py" --no-cpu-baseline --workload cfg4 --compute bf16 --steps 10 > "$O/trace.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$O/trace.log"; exit 1; }
cp "$(ls "$O"/trace/*kernel_stats.csv "$O"/trace/*/*kernel_stats.csv 2>/dev/null | head -1)" "$O/kernel_stats.csv"
find "$O/trace" -name "*kernel_trace.csv" -delete
tail -2 "$O/trace.log" | cut -c1-300
python3 - "$O/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:32]:
    print("%-110s %6s calls %9.1f us avg %5.1f %%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
