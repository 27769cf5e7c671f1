#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03_trace
rm -rf "$O" && mkdir -p "$O/trace"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d "$O/trace" -o t -- python3 "$R/bench.py" --no-cpu-baseline > "$O/trace.log" 2>&1 || { echo "kernel trace failed"; tail -5 "$O/trace.log"; exit 1; }
DB=$(ls "$O"/trace/*.db "$O"/trace/*/*.db 2>/dev/null | head -1)
python3 "$R/tools/prof_timeline.py" "$DB" 20 > "$O/step_timeline.txt" 2>&1
cp "$(ls "$O"/trace/*kernel_stats.csv "$O"/trace/*/*kernel_stats.csv 2>/dev/null | head -1)" "$O/kernel_stats.csv"
find "$O/trace" -name "*kernel_trace.csv" -delete; find "$O/trace" -name "*.db" -delete
tail -3 "$O/trace.log"
