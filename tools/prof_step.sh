#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + stats of the default bench command, then the
# timeline of one replayed step.  usage: tools/prof_step.sh <tag> [bench args...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
O=$R/gpurun_out/prof_$TAG
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d "$O" -o t -- python3 "$R/bench.py" --no-cpu-baseline "$@" > "$O/bench.log" 2>&1 || { echo "rocprofv3 failed"; tail -5 "$O/bench.log"; exit 1; }
DB=$(ls "$O"/*.db "$O"/*/*.db 2>/dev/null | head -1)
python3 "$R/tools/prof_timeline.py" "$DB" 20 > "$O/timeline.txt" 2>&1
STATS=$(ls "$O"/*kernel_stats.csv "$O"/*/*kernel_stats.csv 2>/dev/null | head -1)
cp "$STATS" "$O/kernel_stats.csv"
grep -h '^{"metric' "$O/bench.log" > "$O/bench_line.json"
find "$O" -name "*kernel_trace.csv" -delete
find "$O" -name "*.db" -delete
head -3 "$O/timeline.txt"
