#!/bin/bash
# Runs on the GPU box (through gpurun): the rocprofv3 passes the committed profiles/ files come from.
#   1. kernel trace + stats of the default bench command (fp32, hipGraph replay)      -> gpurun_out/prof/f32
#   2. the same for --compute bf16                                                      -> gpurun_out/prof/bf16
#   3. PMC FETCH_SIZE and WRITE_SIZE, one pass each (never combined with traces), eager  -> gpurun_out/prof/pmc_*
# Each GPU step runs under its own time limit; the script stops at the first failure.
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d "$O/f32" -o f32 -- python3 "$R/bench.py" --no-cpu-baseline > "$O/f32.log" 2>&1
echo "f32 trace done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv rocpd -d "$O/bf16" -o bf16 -- python3 "$R/bench.py" --no-cpu-baseline --compute bf16 > "$O/bf16.log" 2>&1
echo "bf16 trace done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o f -- python3 "$R/bench.py" --eager --no-cpu-baseline --steps 5 --warmup 2 > "$O/pmc_fetch.log" 2>&1
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o w -- python3 "$R/bench.py" --eager --no-cpu-baseline --steps 5 --warmup 2 > "$O/pmc_write.log" 2>&1
echo "write pass done"
python3 "$R/tools/pmc_traffic.py" "$O/pmc_fetch" "$O/pmc_write" "$O/pmc_traffic.json" > "$O/pmc_traffic.txt"
# keep what travels back small: the raw per-dispatch counter csv files are tens of MB
find "$O" -name "*counter_collection.csv" -delete
find "$O" -name "*kernel_trace.csv" -delete
ls -la "$O" "$O/f32" "$O/bf16"
