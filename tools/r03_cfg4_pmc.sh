#!/bin/bash
# Runs on the GPU box: HBM traffic per kernel launch of the bf16 mode at configs[3] shapes (two rocprofv3 --pmc passes,
# FETCH_SIZE and WRITE_SIZE, of the eager step; tools/pmc_traffic.py applies the gfx950 corrections).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03_cfg4_pmc
rm -rf "$O" && mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
CMD="--eager --no-cpu-baseline --workload cfg4 --compute bf16 --steps 3 --warmup 1"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- python3 "$R/bench.py" $CMD > "$O/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$O/fetch.log"; exit 1; }
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- python3 "$R/bench.py" $CMD > "$O/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$O/write.log"; exit 1; }
python3 "$R/tools/pmc_traffic.py" "$O/fetch" "$O/write" "$O/pmc_traffic.json" > "$O/pmc_traffic.txt" 2>&1
find "$O" -name "*counter_collection.csv" -delete
cat "$O/pmc_traffic.txt"
