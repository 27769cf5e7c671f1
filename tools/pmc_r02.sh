#!/bin/bash
# Runs on the GPU box: HBM traffic per kernel launch of the bench step (two separate rocprofv3 --pmc passes).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/pmc_r02
rm -rf "$O" && mkdir -p "$O/fetch" "$O/write"
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/fetch" -o f -- python3 "$R/bench.py" --eager --no-cpu-baseline --steps 5 --warmup 2 > "$O/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$O/fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/write" -o w -- python3 "$R/bench.py" --eager --no-cpu-baseline --steps 5 --warmup 2 > "$O/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$O/write.log"; exit 1; }
python3 "$R/tools/pmc_traffic.py" "$O/fetch" "$O/write" "$O/r02_pmc_traffic.json" > "$O/r02_pmc_traffic.txt" 2>&1
head -30 "$O/r02_pmc_traffic.txt"
find "$O" -name "*counter_collection.csv" -delete
