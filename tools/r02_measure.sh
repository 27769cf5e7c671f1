#!/bin/bash
# Runs on the GPU box: the round's final measurements (bench lines of the default and the side workloads, rocprofv3
# kernel stats + timeline of the default command and of the eager variant). Everything lands under gpurun_out/r02/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r02
rm -rf "$O" && mkdir -p "$O"
cd "$R"
run() { name=$1; shift; timeout -k 10 400 python3 bench.py "$@" > "$O/$name.log" 2>&1 || { echo "$name FAILED"; tail -5 "$O/$name.log"; exit 1; }; grep -h '^{"metric' "$O/$name.log" > "$O/$name.json"; python3 -c "import json,sys; d=json.load(open('$O/$name.json')); print('$name', round(d['ms_per_step'],3), 'ms', round(d['value']), 'sent/s', 'roofline frac', round(d['roofline']['frac'],3), 'step util', round(d['step_mfma']['utilisation'],3))"; }
run bench_default --gemm-table "$O/gemm_table.txt"
run bench_bf16 --compute bf16 --no-cpu-baseline --gemm-table "$O/gemm_table_bf16.txt"
run bench_cfg4 --workload cfg4 --steps 10 --warmup 3
run bench_cfg4_bf16 --workload cfg4 --compute bf16 --steps 10 --warmup 3
run bench_eager --eager --no-cpu-baseline
run bench_forcedp --force-dp --no-cpu-baseline
CGN_DP_EAGER=1 run bench_forcedp_eager --force-dp --no-cpu-baseline
bash tools/prof_step.sh r02 > "$O/prof.log" 2>&1 && bash tools/prof_step.sh r02e --eager >> "$O/prof.log" 2>&1
tail -n 2 "$O/prof.log"
