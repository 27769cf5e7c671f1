#!/bin/bash
# Runs on the GPU box: the bf16 mode with and without the bf16 x bf16 kernel (csrc/bb_gemm.hip), cfg2 and configs[3] shapes.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03_bb
mkdir -p "$O"
cd "$R"
for w in cfg2 cfg4; do
  CGN_BB_GEMM=0 timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload $w --compute bf16 --steps 10 > "$O/line_${w}_off.json" 2> "$O/err.txt" || { tail "$O/err.txt"; exit 1; }
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --workload $w --compute bf16 --steps 10 --gemm-table "$O/gemm_table_${w}_bf16.txt" > "$O/line_${w}_on.json" 2> "$O/err.txt" || { tail "$O/err.txt"; exit 1; }
done
cat "$O"/line_*.json | python3 -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print('%.3f ms/step %s gemm %.2f ms | %s' % (d['ms_per_step'], d['dtype'], d['step_mfma']['gemm_kernel_ms_per_step'], d['roofline']['kernel'][:80]))"
