#!/bin/bash
# Runs on the GPU box: the step-parity cases at BASELINE configs[1] and configs[3] model size under other seeds
# (initial weights, batches, dropout masks): fp32 mode at 1e-4, bf16 mode at 1e-2, reports under gpurun_out/seed_sweep.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/seed_sweep
rm -rf "$O" && mkdir -p "$O"
cd "$R"
for s in 1 2 3 4; do
    CGN_PARITY_SEED=$s CGN_PARITY_REPORT=$O timeout -k 10 500 python -u -m pytest -x -q --timeout 400 --timeout-method thread \
        tests/test_step_gpu.py -k "(matches_oracle and (cfg2 or cfg4)) or (test_wake_sleep_step_bf16_mode and (cfg2 or cfg4) and not fp32_storage)" \
        > "$O/seed$s.log" 2>&1 || { echo "seed $s FAILED"; tail -30 "$O/seed$s.log"; exit 1; }
    tail -1 "$O/seed$s.log"
done
grep -H "worst per kind" "$O"/parity_*.txt | sed "s|$O/||"
