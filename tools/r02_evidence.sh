#!/bin/bash
# Runs on the GPU box: the round's evidence files (parity reports, two-rank rehearsal), written under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
rm -rf gpurun_out/parity_r02 && mkdir -p gpurun_out/parity_r02
CGN_PARITY_REPORT=gpurun_out/parity_r02 timeout -k 10 900 python -u -m pytest tests/test_step_gpu.py -x -q --timeout 600 --timeout-method thread > gpurun_out/parity_r02/pytest.log 2>&1 || { tail -20 gpurun_out/parity_r02/pytest.log; exit 1; }
tail -2 gpurun_out/parity_r02/pytest.log
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp2_gpu_check.py --graphed > gpurun_out/dp2_r02.log 2>&1
echo "dp2 rc=$?" >> gpurun_out/dp2_r02.log
grep "^dp2" gpurun_out/dp2_r02.log
