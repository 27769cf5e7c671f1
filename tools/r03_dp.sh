#!/bin/bash
# round 3: the data-parallel schedule on a one-GPU box: one-rank RCCL tests, --force-dp bench against the plain bench,
# and the two-rank (gloo, both ranks on one GPU) rehearsal, eager and graphed
set -o pipefail
out=gpurun_out/r03_dp
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_graph_gpu.py -m gpu -x -q --timeout 200 --timeout-method thread > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
for mode in "" "--force-dp"; do
  timeout -k 10 200 python bench.py --steps 80 --warmup 10 --no-cpu-baseline $mode > $out/bench$mode.json 2> $out/bench$mode.err || { tail -20 $out/bench$mode.err; exit 1; }
  python -c "
import json
d=json.loads(open('$out/bench$mode.json').read().strip().splitlines()[-1])
print('bench %-12s %.3f ms/step  launch: %s' % ('$mode', d['ms_per_step'], d['config']['launch']))" | tee -a $out/summary.txt
done
DP2_DIAG=20 DP2_WATCHDOG=90 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/dp2_gpu_check.py > $out/dp2.log 2>&1
rc=$?
grep "^dp2 " $out/dp2.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || { tail -30 $out/dp2.log; exit $rc; }
timeout -k 10 200 python tools/step_stamps.py --force-dp --out $out/stamps_force_dp.txt 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp" | head -3
DP2_DIAG=20 DP2_WATCHDOG=90 timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 tools/dp2_gpu_check.py --graphed > $out/dp2_graphed.log 2>&1
rc=$?
grep "^dp2 " $out/dp2_graphed.log | tee -a $out/summary.txt
[ $rc -eq 0 ] || { tail -30 $out/dp2_graphed.log; exit $rc; }
