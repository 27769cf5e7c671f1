#!/bin/bash
# round 3: persistent LSTM launches after the XCD-contiguous placement + self-resetting counters: tests, then A/B timing
set -o pipefail
out=gpurun_out/r03_lstm
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -k "lstm" tests/test_zz_no_alloc.py \
    "tests/test_loop_gpu.py::test_timed_out_persistent_launch_stops_the_training_loop" -m gpu -x -q --timeout 180 \
    --timeout-method thread > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
for m in 1 0; do
  echo "== CGN_LSTM_XCD_MAP=$m" | tee -a $out/lstm_bench.txt
  CGN_LSTM_XCD_MAP=$m timeout -k 10 120 python tools/lstm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out/lstm_bench.txt || exit 1
done
echo "== cfg4 shapes" | tee -a $out/lstm_bench.txt
timeout -k 10 120 python tools/lstm_bench.py 256 1024 40 2>&1 | grep -v amdgpu.ids | tee -a $out/lstm_bench.txt || exit 1
