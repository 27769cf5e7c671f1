#!/bin/bash
# round 3: persistent LSTM launches: tests, then A/B timing of the X6 products (and of the XCD placement)
set -o pipefail
out=gpurun_out/r03_lstm
mkdir -p $out
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -k "lstm" tests/test_zz_no_alloc.py \
    "tests/test_loop_gpu.py::test_timed_out_persistent_launch_stops_the_training_loop" -m gpu -x -q --timeout 180 \
    --timeout-method thread > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -3 $out/tests.log
: > $out/lstm_bench.txt
for m in "CGN_LSTM_X6=1" "CGN_LSTM_X6=0" "CGN_LSTM_X6=1 CGN_LSTM_XCD_MAP=0"; do
  echo "== $m" | tee -a $out/lstm_bench.txt
  env $m timeout -k 10 120 python tools/lstm_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $out/lstm_bench.txt || exit 1
done
echo "== cfg1 shapes (B 32, H 256, L 15)" | tee -a $out/lstm_bench.txt
timeout -k 10 120 python tools/lstm_bench.py 32 256 15 2>&1 | grep -v amdgpu.ids | tee -a $out/lstm_bench.txt || exit 1
echo "== cfg4 shapes" | tee -a $out/lstm_bench.txt
timeout -k 10 120 python tools/lstm_bench.py 256 1024 40 2>&1 | grep -v amdgpu.ids | tee -a $out/lstm_bench.txt || exit 1
