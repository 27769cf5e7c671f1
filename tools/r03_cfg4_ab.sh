#!/bin/bash
# fused-kernel tests, then a same-box A/B of the round-3 switches at cfg4 (the r02 line was 54.9 ms)
out=gpurun_out/r03_cfg4
mkdir -p $out
: > $out/summary.txt
timeout -k 10 600 python -u -m pytest tests/test_fused_gpu.py "tests/test_fullsize_gpu.py::test_fused_chains_match_the_chains_of_small_kernels" -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() {
  name=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step  step_mfma %.3f' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation']))" | tee -a $out/summary.txt
}
run cfg2_default cfg2 80 A=1 || exit 1
run cfg4_default cfg4 20 A=1 || exit 1
run cfg4_no_fused cfg4 20 CGN_FUSED_CHAINS=0 || exit 1
run cfg4_no_xcd cfg4 20 CGN_LSTM_XCD_MAP=0 || exit 1
run cfg4_no_x6 cfg4 20 CGN_LSTM_X6=0 || exit 1
run cfg4_default_again cfg4 20 A=1 || exit 1
