#!/bin/bash
out=gpurun_out/x3
mkdir -p $out
: > $out/summary.txt
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py tests/test_graph_fullsize_gpu.py -m gpu -q -x --timeout 300 --timeout-method thread -k "two_term or bf16 or cfg4" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() {
  name=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --compute bf16 > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step' % ('$name', d['ms_per_step']))" | tee -a $out/summary.txt
}
run cfg4_bf16_x3 cfg4 20 A=1 || exit 1
run cfg4_bf16_fp32rec cfg4 20 CGN_LSTM_BF16_RT=0 || exit 1
run cfg4_bf16_x3_again cfg4 20 A=1 || exit 1
run cfg2_bf16 cfg2 80 A=1 || exit 1
