#!/bin/bash
out=gpurun_out/misc
mkdir -p $out
: > $out/summary.txt
timeout -k 10 900 python -u -m pytest tests/test_kernels_gpu.py tests/test_gemm_gpu.py tests/test_step_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x --timeout 300 --timeout-method thread -k "adam or fused_log or cfg2 or switches or fused_chains" > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() {
  name=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step' % ('$name', d['ms_per_step']))" | tee -a $out/summary.txt
}
run base cfg2 80 A=1 || exit 1
run few256 cfg2 80 CGN_X6_FEW_MINK=256 || exit 1
run few128 cfg2 80 CGN_X6_FEW_MINK=128 || exit 1
run base_again cfg2 80 A=1 || exit 1
run few256_again cfg2 80 CGN_X6_FEW_MINK=256 || exit 1
