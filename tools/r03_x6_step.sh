#!/bin/bash
# X6 products inside the step: parity tests, then the bench with and without them (same box)
out=gpurun_out/x6_step
mkdir -p $out
: > $out/summary.txt
if [ "$1" != "notest" ]; then
timeout -k 10 900 python -u -m pytest tests/test_gemm_gpu.py tests/test_step_gpu.py tests/test_step_golden_gpu.py tests/test_fullsize_gpu.py -m gpu -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
fi
run() {
  name=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --gemm-table $out/$name.table.txt > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step  step_mfma %.3f' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation']))" | tee -a $out/summary.txt
}
run cfg2_x6 cfg2 80 CGN_GEMM_X6=1 || exit 1
run cfg2_mfma cfg2 80 CGN_GEMM_X6=0 || exit 1
run cfg4_x6 cfg4 20 CGN_GEMM_X6=1 || exit 1
run cfg4_mfma cfg4 20 CGN_GEMM_X6=0 || exit 1
run cfg2_x6_again cfg2 80 CGN_GEMM_X6=1 || exit 1
