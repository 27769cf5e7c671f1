#!/bin/bash
# X6: new GEMM tests, then the 128-tile for the mid-size products (planner experiment knobs), same box
out=gpurun_out/x6_tiles
mkdir -p $out
: > $out/summary.txt
timeout -k 10 900 python -u -m pytest tests/test_gemm_gpu.py -m gpu -q -x --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() {
  name=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --gemm-table $out/$name.table.txt > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step  step_mfma %.3f  gemm kernels %.0f us' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation'], 1e3*d['step_mfma']['gemm_kernel_ms_per_step']))" | tee -a $out/summary.txt
}
run default cfg2 80 A=1 || exit 1
run t128_b1024 cfg2 80 CGN_X6_TILE=128 CGN_X6_BLOCKS=1024 || exit 1
run t128_b512 cfg2 80 CGN_X6_TILE=128 CGN_X6_BLOCKS=512 || exit 1
run t128_b768 cfg2 80 CGN_X6_TILE=128 CGN_X6_BLOCKS=768 || exit 1
run default_again cfg2 80 A=1 || exit 1
