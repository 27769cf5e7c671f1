#!/bin/bash
out=gpurun_out/x6_tiles2
mkdir -p $out
: > $out/summary.txt
run() {
  name=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline --gemm-table $out/$name.table.txt > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step  step_mfma %.3f  gemm kernels %.0f us' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation'], 1e3*d['step_mfma']['gemm_kernel_ms_per_step']))" | tee -a $out/summary.txt
}
run t128_mink512 cfg2 80 CGN_X6_TILE=128 || exit 1
run t128_mink256_b512 cfg2 80 CGN_X6_TILE=128 CGN_X6_MINK=256 CGN_X6_BLOCKS=512 || exit 1
run t128_mink256_b1024 cfg2 80 CGN_X6_TILE=128 CGN_X6_MINK=256 CGN_X6_BLOCKS=1024 || exit 1
run t128_mink128_b512 cfg2 80 CGN_X6_TILE=128 CGN_X6_MINK=128 CGN_X6_BLOCKS=512 || exit 1
run t128_mink128_b1024 cfg2 80 CGN_X6_TILE=128 CGN_X6_MINK=128 CGN_X6_BLOCKS=1024 || exit 1
run cfg4_t64 cfg4 20 A=1 || exit 1
run cfg4_t128 cfg4 20 CGN_X6_TILE=128 || exit 1
timeout -k 10 200 env CGN_X6_TILE=128 python tools/step_stamps.py --out $out/stamps_cfg2.txt 2>&1 | grep -v amdgpu.ids | tail -40
