#!/bin/bash
out=gpurun_out/group
mkdir -p $out
: > $out/summary.txt
timeout -k 10 900 python -u -m pytest tests/test_gemm_gpu.py tests/test_fullsize_gpu.py tests/test_step_gpu.py tests/test_step_golden_gpu.py tests/test_graph_gpu.py -m gpu -q -x --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -2 $out/tests.log
run() {
  name=$1; wl=$2; steps=$3; shift 3
  env "$@" timeout -k 10 200 python bench.py --workload $wl --steps $steps --warmup 5 --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step  step_mfma %.3f' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation']))" | tee -a $out/summary.txt
}
run group_on cfg2 80 A=1 || exit 1
run group_off cfg2 80 CGN_GROUP_SIBLINGS=0 || exit 1
run group_on_again cfg2 80 A=1 || exit 1
run group_off_again cfg2 80 CGN_GROUP_SIBLINGS=0 || exit 1
run cfg4_group_on cfg4 20 A=1 || exit 1
run cfg4_group_off cfg4 20 CGN_GROUP_SIBLINGS=0 || exit 1
timeout -k 10 200 python tools/step_stamps.py --out $out/stamps_cfg2.txt 2>&1 | grep -v amdgpu.ids | tail -36
