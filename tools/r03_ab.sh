#!/bin/bash
# same-box A/B of the round-3 switches: bench.py --steps 80, no CPU baseline
out=gpurun_out/r03_ab
mkdir -p $out
: > $out/summary.txt
if [ "$1" = "test" ]; then
  timeout -k 10 600 python -u -m pytest tests/test_fullsize_gpu.py "tests/test_step_gpu.py::test_wake_sleep_step_matches_oracle" tests/test_step_golden_gpu.py tests/test_graph_fullsize_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
  tail -2 $out/tests.log
fi
run() {
  name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 80 --warmup 10 --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step  step_mfma %.3f' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation']))" | tee -a $out/summary.txt
}
run default A=1 || exit 1
run no_fused CGN_FUSED_CHAINS=0 || exit 1
run no_x6 CGN_LSTM_X6=0 || exit 1
run default_again A=1 || exit 1
timeout -k 10 200 python tools/step_stamps.py --out $out/stamps_cfg2.txt 2>&1 | grep -v amdgpu.ids
