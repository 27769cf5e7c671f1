#!/bin/bash
# same-box A/B of the round-3 switches: bench.py --steps 80, no CPU baseline
out=gpurun_out/r03_ab
mkdir -p $out
: > $out/summary.txt
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
run no_fused_no_x6 CGN_FUSED_CHAINS=0 CGN_LSTM_X6=0 || exit 1
run no_fused_no_x6_no_xcd CGN_FUSED_CHAINS=0 CGN_LSTM_X6=0 CGN_LSTM_XCD_MAP=0 || exit 1
run default_again A=1 || exit 1
timeout -k 10 200 python tools/step_stamps.py --out $out/stamps_cfg2.txt 2>&1 | grep -v amdgpu.ids
