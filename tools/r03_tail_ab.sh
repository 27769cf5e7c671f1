#!/bin/bash
# A/B of the tail-schedule knobs (hip/step.py): bench.py --steps 60, no CPU baseline
out=gpurun_out/r03_tail_ab
mkdir -p $out
run() {
  name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 60 --warmup 10 --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-28s %.3f ms/step  step_mfma %.3f' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation']))" | tee -a $out/summary.txt
}
run base A=1 || exit 1
run splitk1 CGN_TAIL_SPLITK=1 || exit 1
run splitk1_tile128 CGN_TAIL_SPLITK=1 CGN_TAIL_TILE=128 || exit 1
run splitk2 CGN_TAIL_SPLITK=2 || exit 1
run hw_early CGN_HW_DW_EARLY=1 || exit 1
run hw_early_splitk1 CGN_HW_DW_EARLY=1 CGN_TAIL_SPLITK=1 || exit 1
run base2 A=1 || exit 1
