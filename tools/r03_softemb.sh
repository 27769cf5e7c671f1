#!/bin/bash
out=gpurun_out/r03_softemb
mkdir -p $out
timeout -k 10 300 python tools/gemm_bench.py '(1344,300,20000,"nn",128,0)' '(1344,300,20000,"nn",64,0)' '(1344,300,20000,"nn",64,6)' '(1344,300,20000,"nn",64,9)' '(1344,300,20000,"nn",64,12)' '(1344,300,20000,"nn",64,16)' '(1344,300,20000,"nn",64,20)' '(1344,20000,300,"nt",128,0)' '(1344,20000,300,"nt",64,0)' 2>&1 | grep -v amdgpu | tee $out/gemm.txt
run() {
  name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 80 --warmup 10 --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step  step_mfma %.3f' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation']))" | tee -a $out/summary.txt
}
run default A=1
run softemb64 CGN_SOFTEMB_TILE=64
run softemb64_sk12 CGN_SOFTEMB_TILE=64 CGN_SOFTEMB_SPLITK=12
run softemb64_sk16 CGN_SOFTEMB_TILE=64 CGN_SOFTEMB_SPLITK=16
run default2 A=1
