#!/bin/bash
out=gpurun_out/r03_hwq
mkdir -p $out
run() {
  name=$1; shift
  env "$@" timeout -k 10 120 python bench.py --steps 80 --warmup 10 --no-cpu-baseline $EXTRA > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; return 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-34s %.3f ms/step  step_mfma %.3f' % ('$name', d['ms_per_step'], d['step_mfma']['utilisation']))" | tee -a $out/summary.txt
}
EXTRA=""
run plain_q4 A=1
run plain_q8 GPU_MAX_HW_QUEUES=8
run plain_q6 GPU_MAX_HW_QUEUES=6
EXTRA="--force-dp"
run dp_q4 A=1
run dp_q8 GPU_MAX_HW_QUEUES=8
GPU_MAX_HW_QUEUES=8 timeout -k 10 200 python tools/step_stamps.py --force-dp --out $out/stamps_force_dp_q8.txt 2>&1 | grep -v "amdgpu.ids\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl\|socket.cpp" | sed -n 1,3p
sed -n 22,34p $out/stamps_force_dp_q8.txt
timeout -k 10 300 python tools/gemm_bench.py '(1280,2048,825,"nt",0)' '(1280,4096,825,"nt",0)' '(1280,825,2048,"nn",0)' '(1280,825,4096,"nn",0)' '(2048,825,1280,"tn",0)' '(4096,825,1280,"tn",0)' '(1280,2484,825,"nt",0)' '(1344,300,20000,"nn",0)' 2>&1 | grep -v amdgpu | tee $out/gemm_merge.txt
