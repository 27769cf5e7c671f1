#!/bin/bash
# X6 128-tile kernel: schedule variants (terms in front of the barrier QA, VALU instructions per MFMA slot) built with
# -DCGN_X6_QA / -DCGN_X6_VALU_PER_SLOT into lib/variants/ and selected with CGN_HIP_LIB; step time per variant, same box
out=gpurun_out/x6_sched
mkdir -p $out
: > $out/summary.txt
V=learning-disentangled-representations-under-supervision_amd/lib/variants
for round in 1 2; do
for f in $V/lib_q5_v6.so $V/lib_q6_v5.so $V/lib_q6_v4.so $V/lib_q6_v3.so $V/lib_q6_v2.so $V/lib_q5_v4.so; do
  name=$(basename $f .so)
  CGN_HIP_LIB=$PWD/$f timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $out/$name.json 2> $out/$name.err || { echo "$name FAILED"; tail -5 $out/$name.err; exit 1; }
  python -c "
import json,sys
d=json.loads(open('$out/$name.json').read().strip().splitlines()[-1])
print('%-14s %.3f ms/step  gemm kernels %.0f us' % ('$name', d['ms_per_step'], 1e3*d['step_mfma']['gemm_kernel_ms_per_step']))" | tee -a $out/summary.txt
done
done
