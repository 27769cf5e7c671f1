#!/bin/bash
# A/B of the step's GEMM shapes: previous library (tile rounds + split-K reduce) against the stream-K kernel
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CASES=('(1344,20000,512,"nt",0)' '(20000,512,1344,"tn",0)' '(1344,512,20000,"nn",0)' '(1344,300,20000,"nn",0)' '(1344,20000,300,"nt",0)' \
 '(1280,2484,828,"nt",0)' '(1280,828,2484,"nn",0)' '(2484,828,1280,"tn",0)' '(1280,828,2048,"nn",0)' '(2048,512,1344,"tn",0)' \
 '(2048,828,1280,"tn",0)' '(1280,2048,828,"nt",0)' '(2048,300,1344,"tn",0)' '(1344,2048,300,"nt",0)' '(1344,2048,512,"nt",0)' '(4096,4096,4096,"nt",0)' \
 '(1344,20000,512,"nt",64)' '(20000,512,1344,"tn",64)' '(1344,20000,300,"nt",64)' '(1280,2484,828,"nt",128)' '(2484,828,1280,"tn",128)')
for lib in prev new; do
  echo "== $lib"
  if [ $lib = prev ]; then export CGN_HIP_LIB=$R/learning-disentangled-representations-under-supervision_amd/lib/libcgn_hip_prev.so; else unset CGN_HIP_LIB; fi
  timeout -k 10 200 python3 $R/tools/gemm_bench.py "${CASES[@]}" 2>&1 | grep "TFLOP" || exit 1
done
