#!/bin/bash
# A/B of the step's GEMM shapes: lib/libcgn_hip_prev.so (a copy of the library before an edit) against the current build
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CASES=('(4096,4096,4096,"nt",128)' '(4096,4096,4096,"nt",64)' '(1344,20000,512,"nt",0)' '(20000,512,1344,"tn",0)' '(1344,512,20000,"nn",0)' '(1344,300,20000,"nn",0)' '(1344,20000,300,"nt",0)' \
 '(1280,2484,828,"nt",0)' '(1280,828,2484,"nn",0)' '(2484,828,1280,"tn",0)' '(2048,512,1344,"tn",0)' '(1344,2048,512,"nt",0)')
for lib in prev new; do
  echo "== $lib"
  if [ $lib = prev ]; then export CGN_HIP_LIB=$R/learning-disentangled-representations-under-supervision_amd/lib/libcgn_hip_prev.so; else unset CGN_HIP_LIB; fi
  timeout -k 10 200 python3 $R/tools/gemm_bench.py "${CASES[@]}" 2>&1 | grep "TFLOP" || exit 1
done
