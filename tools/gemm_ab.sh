#!/bin/bash
# A/B of two builds of the library on the CGN step's GEMM shapes (runs on the GPU box): $1 = old .so, $2 = new .so
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
CASES='(1344,20000,512,"nt",0,0,True) (1344,300,20000,"nn",0) (1344,512,20000,"nn",0) (1344,20000,300,"nt",0) (20000,512,1344,"tn",0) (1280,2484,828,"nt",0) (1280,2048,828,"nt",0) (1280,828,2484,"nn",0) (1280,828,2048,"nn",0) (2484,828,1280,"tn",0) (2048,828,1280,"tn",0) (1344,2048,300,"nt",0) (1344,2048,512,"nt",0) (1344,512,2048,"nn",0) (2048,512,1344,"tn",0) (2048,300,1344,"tn",0) (64,64,1024,"nt",64) (64,64,1024,"nt",128) (4096,4096,4096,"nt",128) (4096,4096,4096,"nt",64)'
for lib in "$@"; do
  echo "== $lib"
  CGN_HIP_LIB=$lib timeout -k 10 200 python3 $R/tools/gemm_bench.py $CASES 2>&1 | grep "TFLOP"
done
