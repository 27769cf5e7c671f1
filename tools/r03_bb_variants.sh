#!/bin/bash
# Runs on the GPU box: the bb probe's speed table for each variant library given (A/B of kernel schedules).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r03_bb_variants
mkdir -p "$O"
cd "$R"
L=$R/learning-disentangled-representations-under-supervision_amd/lib
for v in "$@"; do
  lib=$L/libcgn_hip$v.so
  echo "== $lib"
  CGN_HIP_LIB=$lib timeout -k 10 400 python3 -u tools/gemm_bb_probe.py "$O/probe$v.txt" > "$O/log$v.txt" 2>&1 || { tail "$O/log$v.txt"; exit 1; }
  grep -v "^check" "$O/probe$v.txt" | cut -c1-60,113-250
done
