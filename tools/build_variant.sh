#!/bin/bash
# Builds lib/libcgn_hip_v.so: the in-tree objects with ONE translation unit recompiled with extra -D flags (A/B runs of
# kernel variants through CGN_HIP_LIB). usage: build_variant.sh <source.hip> <out.so> [-DNAME=VALUE ...]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/learning-disentangled-representations-under-supervision_amd
SRC=$1; OUT=$2; shift 2
O=/tmp/variant_$(basename "$SRC" .hip)_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function "$@" -c "$P/csrc/$SRC" -o "$O"
OBJS=$(ls "$P"/build/*.o | grep -v "/$(basename "$SRC" .hip).o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS "$O"
rm -f "$O"
echo "built $OUT"
