"""Diagnostic: is a mid-size GEMM bound by the memory system or by the block's own instruction stream?
Runs the same launch with (a) real operands, (b) every row aliased to row 0 (stride-0 row axes: all blocks read the
same few cache lines)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
from ldrs_amd.hip._lib import Axis

dev = torch.device("cuda", 0)


def run(M, N, K, alias, tile, sk, reps=30):
    a = torch.rand(M, K, device=dev)
    b = torch.rand(N, K, device=dev)
    out = torch.empty(M, N, device=dev)
    kw = dict(M=M, N=N, K=K, a_m=Axis(0 if alias else K, 0, 0), a_k=Axis(1, 0, 0), b_k=Axis(1, 0, 0),
              b_n=Axis(0 if alias else K, 0, 0), force_tile=tile, force_splitk=sk)
    for _ in range(3):
        ops.gemm(a, b, out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.gemm(a, b, out, **kw)
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / reps
    print("M=%5d N=%5d K=%5d tile=%3d sk=%d alias=%d : %7.1f us %6.1f TF" % (M, N, K, tile, sk, alias, us,
                                                                          2.0 * M * N * K / us / 1e6), flush=True)


for shape in [(1280, 832, 832), (1280, 832, 3328), (1280, 832, 208), (1280, 2048, 832), (64, 64, 1024), (4096, 4096, 4096)]:
    for tile in (64, 128):
        for alias in (0, 1):
            run(*shape, alias, tile, 1)
