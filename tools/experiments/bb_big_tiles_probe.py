"""Diagnostic: the three big products of configs[3] shapes on the bf16 x bf16 kernel, per tile variant and split."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timed(fn, reps=5):
    best = 1e30
    for _ in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


for name, M, N, K, c16, variants in (
        ("projection", 10496, 50000, 1024, True, [(256, 1), (192, 1), (128, 1)]),
        ("dW_fc", 50000, 1024, 10496, False, [(256, 1), (192, 1), (128, 1)]),
        ("dH1", 10496, 1024, 50000, False, [(256, 3), (256, 2), (256, 4), (256, 6), (192, 3), (192, 2), (192, 4), (128, 2), (128, 3), (128, 4)])):
    a = torch.randn(M, K, device=dev).bfloat16()
    b = torch.randn(N, K, device=dev).bfloat16()
    o = torch.empty(M, N, device=dev, dtype=torch.bfloat16 if c16 else torch.float32)
    res = []
    for t, sk in variants:
        us = timed(lambda: ops.gemm_bb(a, b, o, force_tile=t, force_splitk=sk))
        res.append("t%d/sk%d %.0f us %.0f TF/s" % (t, sk, us, 2.0 * M * N * K / us / 1e6))
    print(name, " | ".join(res), flush=True)
    del a, b, o
