"""Diagnostic: the dW_fc product (dlogits^T . h1 over the [B,T+1,H] history buffer) and the vocabulary projection alone
on the GPU, cfg2 shapes: us per launch (min of 3 batches after warm-up)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import engine, ops
from ldrs_amd.hip._lib import Axis

dev = torch.device("cuda", 0)
ops.WORKSPACE_BYTES = 512 << 20
B, T, H, V = 64, 21, 512, 20000
NT = B * T
LIN = Axis(1, 0, 0)
g1 = engine.LstmSeq(B, T, H, dev, False)
g1.hbuf.normal_()
h1, h1ax = g1.hout_operand()
dlogits = torch.randn(NT, V, device=dev)
W = torch.randn(V, H, device=dev)
G = torch.empty(V, H, device=dev)
logits = torch.empty(NT, V, device=dev)
bias = torch.zeros(V, device=dev)


def bench(fn):
    best = 1e9
    for _ in range(3):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 20)
    return best


print("dW_fc  %7.1f us" % bench(lambda: ops.gemm(dlogits.t(), h1, G, N=H, K=NT, b_k=h1ax, b_n=LIN)))
for sk in (2, 3, 4):
    print("dW_fc split-K %d (product + reduce) %7.1f us" % (sk, bench(lambda: ops.gemm(dlogits.t(), h1, G, N=H, K=NT, b_k=h1ax,
                                                                                     b_n=LIN, force_splitk=sk))))
for sk in (2, 3):
    print("fc split-K %d (product + reduce) %7.1f us" % (sk, bench(lambda: ops.gemm(h1, W.t(), logits, M=NT, K=H, a_m=h1ax,
                                                                                  a_k=LIN, bias0=bias, force_splitk=sk))))
print("fc     %7.1f us" % bench(lambda: ops.gemm(h1, W.t(), logits, M=NT, K=H, a_m=h1ax, a_k=LIN, bias0=bias)))
