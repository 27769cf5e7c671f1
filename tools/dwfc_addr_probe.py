"""Diagnostic: is dW_fc's duration sensitive to where its logits-sized operand / its output lie in memory?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import engine, ops
from ldrs_amd.hip._lib import Axis

dev = torch.device("cuda", 0)
B, T, H, V = 64, 21, 512, 20000
NT = B * T
LIN = Axis(1, 0, 0)
g1 = engine.LstmSeq(B, T, H, dev, False)
g1.hbuf.normal_()
h1, h1ax = g1.hout_operand()
big = torch.randn(NT * V + (64 << 20) // 4, device=dev)
gbig = torch.empty(V * H + (64 << 20) // 4, device=dev)


def bench(fn):
    best, worst = 1e9, 0.0
    for _ in range(3):
        for _ in range(10):
            fn()
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) * 1e3
            best, worst = min(best, t), max(worst, t)
    return best, worst


for off_a in (0, 64, 1024, 4096 + 64, 1 << 18, (1 << 20) + 4096, (3 << 20) + 128, (13 << 20) + 8192):
    for off_g in (0, (1 << 20) + 1024):
        a = big[off_a:off_a + NT * V].view(NT, V)
        G = gbig[off_g:off_g + V * H].view(V, H)
        lo, hi = bench(lambda: ops.gemm(a.t(), h1, G, N=H, K=NT, b_k=h1ax, b_n=LIN))
        print("A offset %9d floats, C offset %8d: %6.1f .. %6.1f us" % (off_a, off_g, lo, hi))
