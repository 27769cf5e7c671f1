"""Diagnostic: dW_fc timed inside its neighbourhood of the step (fused softmax backward -> dH1 -> dW_fc), cfg2 shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import engine, ops
from ldrs_amd.hip._lib import Axis

dev = torch.device("cuda", 0)
B, T, H, V, E = 64, 21, 512, 20000, 300
NT = B * T
LIN = Axis(1, 0, 0)
g = torch.Generator().manual_seed(0)
g1 = engine.LstmSeq(B, T, H, dev, False)
g1.hbuf.normal_()
h1, h1ax = g1.hout_operand()
W = (torch.randn(V, H, generator=g) * 0.05).to(dev)
emb = (torch.randn(V, E, generator=g) * 0.3).to(dev)
dout = (torch.randn(NT, E, generator=g) * 1e-3).to(dev)
bias = torch.zeros(V, device=dev)
tgt = torch.randint(0, V, (NT,), generator=g).to(dev)
G = torch.empty(V, H, device=dev)
dH1 = torch.empty(NT, H, device=dev)
rowdot = torch.zeros(NT, device=dev)
mode = sys.argv[1] if len(sys.argv) > 1 else "chain"


def once(timed):
    logits = torch.empty(NT, V, device=dev)
    ops.gemm(h1, W.t(), logits, M=NT, K=H, a_m=h1ax, a_k=LIN, bias0=bias)
    lse, nll, ce = ops.softmax_nll_fwd(logits, tgt, 1.0)
    if mode == "chain":
        ops.gemm(dout, emb.t(), logits,
                 softmax_bwd=dict(logits=logits, lse=lse, rowdot=rowdot, target=tgt, inv_tau=1.0, ce_scale=79.0 / NT))
    elif mode == "randn":
        logits.normal_()
    ops.gemm(logits, W, dH1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.gemm(logits.t(), h1, G, N=H, K=NT, b_k=h1ax, b_n=LIN)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3, float(logits.abs().mean()), float((logits == 0).float().mean())


for _ in range(3):
    once(False)
ts = [once(True) for _ in range(5)]
print(mode, "dW_fc us:", " ".join("%.1f" % t[0] for t in ts), " mean|dlogits| %.2e zero fraction %.3f" % ts[-1][1:])
