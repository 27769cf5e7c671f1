"""Diagnostic: the fused softmax-backward product (dlogits) at cfg2 shapes with fp32 / bf16 logits, with and without the
transposed copy: us per launch (min of 5 after warm-up)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops

dev = torch.device("cuda", 0)
M, V, E = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1344, 20000, 300)
g = torch.Generator().manual_seed(0)
dout = torch.randn(M, E, generator=g).to(dev)
emb = (torch.randn(V, E, generator=g) * 0.3).to(dev)
lse = torch.full((M,), 10.0, device=dev)
rowdot = torch.zeros(M, device=dev)
tgt = torch.randint(0, V, (M,), generator=g).to(dev)


def run(dt, with_ct):
    logits = torch.randn(M, V, device=dev).to(dt)
    ct = torch.empty(V, M, device=dev, dtype=torch.bfloat16) if with_ct else None
    best = 1e9
    for rep in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.gemm(dout, emb.t(), logits, compute="bf16", c_t=ct,
                 softmax_bwd=dict(logits=logits, lse=lse, rowdot=rowdot, target=tgt, inv_tau=1.0, ce_scale=0.01))
        e1.record()
        torch.cuda.synchronize()
        if rep >= 3:
            best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


print("fp32 logits            %7.1f us" % run(torch.float32, False))
print("bf16 logits            %7.1f us" % run(torch.bfloat16, False))
print("bf16 logits + ct       %7.1f us" % run(torch.bfloat16, True))
