"""Diagnostic: the split-K launches of the bf16 x bf16 kernel alone (soft embedding with the exp loader, dH1, the LSTM dW
shape) at configs[1] and configs[3] shapes; run once per library build (CGN_HIP_LIB) to compare block -> tile mappings."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timed(fn, reps=10):
    best = 1e30
    for _ in range(3):
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


print(os.environ.get("CGN_HIP_LIB", "in-tree library"))
for name, M, V, E, H in (("cfg2", 1344, 20000, 304, 512), ("cfg4", 10496, 50000, 304, 1024)):
    logits = (torch.randn(M, V, device=dev) * 2).bfloat16()
    lse = torch.logsumexp(logits.float(), 1)
    tab_t = (torch.randn(E, V, device=dev) * 0.1).bfloat16()       # B operand [N = E, K = V]
    out = torch.empty(M, E, device=dev)
    us = timed(lambda: ops.gemm_bb(logits, tab_t, out, a_exp=(1.0, lse)))
    print("%s soft embedding (exp loader) M %5d N %4d K %5d: %8.1f us" % (name, M, E, V, us))
    w_t = (torch.randn(H, V, device=dev) * 0.1).bfloat16()         # B operand [N = H, K = V]
    dh = torch.empty(M, H, device=dev)
    us = timed(lambda: ops.gemm_bb(logits, w_t, dh))
    print("%s dH1                         M %5d N %4d K %5d: %8.1f us" % (name, M, H, V, us))
    del logits, tab_t, w_t
a = torch.randn(4096, 10240, device=dev).bfloat16()
b = torch.randn(1024, 10240, device=dev).bfloat16()
o = torch.empty(4096, 1024, device=dev)
print("cfg4 LSTM dW M 4096 N 1024 K 10240: %8.1f us" % timed(lambda: ops.gemm_bb(a, b, o)))
a = torch.randn(2484, 10240, device=dev).bfloat16()
b = torch.randn(832, 10240, device=dev).bfloat16()
o = torch.empty(2484, 832, device=dev)
print("cfg4 highway dW M 2484 N 832 K 10240: %8.1f us" % timed(lambda: ops.gemm_bb(a, b, o)))
