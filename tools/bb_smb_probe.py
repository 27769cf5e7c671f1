"""Diagnostic: the fused softmax / CE backward product of the bf16 x bf16 kernel at configs[3] / configs[1] shapes, with and
without the transposed copy, on both tiles, beside a plain bf16-result product of the same shape (what the epilogue costs).
usage: bb_smb_probe.py [out.txt]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def timed(fn, reps=5):
    us = float("inf")
    for _ in range(3):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = min(us, 1e3 * e0.elapsed_time(e1) / reps)
    return us


def case(M, V, E):
    logits = torch.randn(M, V, device=dev).bfloat16()
    lse = torch.logsumexp(logits.float(), 1)
    tgt = torch.randint(0, V, (M,), device=dev)
    emb = (torch.randn(V, E + 4, device=dev) * 0.5).bfloat16()
    dout = torch.randn(M, E + 4, device=dev).bfloat16()
    rowdot = torch.randn(M, device=dev)
    ct = torch.empty(V, M, device=dev, dtype=torch.bfloat16)
    sb = dict(logits=logits, lse=lse, rowdot=rowdot, target=tgt, inv_tau=1.0, ce_scale=0.01)
    for tile in (192, 256):
        t_plain = timed(lambda: ops.gemm_bb(dout, emb, logits, force_tile=tile))
        t_smb = timed(lambda: ops.gemm_bb(dout, emb, logits, force_tile=tile, softmax_bwd=sb))
        t_ct = timed(lambda: ops.gemm_bb(dout, emb, logits, force_tile=tile, softmax_bwd=sb, c_t=ct))
        gb = M * V * 2 / 1e9
        say("M=%d V=%d K=%d tile %d: plain bf16 result %7.1f us (%.2f TB/s of C) | softmax-bwd %7.1f us | + transposed copy %7.1f us "
            "(%.2f TB/s over 3 passes)" % (M, V, E + 4, tile, t_plain, gb / t_plain * 1e3, t_smb, t_ct, 3 * gb / t_ct * 1e3))


case(1344, 20000, 300)
case(10496, 50000, 300)
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
