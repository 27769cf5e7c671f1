"""Diagnostic: the one-read softmax / NLL kernel over bf16-stored logits at configs[1] and configs[3] sizes: time, TB/s,
and its lse against fp64 on the same bf16 values."""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
for M, V in ((1344, 20000), (10496, 50000)):
    lg = (torch.randn(M, V, device=dev) * 3).bfloat16()
    tg = torch.randint(0, V, (M,), device=dev)
    fn = lambda: ops.softmax_nll_fwd(lg, tg, 1.0 / 0.9)
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lse, nll, loss = fn()
    e1.record()
    torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    ref = torch.logsumexp(lg[:256].double() / 0.9, 1)
    err = ((lse[:256].double() - ref).abs() / ref.abs()).max().item()
    print("M %5d V %5d: %7.1f us  %.2f TB/s  lse vs fp64 %.1e" % (M, V, us, M * V * 2 / us / 1e6, err))
