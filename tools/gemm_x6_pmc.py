"""Runs ONE product shape a few times in one compute mode (for rocprofv3 --pmc passes).
usage: gemm_x6_pmc.py <layout> <M> <N> <K> <mode: f32_mfma|f32x6|bf16> [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops

layout, M, N, K, mode = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
dev = torch.device("cuda", 0)
torch.manual_seed(0)
a = torch.randn(M, K, device=dev)
b = torch.randn(N, K, device=dev)
A = a if layout[0] == "n" else a.t().contiguous().t()
Bm = b.t() if layout[1] == "t" else b.t().contiguous()
out = torch.empty(M, N, device=dev)
for _ in range(reps):
    ops.gemm(A, Bm, out, compute=mode)
torch.cuda.synchronize()
print("done", layout, M, N, K, mode)
