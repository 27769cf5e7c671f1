"""Diagnostic: the X6 product (three-term bf16 split, six partial bf16 MFMA products, fp32 accumulation) beside the fp32
MFMA product on the step's shapes: time of both, and the error of both against an fp64 product of the same operands.
usage: gemm_x6_probe.py [out.txt]"""
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


def operands(M, N, K, layout):
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    A = a if layout[0] == "n" else a.t().contiguous().t()
    Bm = b.t() if layout[1] == "t" else b.t().contiguous()
    return A, Bm


def timed(fn, reps=20):
    us = float("inf")
    for _ in range(3):
        for _ in range(20):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = min(us, 1e3 * e0.elapsed_time(e1) / reps)
    return us


def case(M, N, K, layout, tile=0, splitk=0):
    A, Bm = operands(M, N, K, layout)
    ref = A.double() @ Bm.double()
    res = {}
    for mode in ("f32_mfma", "f32x6"):
        out = torch.empty(M, N, device=dev)
        fn = lambda: ops.gemm(A, Bm, out, force_tile=tile, force_splitk=splitk, compute=mode)
        fn()
        torch.cuda.synchronize()
        err = ((out.double() - ref).norm() / ref.norm()).item()
        # worst element against the size of its own dot product's terms (|a|.|b|): the bound fp32 accumulation obeys
        scale = (A.abs().double() @ Bm.abs().double())
        worst = ((out.double() - ref).abs() / scale).max().item()
        res[mode] = (timed(fn), err, worst)
    t0, t1 = res["f32_mfma"][0], res["f32x6"][0]
    say("%-2s M=%6d N=%6d K=%6d tile=%3d sk=%2d | fp32 MFMA %7.1f us %6.1f TF/s  rel %.2e worst %.2e | X6 %7.1f us %6.1f TF/s  "
        "rel %.2e worst %.2e | x%.2f" % (layout, M, N, K, tile, splitk, t0, 2.0 * M * N * K / t0 / 1e6, res["f32_mfma"][1],
                                        res["f32_mfma"][2], t1, 2.0 * M * N * K / t1 / 1e6, res["f32x6"][1],
                                        res["f32x6"][2], t0 / t1))


cases = [
    ("nt", 1344, 20000, 512), ("nt", 1344, 20000, 300), ("nn", 1344, 512, 20000), ("nn", 1344, 300, 20000),
    ("tn", 20000, 512, 1344), ("nt", 1280, 2484, 825), ("nn", 1280, 825, 2484), ("tn", 2484, 825, 1280),
    ("nn", 1280, 825, 2048), ("tn", 2048, 825, 1280), ("tn", 2048, 512, 1344), ("nt", 1280, 2048, 825),
    ("tn", 2048, 300, 1344), ("nt", 1344, 2048, 300), ("nt", 4096, 4096, 4096), ("nn", 4096, 4096, 4096),
    ("tn", 4096, 4096, 4096), ("nt", 1344, 20000, 512, 64), ("nt", 10240, 50000, 1024), ("tn", 50000, 1024, 10240),
]
for c in cases:
    case(c[1], c[2], c[3], c[0], *(c[4:]))
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
