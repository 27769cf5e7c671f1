"""Diagnostic: schedule sweep of the step's GEMM shapes. For every shape and tile: one workgroup per tile (DP), every tile
cut into s ranges with the in-kernel fix-up (s = 2..8), and stream-K over all resident workgroups. us per launch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops, _lib

dev = torch.device("cuda", 0)
torch.manual_seed(0)
lib = _lib.load()


def operands(M, N, K, layout):
    a = torch.rand(M, K, device=dev) * 2 - 1
    b = torch.rand(N, K, device=dev) * 2 - 1
    A = a if layout[0] == "n" else a.t().contiguous().t()
    Bm = b.t() if layout[1] == "t" else b.t().contiguous()
    return A, Bm, torch.empty(M, N, device=dev)


def run(A, Bm, out, tile, sk, reps=30):
    """min of 3 timed batches, each behind ~10 ms of the same work (clocks ramp down between cases otherwise)"""
    best = float("inf")
    try:
        for _ in range(3):
            for _ in range(40):
                ops.gemm(A, Bm, out, force_tile=tile, force_splitk=sk)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.gemm(A, Bm, out, force_tile=tile, force_splitk=sk)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    except RuntimeError:
        return float("nan")
    return best


SHAPES = [(1344, 20000, 512, "nt"), (20000, 512, 1344, "tn"), (1344, 512, 20000, "nn"), (1344, 300, 20000, "nn"),
          (1344, 20000, 300, "nt"), (1280, 2484, 828, "nt"), (1280, 828, 2484, "nn"), (2484, 828, 1280, "tn"),
          (1280, 828, 2048, "nn"), (2048, 512, 1344, "tn"), (2048, 828, 1280, "tn"), (1280, 2048, 828, "nt"),
          (2048, 300, 1344, "tn"), (1344, 2048, 300, "nt"), (1344, 2048, 512, "nt"), (2048, 512, 1280, "tn")]
SPLITS = [int(x) for x in os.environ.get("SWEEP_SPLITS", "2,3,4,6,8,12,16").split(",")]
print("%-28s %4s | %7s | %s | %7s" % ("shape", "tile", "DP", " ".join("s=%-5d" % s for s in SPLITS), "streamK"))
for M, N, K, lay in SHAPES:
    A, Bm, out = operands(M, N, K, lay)
    for tile in (64, 128):
        lib.cgn_debug_set_stages(2 | (2 << 4) | (1 << 8))
        dp = run(A, Bm, out, tile, 0)
        sp = []
        for s in SPLITS:
            sp.append(run(A, Bm, out, tile, s) if K // s >= 128 else float("nan"))
        lib.cgn_debug_set_stages(2 | (2 << 4) | (2 << 8))
        skt = run(A, Bm, out, tile, 0)
        gf = 2.0 * M * N * K / 1e6
        best = min([dp, skt] + [x for x in sp if x == x])
        print("%-28s %4d | %7.1f | %s | %7.1f   best %6.1f us = %5.1f TF" % (
            "%s %dx%dx%d" % (lay, M, N, K), tile, dp, " ".join("%-7.1f" % x for x in sp), skt, best, gf / best), flush=True)
lib.cgn_debug_set_stages(0)
