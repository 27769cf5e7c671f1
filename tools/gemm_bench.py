"""Diagnostic: GEMM kernel throughput on chosen shapes (NT unless noted), A/B interleaved in one process."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)

def bench(M, N, K, layout="nt", tile=0, splitk=0, bias=False, reps=20):
    a = torch.rand(M, K, device=dev) * 2 - 1
    b = torch.rand(N, K, device=dev) * 2 - 1
    A = a if layout[0] == "n" else a.t().contiguous().t()
    Bm = b.t() if layout[1] == "t" else b.t().contiguous()
    out = torch.empty(M, N, device=dev)
    bz = torch.rand(N, device=dev) if bias else None
    us = float("inf")
    for _ in range(3):  # min of 3 timed batches, each behind ~10 ms of the same work (clocks ramp down between cases)
        for _ in range(40):
            ops.gemm(A, Bm, out, bias0=bz, force_tile=tile, force_splitk=splitk)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm(A, Bm, out, bias0=bz, force_tile=tile, force_splitk=splitk)
        e1.record()
        torch.cuda.synchronize()
        us = min(us, 1e3 * e0.elapsed_time(e1) / reps)
    print("%-3s M=%6d N=%6d K=%6d tile=%3d sk=%2d bias=%d : %8.1f us %7.1f TFLOP/s" % (
        layout, M, N, K, tile, splitk, bias, us, 2.0 * M * N * K / us / 1e6), flush=True)

cases = [eval(a) for a in sys.argv[1:]] or [
    (4096, 4096, 4096, "nt", 128), (4096, 4096, 4096, "nt", 64),
    (1344, 20000, 512, "nt", 128), (1344, 20000, 512, "nt", 64), (1280, 20000, 512, "nt", 128),
    (1344, 20000, 512, "nt", 128, 0, True),
    (1280, 832, 832, "nt", 64), (1280, 832, 832, "nt", 64, 2), (1280, 832, 832, "nt", 64, 4), (1280, 832, 832, "nt", 128, 4),
    (1280, 825, 825, "nt", 64),
    (2048, 512, 1344, "tn", 64), (2048, 512, 1344, "tn", 64, 4), (2048, 512, 1344, "tn", 128, 8),
    (1344, 512, 20000, "nn", 64, 5), (1344, 512, 20000, "nn", 128, 16), (1344, 512, 20000, "nn", 128, 24),
    (20000, 512, 1344, "tn", 128), (20000, 512, 1344, "tn", 128, 2),
]
from ldrs_amd.hip import _lib
lib = _lib.load()
for mode in os.environ.get("GEMM_COMPUTE", "f32").split(","):
  ops.set_compute(mode)
  print("compute", mode)
  for st in [int(x) for x in os.environ.get('GEMM_STAGES', '2').split(',')]:
    for order in [int(x) for x in os.environ.get('GEMM_ORDER', '2').split(',')]:
        lib.cgn_debug_set_stages(st | (order << 4))
        print("stages", st, "order", order)
        for c in cases:
            bench(*c)
lib.cgn_debug_set_stages(0)
