import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
dev = torch.device("cuda", 0)
def timed(fn, reps=10):
    best = 1e30
    for _ in range(3):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best
for M, N, K in ((4096, 304, 10496), (10496, 304, 4096), (10240, 304, 4096), (1344, 304, 2048), (2048, 304, 1344), (4096, 304, 10240), (4096, 384, 10240), (10496, 256, 4096)):
    a = torch.randn(M, K, device=dev).bfloat16(); b = torch.randn(N, K, device=dev).bfloat16(); o = torch.empty(M, N, device=dev)
    r = []
    for ft in (0, 384):
        for sk in ((0,) if ft == 0 else (0, 2, 3, 4, 6, 8)):
            try:
                r.append("t%d/sk%d %.1f" % (ft, sk, timed(lambda: ops.gemm_bb(a, b, o, force_tile=ft, force_splitk=sk))))
            except Exception as e:
                r.append("t%d/sk%d err" % (ft, sk))
    print("M %5d N %3d K %5d: %s" % (M, N, K, "  ".join(r)))
