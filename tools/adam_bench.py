import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ldrs_amd
from ldrs_amd.hip import ops
dev = torch.device("cuda", 0)
n = 14_300_000
def mk(off):
    return [torch.randn(n + 8, device=dev)[off:off + n] for _ in range(4)]
for off, name in ((0, "16-byte form"), (1, "scalar form")):
    p, g, m, v = mk(off)
    v.abs_()
    for _ in range(5): ops.adam(p, g, m, v, 5e-5, 3)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.adam(p, g, m, v, 5e-5, 3)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print("adam %s: %.1f us, %.2f TB/s" % (name, us, 28.0 * n / us / 1e6))
