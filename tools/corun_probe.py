"""Diagnostic: does a 64-tile GEMM chain on one stream really run beside the 128-tile dW_fc product on another?
Eager launches on two torch streams, timed with events per stream and in total."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
V, H, NT = 20000, 512, 1344
dlog = torch.randn(NT, V, device=dev)
h1 = torch.randn(NT, H, device=dev)
dW = torch.empty(V, H, device=dev)
a = torch.randn(1280, 2048, device=dev)
w = torch.randn(2048, 828, device=dev)
o = torch.empty(1280, 828, device=dev)


def big():
    ops.gemm(dlog.t(), h1, dW)


def small(n=6):
    for _ in range(n):
        ops.gemm(a, w, o)


sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
for st in (sa, sb):       # per-stream workspaces, module load
    with torch.cuda.stream(st):
        big(), small()
torch.cuda.synchronize()


def timed(fa, fb, reps=10):
    tot = []
    for _ in range(reps):
        torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
        e[0].record()
        sa.wait_event(e[0]); sb.wait_event(e[0])
        with torch.cuda.stream(sa):
            e[1].record()
            if fa: fa()
            e[2].record()
        with torch.cuda.stream(sb):
            e[3].record()
            if fb: fb()
            e[4].record()
        torch.cuda.current_stream().wait_event(e[2]); torch.cuda.current_stream().wait_event(e[4])
        e[5].record()
        torch.cuda.synchronize()
        tot.append((e[0].elapsed_time(e[5]), e[0].elapsed_time(e[1]), e[0].elapsed_time(e[2]), e[0].elapsed_time(e[3]),
                    e[0].elapsed_time(e[4])))
    tot.sort()
    return tot[len(tot) // 2]


for name, fa, fb in (("big alone", big, None), ("small x6 alone", None, small), ("big || small x6", big, small),
                     ("small x6 || small x6", small, small), ("big || big", big, big)):
    t = timed(fa, fb)
    print("%-22s total %7.1f us | A %6.1f -> %6.1f | B %6.1f -> %6.1f" % ((name,) + tuple(1e3 * x for x in t)))
