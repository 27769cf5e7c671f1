"""Which torch streams really run side by side on this ROCm stack? HIP multiplexes streams onto GPU_MAX_HW_QUEUES (4)
hardware queues; two streams that share one run their kernels one after the other. Probe: a one-block spin kernel
(torch.cuda._sleep) on stream a and on stream b; wall time ~1x = concurrent, ~2x = serialized."""
import time
import torch

dev = torch.device("cuda:0")
torch.cuda._sleep(1000)
torch.cuda.synchronize()
null = torch.cuda.default_stream(dev)
streams = [("null", null)] + [("s%d" % i, torch.cuda.Stream(dev)) for i in range(9)]
CYC = 2_000_000


def run(pair):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _, st in pair:
        with torch.cuda.stream(st):
            torch.cuda._sleep(CYC)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    one = run(streams[:1])
print("one spin: %.3f ms" % one)
names = [n for n, _ in streams]
print("      " + " ".join("%5s" % n for n in names))
for i, a in enumerate(streams):
    row = []
    for j, b in enumerate(streams):
        row.append("  -  " if i == j else "%5.2f" % (run([a, b]) / one))
    print("%5s " % a[0] + " ".join(row))
print("three at once (s0,s1,s2): %.2f" % (run(streams[1:4]) / one))
print("four at once (null,s0,s1,s2): %.2f" % (run(streams[0:4]) / one))
print("five at once: %.2f" % (run(streams[0:5]) / one))
