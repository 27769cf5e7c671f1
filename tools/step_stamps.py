#!/usr/bin/env python3
"""Un-profiled timeline of the replayed wake-sleep step: the step is captured with phase stamps in it (hip/ops.py STAMPS:
1-thread kernels that write the device wall clock when their stream reaches them), replayed back to back like bench.py's
timed region, and the stamps of the LAST replays are printed per stream. Settles what rocprofv3's kernel trace cannot
(the profiler serialises graph launches differently): where the main chain waits, when the side graphs really start.
usage: tools/step_stamps.py [--workload cfg2|cfg4] [--compute f32|bf16] [--replays N] [--out PATH]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
from ldrs_amd.hip.graph import GraphedWakeSleep
from ldrs_amd.model.cgn import Controlled_Generation_Sentence
from ldrs_amd.utils import synthetic
from ldrs_amd.utils.config import Config

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="cfg2")
ap.add_argument("--compute", default="f32")
ap.add_argument("--replays", type=int, default=30)
ap.add_argument("--out")
ap.add_argument("--force-dp", action="store_true", help="the data-parallel schedule on a one-rank RCCL group")
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfgd, B, L = bench.WORKLOADS[args.workload]
cfg = bench.make_config(Config, cfgd)
ops.set_compute(args.compute)
torch.manual_seed(1234)
model = Controlled_Generation_Sentence(cfg, synthetic.synthetic_word_embed(cfg)).cuda()
if args.force_dp:
    from ldrs_amd import parallel
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
    torch.cuda.set_device(0)
    torch.distributed.init_process_group(backend="nccl", rank=0, world_size=1)
    parallel.attach(model, force=True)
eng = model.step_engine()
opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
data, _ = bench.device_data(synthetic, cfg, B, L, 1234, dev)


class Stamps(dict):
    """names restart with every pass over the step (warm-up, capture): the captured pass's list is the one kept"""


buf = torch.zeros(128, dtype=torch.int64, device=dev)
ops.STAMPS = {"buf": buf, "names": []}
real_run = GraphedWakeSleep._run


def run(self):
    ops.STAMPS["names"] = []
    return real_run(self)


GraphedWakeSleep._run = run
g = GraphedWakeSleep(eng, opts, data, warmup_kld=0.88, warmup_tau=0.9)
names = list(ops.STAMPS["names"])
ops.STAMPS = None
for _ in range(5):
    g(0.88, 0.9, totals=False)
torch.cuda.synchronize()
rows = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
import time
e0.record()
h0 = time.perf_counter()
for i in range(args.replays):
    g(0.88, 0.9, totals=False)
host_ms = 1e3 * (time.perf_counter() - h0) / args.replays  # time the HOST needs to hand one step to the GPU
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / args.replays
t = buf[:len(names)].cpu().tolist()  # the last of the back-to-back replays (the host was far ahead of the GPU)
# host cost of each graph launch of one replay (one more, synchronised, replay with a clock around every launch)
real_replay = torch.cuda.CUDAGraph.replay
costs = []


def timed_replay(self):
    a = time.perf_counter()
    real_replay(self)
    costs.append(1e6 * (time.perf_counter() - a))


torch.cuda.CUDAGraph.replay = timed_replay
h0 = time.perf_counter()
g(0.88, 0.9, totals=False)
one_host_us = 1e6 * (time.perf_counter() - h0)
torch.cuda.CUDAGraph.replay = real_replay
torch.cuda.synchronize()
t0 = min(t)
lines = ["# un-profiled replay, %s %s: %d back-to-back replays, %.3f ms per step (stamp nodes included); stamps of the last"
         " replay, us since its first stamp; M = main stream, S0 / S1 = side streams" % (args.workload, args.compute,
                                                                                        args.replays, ms),
         "# host: %.3f ms per step to ISSUE the replay loop (no synchronisation); one replay alone: %.0f us on the host, "
         "of which %d graph launches: %s us" % (host_ms, one_host_us, len(costs), " ".join("%.0f" % c for c in costs))]
order = sorted(range(len(names)), key=lambda i: t[i])
last = {}
for i in order:
    us = (t[i] - t0) / 100.0  # wall_clock64: 100 MHz
    st = names[i].split()[0]
    lines.append("%9.1f us  (+%7.1f on %s)  %s" % (us, us - last.get(st, us), st, names[i]))
    last[st] = us
text = "\n".join(lines)
print(text)
if args.force_dp:
    torch.distributed.destroy_process_group()
if args.out:
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    open(args.out, "w").write(text + "\n")
