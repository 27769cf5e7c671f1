#!/usr/bin/env python3
"""Diagnostic: which operands the bf16 mode casts in one wake-sleep step (hip/ops.py cast_bf16), in launch order, with the
caller that asked and whether the same source was cast before in the step — the list the cast-sharing work is read from.
usage: tools/bb_cast_census.py [--workload cfg2|cfg4] [--out PATH]"""
import argparse
import os
import sys
import traceback

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
ap.add_argument("--workload", default="cfg4")
ap.add_argument("--out")
args = ap.parse_args()
dev = torch.device("cuda", 0)
cfgd, B, L = bench.WORKLOADS[args.workload]
cfg = bench.make_config(Config, cfgd)
ops.set_compute("bf16")
torch.manual_seed(1234)
model = Controlled_Generation_Sentence(cfg, synthetic.synthetic_word_embed(cfg)).cuda()
eng = model.step_engine()
opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
data, _ = bench.device_data(synthetic, cfg, B, L, 1234, dev)
log = []
real_cast = ops.cast_bf16
real_run = GraphedWakeSleep._run


def cast(src, ax, rows, K, transpose=False, **kw):
    fr = [f for f in traceback.extract_stack()[:-1] if "/hip/" in f.filename or "/model/" in f.filename]
    who = " < ".join("%s:%d" % (f.name, f.lineno) for f in reversed(fr[-5:]) if f.name not in ("gemm", "_gemm_bb_route", "_bb_operand"))
    log.append((src.data_ptr(), ax.stride, ax.rpg, ax.gs, rows, K, bool(transpose), who))
    return real_cast(src, ax, rows, K, transpose=transpose, **kw)


def run(self):
    del log[:]
    return real_run(self)


ops.cast_bf16 = cast
GraphedWakeSleep._run = run
g = GraphedWakeSleep(eng, opts, data, warmup_kld=0.88, warmup_tau=0.9)
torch.cuda.synchronize()
seen = {}
lines = ["# casts of one %s step in the bf16 mode, in launch order (MB = fp32 bytes read)" % args.workload]
tot = rep = 0.0
for i, (p, st, rpg, gs, rows, K, tr, who) in enumerate(log):
    key = (p, st, rpg, gs, rows, K, tr)
    mb = rows * K * 4 / 1e6
    again = seen.get(key)
    seen.setdefault(key, i)
    tot += mb
    rep += mb if again is not None else 0.0
    lines.append("%3d %s rows %6d K %6d %7.1f MB %s %s" % (i, "T" if tr else "-", rows, K, mb,
                                                          ("again(%d)" % again) if again is not None else "         ", who))
lines.append("# %d casts, %.0f MB read; %.0f MB of it a source cast before in the step" % (len(log), tot, rep))
print("\n".join(lines))
if args.out:
    open(args.out, "w").write("\n".join(lines) + "\n")
