"""Diagnostic: the plain-bf16 recurrences (operand mode 2) at configs[3] shapes: sticky word after each launch.
usage: lstm_x1_diag.py <case>  (a: forward seq first thing in the process, b: reversed seq, c: two seqs back to back,
d: 30 unsynchronised forward launches)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import engine, ops

case = sys.argv[1]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, T, H = 256, 40, 1024
W = torch.randn(4 * H, H, device=dev) / H ** 0.5
ops.LSTM_BF16_RT = 2
ops.set_compute("bf16")


def mk(rev):
    s = engine.LstmSeq(B, T, H, dev, rev)
    s.xg.normal_()
    s.set_initial()
    return s


def report(tag):
    torch.cuda.synchronize()
    print(case, tag, "sticky", ops.lstm_sticky_timeouts(), flush=True)


if case == "a":
    s = mk(False)
    torch.cuda.synchronize()
    engine.lstm_run_fwd([s], [W]); report("first fwd")
    engine.lstm_run_fwd([s], [W]); report("second fwd")
elif case == "b":
    s = mk(True)
    torch.cuda.synchronize()
    engine.lstm_run_fwd([s], [W]); report("reversed fwd")
elif case == "c":
    s0, s1 = mk(False), mk(True)
    torch.cuda.synchronize()
    engine.lstm_run_fwd([s0, s1], [W, W]); report("two seqs")
    engine.lstm_run_fwd([s0, s0], [W, W]); report("two forward seqs")
elif case == "d":
    s = mk(False)
    torch.cuda.synchronize()
    for i in range(30):
        engine.lstm_run_fwd([s], [W])
    report("30 unsynchronised launches")
