"""Diagnostic: the LSTM launches ALONE on the GPU: us per launch and per timestep.
usage: lstm_bench.py [B H [T]]   (default 64 512 = cfg2; 256 1024 40 = cfg4). CGN_LSTM_PERSISTENT=0 times the per-timestep kernels."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import ldrs_amd  # noqa
from ldrs_amd.hip import engine, ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


def seqs(n, B, T, H, revs):
    out = []
    for r in revs:
        s = engine.LstmSeq(B, T, H, dev, r)
        s.xg.normal_()
        s.set_initial()
        out.append(s)
    return out


if os.environ.get("CGN_LSTM_OPERAND_MODE"):  # 0 fp32 accuracy | 1 two-term | 2 plain bf16 operands (row-tiled kernels at H > 512)
    from ldrs_amd.hip import _lib
    _lib.check(_lib.load().cgn_lstm_set_operand_mode(int(os.environ["CGN_LSTM_OPERAND_MODE"])), "cgn_lstm_set_operand_mode")
    print("operand mode", os.environ["CGN_LSTM_OPERAND_MODE"])
B, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 512)
T0 = int(sys.argv[3]) if len(sys.argv) > 3 else 20
print("B %d H %d: sequences per persistent launch: encoder %d, generator layer %d" % (
    B, H, ops.lstm_persistent_group(2, B, T0, H, dev), ops.lstm_persistent_group(1, B, T0 + 1, H, dev)))
for name, T, revs in (("encoder biLSTM (2 sequences)", T0, (False, True)), ("one generator layer", T0 + 1, (False,))):
    ss = seqs(len(revs), B, T, H, revs)
    W = [torch.randn(4 * H, H, device=dev) / H ** 0.5 for _ in revs]
    us = timeit(lambda: engine.lstm_run_fwd(ss, W))
    print("%-32s fwd  %7.1f us  %5.2f us/step   (sticky %d)" % (name, us, us / T, ops.lstm_sticky_timeouts()))
    dh = torch.randn(B, T, H, device=dev)
    dcs = [torch.zeros(B, H, device=dev) for _ in revs]
    us = timeit(lambda: engine.lstm_run_bwd(ss, W, [(dh, T * H, H)] * len(revs), [None] * len(revs), dcs))
    print("%-32s bwd  %7.1f us  %5.2f us/step   (sticky %d)" % (name, us, us / T, ops.lstm_sticky_timeouts()))
T = T0 + 1
if not ops.lstm2_persistent_bytes(B, T, H, dev):
    print("two-layer wavefront form: shape not eligible; sticky timeouts:", ops.lstm_sticky_timeouts())
    sys.exit(0)
g0, g1 = seqs(2, B, T, H, (False, False))
Ws = [torch.randn(4 * H, H, device=dev) / H ** 0.5 for _ in range(3)]
bs = [torch.randn(4 * H, device=dev) * 0.1 for _ in range(2)]
us = timeit(lambda: ops.lstm2_fwd(g0.xg, None, Ws[0], Ws[1], Ws[2], bs[0], bs[1], g0.hbuf, g0.cbuf, g0.gates, g1.hbuf,
                                  g1.cbuf, g1.gates, B, T, H))
print("%-32s fwd  %7.1f us  %5.2f us/step" % ("generator, 2 layers fused", us, us / T))
dh = torch.randn(B, T, H, device=dev)
dg0, dg1 = torch.empty(B, T, 4 * H, device=dev), torch.empty(B, T, 4 * H, device=dev)
us = timeit(lambda: ops.lstm2_bwd(Ws[0], Ws[1], Ws[2], g0.cbuf, g0.gates, g1.cbuf, g1.gates, dh, T * H, H, dg0, dg1, B, T, H))
print("%-32s bwd  %7.1f us  %5.2f us/step" % ("generator, 2 layers fused", us, us / T))
print("sticky timeouts:", ops.lstm_sticky_timeouts())
