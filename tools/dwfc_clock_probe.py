"""Diagnostic: is the slow mode of dW_fc inside the step a clock state? Runs the eager step of bench.py's workload and
brackets the dW_fc product with a one-wave kernel that measures the shader clock (tools/experiments/clock_sample.hip):
prints, per step, dW_fc's duration and the clock right before and right after it."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
from ldrs_amd.model.cgn import Controlled_Generation_Sentence
from ldrs_amd.utils import synthetic
from ldrs_amd.utils.config import Config
from ldrs_amd.utils.functional import kld_coef, temp_coef

lib = C.CDLL(os.path.join(ROOT, "tools", "experiments", "probe_clock_sample.so"))
lib.clock_sample.argtypes = [C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
cfgd, BATCH, SEQ = bench.WORKLOADS["cfg2"]
cfg = bench.make_config(Config, cfgd)
torch.manual_seed(1234)
model = Controlled_Generation_Sentence(cfg, synthetic.synthetic_word_embed(cfg)).cuda()
eng = model.step_engine()
opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
data, _ = bench.device_data(synthetic, cfg, BATCH, SEQ, 1234, dev)
kc, tau = kld_coef(5000), temp_coef(5000)
V = cfg.word_vocab_size
slots = torch.zeros(64, 4, dtype=torch.int64, device=dev)
records = []
PRE = sys.argv[1] if len(sys.argv) > 1 else ''
tmp_bias = torch.empty(V, device=dev)
extra = []
real_gemm = ops.gemm
n = [0]


def sample(i):
    lib.clock_sample(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(slots[i].data_ptr()))


def gemm(a, b, out, **kw):
    if not kw.get("_desc_only") and out.dim() == 2 and tuple(out.shape) == (V, cfg.decoder_rnn_size) and kw.get("K") == BATCH * (SEQ + 1):
        i = n[0] % 32
        if n[0] % 2 == 1:  # every other step: touch one operand right before the product
            if PRE == "colsum":
                ops.colsum(a.t(), tmp_bias)
            elif PRE == "fillG":
                ops.fill(out, 0.0)
            elif PRE == "readB":
                b.sum()
            elif PRE == "sync":
                torch.cuda.synchronize()
            elif PRE == "small":  # the same kernel variant on a small problem (instruction cache)
                real_gemm(a[:256], b, out[:256], **kw)
        sample(2 * i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = real_gemm(a, b, out, **kw)
        e1.record()
        sample(2 * i + 1)
        e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        real_gemm(a, b, out, **kw)  # same product again, back to back
        e2.record()
        real_gemm(a, b, out, **kw)
        e3.record()
        extra.append((e1, e2, e3))
        records.append((i, e0, e1))
        n[0] += 1
        return r
    return real_gemm(a, b, out, **kw)


ops.gemm = gemm
for step in range(30):
    eng.wake_sleep(opts, data, kc, tau)
torch.cuda.synchronize()
s = slots.cpu()
for i, e0, e1 in records[-24:]:
    mhz = lambda k: 100.0 * float(s[k, 0]) / max(float(s[k, 1]), 1.0)  # noqa: E731
    x1, x2, x3 = extra[records.index((i, e0, e1))]
    print("dW_fc %6.1f us   clock before %5.0f MHz   after %5.0f MHz   | repeated at once: %6.1f us, again: %6.1f us" % (
        e0.elapsed_time(e1) * 1e3, mhz(2 * i), mhz(2 * i + 1), x1.elapsed_time(x2) * 1e3, x2.elapsed_time(x3) * 1e3))
