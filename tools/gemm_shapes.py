"""Diagnostic: per-GEMM-shape timing of one cfg2 wake-sleep step (HIP events around every cgn_gemm_f32 launch)."""
import collections
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
from ldrs_amd.model.cgn import Controlled_Generation_Sentence
from ldrs_amd.utils.config import Config
from ldrs_amd.utils.functional import kld_coef, temp_coef
from ldrs_amd.utils import synthetic

dev = torch.device("cuda", 0)
cfg = bench.make_config(Config)
torch.manual_seed(1234)
model = Controlled_Generation_Sentence(cfg, synthetic.synthetic_word_embed(cfg)).cuda()
eng = model.step_engine()
opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
data, _ = bench.device_data(synthetic, cfg, bench.BATCH, bench.SEQ, 1234, dev)
zl = "--z-loss" in sys.argv

def step():
    eng.disc_train(opts["disc"], data["x_labelled"], data["y_labelled"], data["x_generated"], data["c_generated"],
                   data["mask_labelled"], data["mask_generated"])
    c, _ = eng.infer_c(data["enc_word"], data["c_uniform"])
    eng.enc_gen_train((opts["enc"], opts["gen"]), data["enc_word"], data["enc_char"], data["dec_word"], data["target"],
                      c, data["eps"], kld_coef(5000), temp_coef(5000), calc_z_loss=zl)

for _ in range(3):
    step()
torch.cuda.synchronize()
ops.GEMM_PROFILE = {"variant": None, "records": []}
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize()
recs = ops.GEMM_PROFILE["records"]
ops.GEMM_PROFILE = None
agg = collections.OrderedDict()
for key, sk, M, N, K, e0, e1 in recs:
    k = (key, sk, M, N, K)
    a = agg.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += e0.elapsed_time(e1)
tot = 0
print("%-16s %3s %7s %7s %7s %6s %9s %8s" % ("variant", "sk", "M", "N", "K", "calls", "us/call", "TFLOP/s"))
for (key, sk, M, N, K), (cnt, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    us = 1e3 * ms / cnt
    tf = 2.0 * M * N * K / (us * 1e-6) / 1e12 if us > 0 else 0
    tot += ms / n
    print("%-16s %3d %7d %7d %7d %6.1f %9.1f %8.1f" % (key, sk, M, N, K, cnt / n, us, tf))
print("total GEMM ms/step (event-bracketed):", tot)
