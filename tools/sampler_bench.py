"""Diagnostic: wall time of the sleep-phase beam sampler at the benchmark model size (cfg2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ldrs_amd  # noqa
from ldrs_amd.hip.sampler import BeamSampler
from ldrs_amd.model.cgn import Controlled_Generation_Sentence
from ldrs_amd.utils.config import Config
from ldrs_amd.utils import synthetic

dev = torch.device("cuda", 0)
cfg = bench.make_config(Config)
torch.manual_seed(1234)
model = Controlled_Generation_Sentence(cfg, synthetic.synthetic_word_embed(cfg)).cuda()
eng = model.step_engine()
V = cfg.word_vocab_size
for S, K in ((10, 10), (64, 10), (64, 1)):
    seed = torch.cat([torch.randn(S, cfg.latent_variable_size), torch.bernoulli(torch.full((S, 1), 0.5))], 1).to(dev)
    smp = BeamSampler(eng)
    smp.sample(seed, 16, K, V - 2, V - 1, V - 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        smp.sample(seed, 16, K, V - 2, V - 1, V - 3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("S=%3d K=%2d T=16: %.2f ms per call, %.1f us per token step, %.0f sentences/s" % (S, K, dt * 1e3, dt * 1e6 / 16, S / dt))
