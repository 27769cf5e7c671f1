"""Diagnostic: cfg2 step with the per-timestep LSTM kernels: eager vs hipGraph replay (branch graphs / one graph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import ldrs_amd  # noqa
from ldrs_amd.hip import ops
from ldrs_amd.hip.graph import GraphedWakeSleep
from ldrs_amd.model.cgn import Controlled_Generation_Sentence
from ldrs_amd.utils.config import Config
from ldrs_amd.utils import synthetic

ops.LSTM_PERSISTENT = os.environ.get("NP", "1") != "1"
dev = torch.device("cuda", 0)
KC, TAU = float(os.environ.get("KC", "0.5")), float(os.environ.get("TAU", "0.9"))
cfg = bench.make_config(Config, bench.CFG2)


def build():
    torch.manual_seed(1234)
    model = Controlled_Generation_Sentence(cfg, synthetic.synthetic_word_embed(cfg)).cuda()
    eng = model.step_engine()
    opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
    return model, eng, opts


data, _ = bench.device_data(synthetic, cfg, 64, 20, 1234, dev)
for mode in ("eager", "branch", "single", "nooverlap"):
    model, eng, opts = build()
    if mode == "eager":
        outs = [eng.wake_sleep(opts, data, KC, TAU) for _ in range(61)]
        print(mode, [float(o[2][0]) for o in outs[:2]], float(outs[-1][2][0]), "first nan at", next((i for i, o in enumerate(outs) if float(o[2][0]) != float(o[2][0])), None), flush=True)
    else:
        g = GraphedWakeSleep(eng, opts, data, warmup_kld=KC, warmup_tau=TAU, branch_graphs=(mode == "branch"), overlap=(mode != "nooverlap"))
        vals = []
        sync = os.environ.get("SYNC", "0") == "1"
        for it in range(60):
            d, c, eg = g(KC, TAU, totals=False)
            if it == 4 and os.environ.get("MIDSYNC") == "1":
                torch.cuda.synchronize()
            if sync or (os.environ.get("MIDSYNC") == "1" and it > 4 and it % int(os.environ.get("EVERY", "1")) == 0):
                torch.cuda.synchronize()
                vals.append(float(eg[0]))
        torch.cuda.synchronize()
        vals.append(float(eg[0]))
        bad = next((i for i, v in enumerate(vals) if v != v), None)
        print(mode, "sync" if sync else "nosync", vals[:2], vals[-1], "first nan at", bad, flush=True)
