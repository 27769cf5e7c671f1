"""Diagnostic: which parameters differ between eager training and graph-replayed training (tests/test_graph_gpu.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import test_graph_gpu as T
from ldrs_amd.hip.graph import GraphedWakeSleep
dev = torch.device("cuda", 0)
nsteps = int(os.environ.get("NSTEPS", "1"))
m_e, eng_e, opts_e, ocfg = T._build(dev)
m_g, eng_g, opts_g, _ = T._build(dev)
data = T._batches(ocfg, dev, len(T.SCHED))
for (kc, tau), d in list(zip(T.SCHED, data))[:nsteps]:
    eng_e.wake_sleep(opts_e, d, kc, tau, calc_z_loss=False)
g = GraphedWakeSleep(eng_g, opts_g, data[0], calc_z_loss=False, overlap=True, warmup_kld=T.SCHED[0][0], warmup_tau=T.SCHED[0][1])
for (kc, tau), d in list(zip(T.SCHED[1:], data[1:]))[:nsteps - 1]:
    g(kc, tau, d)
torch.cuda.synchronize()
for (n, p), (_, q) in zip(m_e.named_parameters(), m_g.named_parameters()):
    err = (p - q).abs().max().item() / max(p.abs().max().item(), 1e-12)
    if err > 2e-6:
        bad = ((p - q).abs() > 1e-7).sum().item()
        print("%-50s err %.2e  differing elements %d of %d" % (n, err, bad, p.numel()))
print("done")
