"""GPU: the path that bench.py TIMES, at the sizes it is timed at.

bench.py measures `GraphedWakeSleep` - one hipGraph per branch, two side streams, parameter-gradient work deferred
behind the step's last LSTM launch (hip/step.py `later()` / `flush()`, `exclusive()` joins), per-stream split-K
workspaces. The oracle / golden parity tests run the eager step (they pass `capture=`, which switches the deferral off).
Here the replayed schedule is held to the eager step from identical initial weights over 41 training steps on fresh
batches (one warm-up step inside the constructor + 40 replays: past the ~30 replays after which an earlier replay-only
fault showed), at BASELINE configs[1] (B=64, V=20k, H=512) and at configs[3] shapes (B=256, V=50k, H=1024, L=40), in
fp32 and in the bf16 mode: every loss of every step to 1e-6, every weight after the last step to 2e-6 of its tensor's
largest entry (reference: model/cgn.py:1049-1154 is ONE schedule; the bar is summation order plus Adam's bias
correction formed in double on the device instead of on the host), no persistent LSTM launch timed out.
"""
import os

import pytest
import torch

import bench

pytestmark = pytest.mark.gpu

N_STEPS = 41
N_BATCHES = 4  # fresh batches, cycled


def _build(dev, workload):
    import ldrs_amd  # noqa: F401
    from ldrs_amd.model.cgn import Controlled_Generation_Sentence
    from ldrs_amd.utils import synthetic
    from ldrs_amd.utils.config import Config
    cfgd, B, L = bench.WORKLOADS[workload]
    cfg = bench.make_config(Config, cfgd)
    torch.manual_seed(1234)
    model = Controlled_Generation_Sentence(cfg, synthetic.synthetic_word_embed(cfg)).to(dev)
    eng = model.step_engine()
    opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
    return model, eng, opts, cfg, B, L


def _schedule(i):
    """(kld_coef, temperature) of training step i: both move from step to step, as utils/functional.py:28-42 does."""
    return 0.30 + 0.015 * i, 0.98 - 0.004 * i


def _losses(dl, c, eg):
    return [float(x) for x in dl] + [float(x) for x in eg[:4]] + [float(c.sum())]


@pytest.mark.parametrize("workload,compute", [("cfg2", "f32"), ("cfg2", "bf16"), ("cfg4", "f32"), ("cfg4", "bf16")])
def test_replayed_step_at_benchmark_size_trains_like_eager(dev, workload, compute):
    from ldrs_amd.hip import ops
    from ldrs_amd.hip.graph import GraphedWakeSleep
    from ldrs_amd.utils import synthetic
    prev = ops.set_compute(compute)
    try:
        m_e, eng_e, opts_e, cfg, B, L = _build(dev, workload)
        m_g, eng_g, opts_g, _, _, _ = _build(dev, workload)
        for (n, p), (_, q) in zip(m_e.named_parameters(), m_g.named_parameters()):
            assert torch.equal(p, q), n
        data = [bench.device_data(synthetic, cfg, B, L, 500 + i, dev)[0] for i in range(N_BATCHES)]

        eager = []
        for i in range(N_STEPS):
            kc, tau = _schedule(i)
            eager.append(_losses(*eng_e.wake_sleep(opts_e, data[i % N_BATCHES], kc, tau, calc_z_loss=False)))
        torch.cuda.synchronize()

        kc, tau = _schedule(0)  # the constructor's warm-up IS training step 0
        g = GraphedWakeSleep(eng_g, opts_g, data[0], calc_z_loss=False, warmup_kld=kc, warmup_tau=tau)
        assert g.branches is not None and len(g.branches) > 3, "bench.py's schedule = one graph per branch"
        replayed = []
        for i in range(1, N_STEPS):
            kc, tau = _schedule(i)
            dl, c, eg = g(kc, tau, data[i % N_BATCHES], totals=False)
            replayed.append(_losses(dl, c, eg))  # float() waits for the replay: every step's losses are read
        torch.cuda.synchronize()
        assert ops.lstm_sticky_timeouts() == 0

        worst_loss = 0.0
        for i, (le, lg) in enumerate(zip(eager[1:], replayed), 1):
            for a, b in zip(le, lg):
                assert b == b, ("NaN in replay", i, lg)
                worst_loss = max(worst_loss, abs(a - b) / max(1.0, abs(a)))
                assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (i, le, lg)
        assert all(st.step == N_STEPS for st in opts_g.values())
        assert all(int(st.step_dev.item()) == N_STEPS for st in opts_g.values())
        worst = ("", 0.0)
        for (n, p), (_, q) in zip(m_e.named_parameters(), m_g.named_parameters()):
            err = (p - q).abs().max().item() / max(p.abs().max().item(), 1e-12)
            worst = max(worst, (n, err), key=lambda t: t[1])
        line = ("replay vs eager, %s %s, %d steps (B=%d): worst loss difference %.2e, worst weight difference %.2e (%s)"
                % (workload, compute, N_STEPS, B, worst_loss, worst[1], worst[0]))
        print(line)
        d = os.environ.get("CGN_PARITY_REPORT")
        if d:
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "replay_vs_eager_%s_%s.txt" % (workload, compute)), "w") as f:
                f.write(line + "\n")
        assert worst[1] <= 2e-6, worst
    finally:
        ops.set_compute(prev)


@pytest.mark.parametrize("workload", ["cfg2", "cfg4"])
def test_unsynchronised_replays_at_benchmark_size(dev, workload):
    """bench.py's timed region issues its replays back to back without reading anything between them: 40 replays of
    ONE batch with no host synchronisation must end where 40 eager steps end."""
    from ldrs_amd.hip import ops
    from ldrs_amd.hip.graph import GraphedWakeSleep
    from ldrs_amd.utils import synthetic
    m_e, eng_e, opts_e, cfg, B, L = _build(dev, workload)
    m_g, eng_g, opts_g, _, _, _ = _build(dev, workload)
    d = bench.device_data(synthetic, cfg, B, L, 77, dev)[0]
    for _ in range(N_STEPS):
        _, _, eg_e = eng_e.wake_sleep(opts_e, d, 0.88, 0.9)
    g = GraphedWakeSleep(eng_g, opts_g, d, warmup_kld=0.88, warmup_tau=0.9)
    for _ in range(N_STEPS - 1):
        _, _, eg_g = g(0.88, 0.9, totals=False)
    torch.cuda.synchronize()
    assert ops.lstm_sticky_timeouts() == 0
    for a, b in zip(eg_e[:4], eg_g[:4]):
        a, b = float(a), float(b)
        assert b == b and abs(a - b) <= 1e-6 * max(1.0, abs(a)), (a, b)
    for (n, p), (_, q) in zip(m_e.named_parameters(), m_g.named_parameters()):
        err = (p - q).abs().max().item() / max(p.abs().max().item(), 1e-12)
        assert err <= 2e-6, (n, err)
