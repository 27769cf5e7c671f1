"""GPU: the wake-sleep step replayed from a captured hipGraph (hip/graph.py: device-resident schedules and Adam step
counts, side-stream overlap) must train exactly like the eager step issued kernel by kernel."""
import pytest
import torch

from oracle import cgn_oracle as O

pytestmark = pytest.mark.gpu

CFG = dict(max_word_len=12, max_seq_len=10, word_vocab_size=1200, char_vocab_size=64, encoder_rnn_size=64,
           latent_variable_size=32, decoder_rnn_size=96, sentiment_kernel_num=20)
SCHED = [(0.31, 0.97), (0.52, 0.91), (0.88, 0.64), (0.93, 0.60)]  # (kld_coef, temperature) per step


def _build(dev):
    import ldrs_amd  # noqa: F401
    from ldrs_amd.model.cgn import Controlled_Generation_Sentence
    from ldrs_amd.utils.config import Config
    kw = dict(CFG)
    pos = [kw.pop(k) for k in ("max_word_len", "max_seq_len", "word_vocab_size", "char_vocab_size")]
    cfg = Config(*pos, 1e-3, 7.9, 0.1, 0.1, 1.0, **kw)
    ocfg = O.OracleConfig(**CFG)
    torch.manual_seed(3)
    model = Controlled_Generation_Sentence(cfg, O.synthetic_word_embed(ocfg)).to(dev)
    eng = model.step_engine()
    opts = dict(disc=eng.new_optimizer("disc"), enc=eng.new_optimizer("enc"), gen=eng.new_optimizer("gen"))
    return model, eng, opts, ocfg


def _batches(ocfg, dev, n, B=8, L=10):
    out = []
    for i in range(n):
        d = O.synthetic_batch(ocfg, B, L, seed=100 + i)
        out.append({k: v.to(dev).contiguous() for k, v in d.items()})
    return out


@pytest.mark.parametrize("calc_z,overlap", [(False, True), (True, True), (False, False)])
def test_graph_replay_trains_like_eager(dev, calc_z, overlap):
    from ldrs_amd.hip.graph import GraphedWakeSleep
    m_e, eng_e, opts_e, ocfg = _build(dev)
    m_g, eng_g, opts_g, _ = _build(dev)
    for (n, p), (_, q) in zip(m_e.named_parameters(), m_g.named_parameters()):
        assert torch.equal(p, q), n
    data = _batches(ocfg, dev, len(SCHED))

    eager_losses = []
    for (kc, tau), d in zip(SCHED, data):
        dl, c, eg = eng_e.wake_sleep(opts_e, d, kc, tau, calc_z_loss=calc_z)
        eager_losses.append([float(x) for x in dl] + [float(x) for x in eg] + [float(c.sum())])

    # step 1 = the warm-up step inside the constructor (a real training step on data[0] with SCHED[0])
    g = GraphedWakeSleep(eng_g, opts_g, data[0], calc_z_loss=calc_z, overlap=overlap, warmup_kld=SCHED[0][0],
                         warmup_tau=SCHED[0][1])
    graph_losses = []
    for (kc, tau), d in zip(SCHED[1:], data[1:]):
        dl, c, eg = g(kc, tau, d)
        graph_losses.append([float(x) for x in dl] + [float(x) for x in eg] + [float(c.sum())])
    torch.cuda.synchronize()

    for le, lg in zip(eager_losses[1:], graph_losses):
        for a, b in zip(le, lg):
            assert abs(a - b) <= 1e-6 * max(1.0, abs(a)), (le, lg)
    assert all(st.step == len(SCHED) for st in opts_g.values())
    assert all(int(st.step_dev.item()) == len(SCHED) for st in opts_g.values())
    worst = 0.0
    for (n, p), (_, q) in zip(m_e.named_parameters(), m_g.named_parameters()):
        err = (p - q).abs().max().item() / max(p.abs().max().item(), 1e-12)
        worst = max(worst, err)
        assert err <= 2e-6, (n, err)  # device-side vs host-side double pow() in Adam's bias correction


def test_replay_loop_notices_a_timed_out_persistent_launch_without_synchronising(dev):
    """A caller that replays and never reads a loss back still stops: every 8th replay copies the persistent kernels'
    sticky words to pinned host memory behind itself, every call looks at what came back (hip/graph.py). The word is
    forced here (a healthy GPU never sets it)."""
    from ldrs_amd.hip import ops
    from ldrs_amd.hip.graph import GraphedWakeSleep
    _, eng, opts, ocfg = _build(dev)
    data = _batches(ocfg, dev, 1)
    g = GraphedWakeSleep(eng, opts, data[0])
    for _ in range(3):
        g(0.5, 1.0)
    if not ops._lstm_sync:
        pytest.skip("this shape does not take the persistent kernels")
    try:
        for ws in ops._lstm_sync.values():
            ws[-32] = 1
        with pytest.raises(ops.LstmTimeoutError):
            for _ in range(2 * GraphedWakeSleep.STICKY_EVERY + 2):
                g(0.5, 1.0)
                torch.cuda.synchronize()  # (only so that the test does not depend on how far the host runs ahead)
    finally:
        ops.clear_lstm_sticky_timeouts()
    torch.cuda.synchronize()
    for _ in range(2 * GraphedWakeSleep.STICKY_EVERY + 2):  # re-armed: the loop runs again
        g(0.5, 1.0)
    torch.cuda.synchronize()
    assert ops.lstm_sticky_timeouts() == 0


def test_graph_rejects_other_shapes(dev):
    from ldrs_amd.hip.graph import GraphedWakeSleep
    _, eng, opts, ocfg = _build(dev)
    data = _batches(ocfg, dev, 1)
    g = GraphedWakeSleep(eng, opts, data[0])
    small = {k: v[:4] for k, v in data[0].items()}
    with pytest.raises(ValueError):
        g.load(small)


@pytest.fixture(scope="module")
def one_rank_group(dev):
    """A one-rank RCCL process group: lets the data-parallel schedule (collectives included) run on a single MI355X."""
    import socket
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    torch.cuda.set_device(dev)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    yield
    torch.cuda.synchronize()  # no collective may be in flight (or on the watchdog's list) when the group goes away
    dist.destroy_process_group()


@pytest.mark.parametrize("graphed,calc_z", [(False, False), (True, False), (True, True)])
def test_data_parallel_schedule_one_rank_matches_plain_step(dev, one_rank_group, graphed, calc_z):
    """The data-parallel schedule (each optimizer group's gradients all-reduced on the stream that formed them, right in
    front of the group's Adam kernel: disc before c inference, generator beside the encoder's dX chain, encoder at the
    end) with real RCCL calls on a one-rank group — eager, and as per-branch graphs with the collectives between them —
    must train exactly like the plain step (a one-rank sum is the identity)."""
    from ldrs_amd import parallel
    from ldrs_amd.hip.graph import GraphedWakeSleep
    m_e, eng_e, opts_e, ocfg = _build(dev)
    m_d, _, _, _ = _build(dev)
    parallel.attach(m_d, force=True)
    eng_d = m_d.step_engine()
    assert eng_d.dp and not eng_e.dp
    opts_d = dict(disc=eng_d.new_optimizer("disc"), enc=eng_d.new_optimizer("enc"), gen=eng_d.new_optimizer("gen"))
    data = _batches(ocfg, dev, len(SCHED))
    ref = []
    for (kc, tau), d in zip(SCHED, data):
        dl, c, eg = eng_e.wake_sleep(opts_e, d, kc, tau, calc_z_loss=calc_z)
        ref.append([float(x) for x in dl] + [float(x) for x in eg[:4]] + [float(c.sum())])
    got = []
    if graphed:
        g = GraphedWakeSleep(eng_d, opts_d, data[0], calc_z_loss=calc_z, warmup_kld=SCHED[0][0], warmup_tau=SCHED[0][1])
        # the communication points, each on the stream that formed the gradients: discriminator on side stream 0,
        # generator on side stream 1, the encoder's early-final part (biLSTM, mu|logvar) on side stream 0 behind the LSTM
        # backward, the rest of the encoder group on the main stream at the end
        comms = [(seg[2], [grp for _, grp in seg[1]]) for seg in g.branches if seg[0] == "comm"]
        assert comms == [(0, ["disc", "disc"]), (1, ["gen", "gen"]), (0, ["enc.a"]),
                         (None, ["enc.b", "enc.a", "enc.b"])], comms
        for (kc, tau), d in zip(SCHED[1:], data[1:]):
            dl, c, eg = g(kc, tau, d)
            got.append([float(x) for x in dl] + [float(x) for x in eg[:4]] + [float(c.sum())])
        ref = ref[1:]
    else:
        for (kc, tau), d in zip(SCHED, data):
            dl, c, eg = eng_d.wake_sleep(opts_d, d, kc, tau, calc_z_loss=calc_z)
            got.append([float(x) for x in dl] + [float(x) for x in eg[:4]] + [float(c.sum())])
    torch.cuda.synchronize()
    assert not eng_d._pending
    for a, b in zip(ref, got):
        for x, y in zip(a, b):
            assert abs(x - y) <= 1e-6 * max(1.0, abs(x)), (a, b)
    for (n, p), (_, q) in zip(m_e.named_parameters(), m_d.named_parameters()):
        err = (p - q).abs().max().item() / max(p.abs().max().item(), 1e-12)
        assert err <= 2e-6, (n, err)


@pytest.mark.parametrize("branch_graphs", [True, False])
def test_graph_replay_with_per_timestep_lstm_kernels_survives_a_mid_run_synchronise(dev, branch_graphs, monkeypatch):
    """Shapes outside the persistent LSTM form (H > 512: BASELINE configs[3]) take one launch per timestep. Regression:
    their initial-state slots were zeroed with hipMemset2DAsync, which as a node of a replayed hipGraph was not reliably
    ordered against the step kernels around it (ROCm 7.2): replays issued back to back after a device synchronise
    drifted from the eager step and reached NaN within ~30 steps (bench.py --workload cfg4). The slots are zeroed by a
    kernel now; 24 unsynchronised replays with a synchronise after the 5th must train exactly like 25 eager steps."""
    from ldrs_amd.hip import ops
    from ldrs_amd.hip.graph import GraphedWakeSleep
    monkeypatch.setattr(ops, "LSTM_PERSISTENT", False)
    monkeypatch.setattr(ops, "LSTM_FUSE_LAYERS", False)
    m_e, eng_e, opts_e, ocfg = _build(dev)
    m_g, eng_g, opts_g, _ = _build(dev)
    (d,) = _batches(ocfg, dev, 1)
    for _ in range(25):
        _, _, eg_e = eng_e.wake_sleep(opts_e, d, 0.88, 0.9)
    g = GraphedWakeSleep(eng_g, opts_g, d, warmup_kld=0.88, warmup_tau=0.9, branch_graphs=branch_graphs)
    for it in range(24):
        _, _, eg_g = g(0.88, 0.9, totals=False)
        if it == 4:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    assert float(eg_g[0]) == float(eg_g[0]), "NaN after replays"
    assert abs(float(eg_g[0]) - float(eg_e[0])) <= 2e-6 * abs(float(eg_e[0])), (float(eg_g[0]), float(eg_e[0]))
    for (n, p), (_, q) in zip(m_e.named_parameters(), m_g.named_parameters()):
        err = (p.double() - q.double()).norm().item() / max(p.double().norm().item(), 1e-30)
        assert err < 2e-6, (n, err)
