"""GPU parity of the individual HIP building blocks (hip/engine.py over the C-ABI) against fp64 CPU PyTorch
autograd of the same op. fp32 tolerance: relative Frobenius error <= 1e-5 per output / gradient (the step-level bar
of BASELINE.json is 1e-4; blocks are held to a tighter one so that errors cannot hide in composition)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _mods():
    import ldrs_amd  # noqa: F401
    from ldrs_amd.hip import engine, ops
    return engine, ops


def rel(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double()
    return (a - ref).norm().item() / max(ref.norm().item(), 1e-30)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return torch.randn(*shape, generator=g) * scale


@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("B,T,H,Din,bidir", [(5, 4, 16, 12, True), (17, 6, 36, 23, False), (64, 3, 512, 40, True),
                                             (33, 7, 48, 20, True), (64, 21, 512, 24, False), (32, 15, 256, 24, True),
                                             (64, 20, 512, 24, True),
                                             # row-tiled persistent kernels: (CPW 8, RT 1), (one sequence per launch,
                                             # RT 4), (RT 2, CPW 4), (RT 2, CPW 2, ragged last tile), (RT 4, ragged)
                                             (64, 5, 1024, 24, False), (256, 4, 1024, 16, True),
                                             (256, 5, 512, 16, False), (250, 3, 256, 16, True),
                                             (200, 3, 1024, 8, False)])
def test_lstm_fwd_bwd(dev, B, T, H, Din, bidir, persistent, monkeypatch):
    """persistent=True: the one-launch recurrence (register-resident W_hh, per-batch-tile hand-off of h_t) where the
    shape is eligible; False: one launch per timestep. Both against torch.nn.LSTM in fp64."""
    engine, ops = _mods()
    monkeypatch.setattr(ops, "LSTM_PERSISTENT", persistent)
    nseq = 2 if bidir else 1
    eligible = ops.lstm_persistent_group(nseq, B, T, H, dev) > 0
    assert eligible == (persistent and H % 16 == 0 and H <= 1024)
    torch.manual_seed(B + T + H)
    ref = torch.nn.LSTM(Din, H, 1, batch_first=True, bidirectional=bidir).double()
    x = rnd(B, T, Din, seed=1).double().requires_grad_(True)
    out, (h_n, c_n) = ref(x)
    w_out = rnd(*out.shape, seed=2).double()
    w_c = rnd(*c_n.shape, seed=3).double()
    ((out * w_out).sum() + (c_n * w_c).sum()).backward()

    dirs = [("", False)] + ([("_reverse", True)] if bidir else [])
    P = {n: p.detach().float().to(dev) for n, p in ref.named_parameters()}
    xd = x.detach().float().to(dev).view(B * T, Din)
    seqs = []
    for sfx, rev in dirs:
        s = engine.LstmSeq(B, T, H, dev, rev)
        ops.gemm(xd, P["weight_ih_l0" + sfx].t(), s.xg.view(B * T, 4 * H), bias0=P["bias_ih_l0" + sfx],
                 bias1=P["bias_hh_l0" + sfx])
        s.set_initial()
        seqs.append(s)
    engine.lstm_run_fwd(seqs, [P["weight_hh_l0" + sfx] for sfx, _ in dirs])
    assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
    for di, (s, (sfx, rev)) in enumerate(zip(seqs, dirs)):
        hout = s.hbuf[:, :T] if rev else s.hbuf[:, 1:]
        assert rel(hout, out[:, :, di * H:(di + 1) * H]) < TOL
        assert rel(s.c_final(), c_n[di]) < TOL and rel(s.h_final(), h_n[di]) < TOL
    # backward: upstream grads on every emitted h and on the final cell state
    exts, dcs = [], []
    for di in range(len(dirs)):
        e = w_out[:, :, di * H:(di + 1) * H].float().contiguous().to(dev)
        exts.append((e, T * H, H))
        dcs.append(w_c[di].float().contiguous().to(dev))
    dgs = engine.lstm_run_bwd(seqs, [P["weight_hh_l0" + sfx] for sfx, _ in dirs], exts, [None] * len(dirs), dcs)
    assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
    dx = torch.zeros(B * T, Din, device=dev)
    for di, (s, (sfx, rev)) in enumerate(zip(seqs, dirs)):
        G = {k: torch.full_like(P[k + sfx], float("nan")) for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0",
                                                                    "bias_hh_l0")}
        dg2 = engine.lstm_param_grads(s, dgs[di], (xd, engine.Axis(Din, 0, 0)), G["weight_ih_l0"], G["weight_hh_l0"],
                                      G["bias_ih_l0"], G["bias_hh_l0"])
        ops.gemm(dg2, P["weight_ih_l0" + sfx], dx, beta=1.0)
        for k, g in G.items():
            assert rel(g, dict(ref.named_parameters())[k + sfx].grad) < TOL, k + sfx
    assert rel(dx.view(B, T, Din), x.grad) < TOL


@pytest.mark.parametrize("B,T,H,bidir", [(64, 6, 512, True), (256, 4, 1024, True), (256, 5, 512, False),
                                          (200, 3, 1024, False)])
def test_persistent_lstm_with_the_sticky_word_set_drains_and_recovers(dev, B, T, H, bidir):
    """The fail-stop protocol of the persistent launches (include/cgn_hip.h), at plain and ROW-TILED shapes (RT 1, 4, 2,
    4): with the sticky timeout word set every wait gives up within 64 polls, so the launch drains at once (its results
    are invalid), it still leaves every arrival / exit counter at zero, and after the caller has cleared the word the
    same launch gives the correct result again, bit for bit."""
    engine, ops = _mods()
    nseq = 2 if bidir else 1
    assert ops.lstm_persistent_group(nseq, B, T, H, dev) > 0
    g = torch.Generator().manual_seed(B + H)
    seqs, whh = [], []
    for i in range(nseq):
        s = engine.LstmSeq(B, T, H, dev, bool(i))
        s.xg.copy_(torch.randn(B, T, 4 * H, generator=g))
        s.set_initial()
        seqs.append(s)
        whh.append((torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev))
    engine.lstm_run_fwd(seqs, whh)
    assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
    good = [(s.hbuf.clone(), s.cbuf.clone(), s.gates.clone()) for s in seqs]
    dcs = [torch.randn(B, H, generator=g).to(dev) for _ in seqs]
    exts = [(torch.randn(B, T, H, generator=g).to(dev), T * H, H) for _ in seqs]
    good_dg = [d.clone() for d in engine.lstm_run_bwd(seqs, whh, exts, [None] * nseq, [d.clone() for d in dcs])]
    assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
    try:
        for ws in ops._lstm_sync.values():
            ws[-32] = 1
        for s in seqs:
            s.hbuf.fill_(float("nan"))
        engine.lstm_run_fwd(seqs, whh)
        engine.lstm_run_bwd(seqs, whh, exts, [None] * nseq, [d.clone() for d in dcs])
        torch.cuda.synchronize()  # returns: the grid drained although no hand-off completed
        assert ops.lstm_sticky_timeouts() > 0 and ops.lstm_counters_at_rest(dev)
    finally:
        ops.clear_lstm_sticky_timeouts()
    engine.lstm_run_fwd(seqs, whh)
    for s, (h, c, gt) in zip(seqs, good):
        assert torch.equal(s.hbuf, h) and torch.equal(s.cbuf, c) and torch.equal(s.gates, gt)
    again = engine.lstm_run_bwd(seqs, whh, exts, [None] * nseq, [d.clone() for d in dcs])
    for a, b in zip(again, good_dg):
        assert torch.equal(a, b)
    assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)


@pytest.mark.parametrize("B,T,H,Din", [(5, 4, 16, 12), (64, 21, 512, 24), (32, 16, 256, 20), (33, 7, 48, 9),
                                        (64, 1, 512, 8), (20, 3, 128, 16)])
def test_lstm2_wavefront_fwd_bwd(dev, B, T, H, Din):
    """The two-layer wavefront launches (cgn_lstm2_fwd/bwd_persistent_f32: layer 1 one timestep behind layer 0, its
    input projection folded into the recurrence; backward the mirror image) against torch.nn.LSTM(num_layers=2) in
    fp64: emitted h1, both layers' final states, every parameter gradient and the gradient wrt the input."""
    engine, ops = _mods()
    assert ops.lstm2_persistent_bytes(B, T, H, dev) > 0
    torch.manual_seed(B + T + H)
    ref = torch.nn.LSTM(Din, H, 2, batch_first=True).double()
    x = rnd(B, T, Din, seed=1).double().requires_grad_(True)
    out, (h_n, c_n) = ref(x)
    w_out = rnd(*out.shape, seed=2).double()
    (out * w_out).sum().backward()
    R = dict(ref.named_parameters())
    P = {n: p.detach().float().to(dev).contiguous() for n, p in R.items()}
    xd = x.detach().float().to(dev).view(B * T, Din)
    g0, g1 = engine.LstmSeq(B, T, H, dev, False), engine.LstmSeq(B, T, H, dev, False, need_xg=False)
    ops.gemm(xd, P["weight_ih_l0"].t(), g0.xg.view(B * T, 4 * H), bias0=P["bias_ih_l0"], bias1=P["bias_hh_l0"])
    for s_ in (g0, g1):  # poison: the launch itself must write the zero initial state
        s_.hbuf.fill_(float("nan")), s_.cbuf.fill_(float("nan")), s_.gates.fill_(float("nan"))
    ops.lstm2_fwd(g0.xg, None, P["weight_hh_l0"], P["weight_ih_l1"], P["weight_hh_l1"], P["bias_ih_l1"],
                  P["bias_hh_l1"], g0.hbuf, g0.cbuf, g0.gates, g1.hbuf, g1.cbuf, g1.gates, B, T, H)
    assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
    assert rel(g1.hbuf[:, 1:], out) < TOL
    for l, s_ in enumerate((g0, g1)):
        assert rel(s_.h_final(), h_n[l]) < TOL and rel(s_.c_final(), c_n[l]) < TOL
        assert float(s_.hbuf[:, 0].abs().max()) == 0.0 and float(s_.cbuf[:, 0].abs().max()) == 0.0
    ext = w_out.float().contiguous().to(dev)
    dg0 = torch.full((B, T, 4 * H), float("nan"), device=dev)
    dg1 = torch.full((B, T, 4 * H), float("nan"), device=dev)
    ops.lstm2_bwd(P["weight_hh_l0"], P["weight_ih_l1"], P["weight_hh_l1"], g0.cbuf, g0.gates, g1.cbuf, g1.gates, ext,
                  T * H, H, dg0, dg1, B, T, H)
    assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
    # the blocked hand-off copies (ops.LSTM_HANDOFF) change where the workgroups read from, not one bit of the result
    was = ops.LSTM_HANDOFF
    ops.LSTM_HANDOFF = not was
    try:
        dg0b, dg1b = torch.full_like(dg0, float("nan")), torch.full_like(dg1, float("nan"))
        ops.lstm2_bwd(P["weight_hh_l0"], P["weight_ih_l1"], P["weight_hh_l1"], g0.cbuf, g0.gates, g1.cbuf, g1.gates, ext,
                      T * H, H, dg0b, dg1b, B, T, H)
    finally:
        ops.LSTM_HANDOFF = was
    assert torch.equal(dg0, dg0b) and torch.equal(dg1, dg1b)
    assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
    h0, h0ax = g0.hout_operand()
    for l, (s_, dg, xop) in enumerate(((g0, dg0, (xd, engine.Axis(Din, 0, 0))), (g1, dg1, (h0, h0ax)))):
        G = {k: torch.full_like(P[k + "_l%d" % l], float("nan")) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
        engine.lstm_param_grads(s_, dg, xop, G["weight_ih"], G["weight_hh"], G["bias_ih"], G["bias_hh"])
        for k, g in G.items():
            assert rel(g, R[k + "_l%d" % l].grad) < TOL, (k, l)
    dx = torch.empty(B * T, Din, device=dev)
    ops.gemm(dg0.view(B * T, 4 * H), P["weight_ih_l0"], dx)
    assert rel(dx.view(B, T, Din), x.grad) < TOL


def test_highway_fwd_bwd(dev):
    engine, ops = _mods()
    N, S = 37, 50
    torch.manual_seed(0)
    lin = [[torch.nn.Linear(S, S).double() for _ in range(3)] for _ in range(2)]  # n, l, g per layer
    x = rnd(N, S, seed=5).double().requires_grad_(True)
    y = x
    for n_, l_, g_ in lin:
        gate = torch.sigmoid(g_(y))
        y = gate * F.relu(n_(y)) + (1 - gate) * l_(y)
    w = rnd(N, S, seed=6).double()
    (y * w).sum().backward()
    layers = [dict(nw=n_.weight, nb=n_.bias, lw=l_.weight, lb=l_.bias, gw=g_.weight, gb=g_.bias) for n_, l_, g_ in lin]
    dl = [{k: v.detach().float().to(dev) for k, v in d.items()} for d in layers]
    yd, saved = engine.highway_fwd(x.detach().float().to(dev), dl)
    assert rel(yd, y) < TOL
    grads = [{k: torch.full_like(v, float("nan")) for k, v in d.items()} for d in dl]
    dx = engine.highway_bwd(saved, dl, w.float().to(dev), grads, need_dx=True)
    assert rel(dx, x.grad) < TOL
    for gl, rl in zip(grads, layers):
        for k in gl:
            assert rel(gl[k], rl[k].grad) < TOL, k


def test_tdnn_fwd_bwd(dev):
    engine, ops = _mods()
    N, Wl, Cc, CV = 29, 9, 15, 21
    kspec = [(1, 7), (2, 10), (3, 5), (6, 9)]
    g = torch.Generator().manual_seed(11)
    table = (torch.rand(CV, Cc, generator=g) * 2 - 1).double().requires_grad_(True)
    kernels = [(torch.rand(od, Cc, kW, generator=g) * 2 - 1).double().requires_grad_(True) for kW, od in kspec]
    idx = torch.randint(0, CV, (N, Wl), generator=g)
    x = table[idx].transpose(1, 2)  # [N, Cc, W]
    feat = torch.cat([torch.tanh(F.conv1d(x, k)).max(2)[0] for k in kernels], 1)
    w = rnd(*feat.shape, seed=12).double()
    (feat * w).sum().backward()
    kd = [k.detach().float().to(dev) for k in kernels]
    td = table.detach().float().to(dev)
    fd, saved = engine.tdnn_fwd(idx.to(dev), td, kd)
    assert rel(fd, feat) < TOL
    dk = [torch.full_like(k, float("nan")) for k in kd]
    dt = torch.full_like(td, float("nan"))
    engine.tdnn_bwd(saved, kd, w.float().to(dev), dk, dt)
    for a, b in zip(dk, kernels):
        assert rel(a, b.grad) < TOL
    assert rel(dt, table.grad) < TOL


@pytest.mark.parametrize("B,L,Lx,E,Co,Ks", [(6, 8, 9, 12, 7, (3, 4, 5)), (64, 20, 21, 300, 100, (3, 4, 5)),
                                           (5, 7, 7, 20, 130, (1, 2, 3, 4, 5))])
def test_textcnn_fwd_bwd(dev, B, L, Lx, E, Co, Ks):
    engine, ops = _mods()
    torch.manual_seed(B + L)
    convs = [torch.nn.Conv2d(1, Co, (K, E)).double() for K in Ks]
    fc = torch.nn.Linear(len(Ks) * Co, 2).double()
    x = rnd(B, Lx, E, seed=20).double().requires_grad_(True)
    mask = (torch.rand(B, len(Ks) * Co) > 0.3).double() / 0.7
    y = torch.randint(0, 2, (B,))
    xs = x[:, :L].unsqueeze(1)
    feat = torch.cat([F.max_pool1d(F.relu(c(xs)).squeeze(3), L - K + 1).squeeze(2) for c, K in zip(convs, Ks)], 1)
    logit = fc(feat * mask)
    ls = F.log_softmax(logit, 1)
    ce = F.nll_loss(ls, y)
    ent = -(ls.exp() * ls).sum(1).mean()
    w_ce, w_ent = 0.7, 0.3
    (w_ce * ce + w_ent * ent).backward()

    cw = [c.weight.detach().float().to(dev) for c in convs]
    cb = [c.bias.detach().float().to(dev) for c in convs]
    xd = x.detach().float().to(dev)
    fd, arg = engine.textcnn_features_fwd(xd, B, L, E, Lx * E, cw, cb)
    assert rel(fd, feat) < TOL
    Wd, bd, md = fc.weight.detach().float().to(dev), fc.bias.detach().float().to(dev), mask.float().to(dev)
    logits = torch.empty(B, 2, device=dev)
    dlog = torch.empty(B, 2, device=dev)
    stats = torch.empty(2, device=dev)
    p1 = torch.empty(B, device=dev)
    ops.disc_head(fd, Wd, bd, logits, mask=md, y=y.to(dev), w_ce=w_ce, w_ent=w_ent, p1=p1, dlogits=dlog, stats=stats)
    assert rel(logits, logit) < TOL and rel(stats, torch.stack([ce, ent])) < TOL
    assert rel(p1, ls.exp()[:, 1]) < TOL
    dW, db, dfeat = torch.full_like(Wd, np.nan), torch.full_like(bd, np.nan), torch.empty(B, len(Ks) * Co, device=dev)
    ops.disc_head_bwd(fd, Wd, dlog, mask=md, dW=dW, db=db, dfeat=dfeat)
    assert rel(dW, fc.weight.grad) < TOL and rel(db, fc.bias.grad) < TOL
    Cos = [Co] * len(Ks)
    dcw = [torch.full_like(w, np.nan) for w in cw]
    dcb = [torch.full_like(b, np.nan) for b in cb]
    ops.conv_bwd_dw(list(Ks), Cos, 0, 0, dcw, dcb, dfeat, None, arg, xd, Lx * E, B, L, E)
    for a, c in zip(dcw, convs):
        assert rel(a, c.weight.grad) < TOL
    for a, c in zip(dcb, convs):
        assert rel(a, c.bias.grad) < TOL
    dx = torch.zeros(B, Lx, E, device=dev)
    ops.conv_bwd_dx(list(Ks), Cos, 0, 0, cw, dfeat, None, arg, B, L, E, dx, Lx * E)
    assert rel(dx, x.grad) < TOL
    # accumulate mode doubles the gradient
    ops.conv_bwd_dw(list(Ks), Cos, 0, 0, dcw, dcb, dfeat, None, arg, xd, Lx * E, B, L, E, beta=1.0)
    assert rel(dcw[0], 2 * convs[0].weight.grad) < TOL


def test_textcnn_short_input_raises(dev):
    engine, ops = _mods()
    x = torch.zeros(2, 4, 8, device=dev)
    with pytest.raises(RuntimeError):
        engine.textcnn_features_fwd(x, 2, 4, 8, 32, [torch.zeros(3, 1, 5, 8, device=dev)], [torch.zeros(3, device=dev)])


def test_reparam_kld_and_adam(dev):
    engine, ops = _mods()
    B, Z, Cc = 9, 14, 2
    mulv = rnd(B, 2 * Z, seed=30, scale=0.5).double().requires_grad_(True)
    eps, c, gz = rnd(B, Z, seed=31).double(), torch.randint(0, 2, (B, Cc)).double(), rnd(B, Z + Cc, seed=32).double()
    mu, lv = mulv[:, :Z], mulv[:, Z:]
    z = eps * torch.exp(0.5 * lv) + mu
    kld = (-0.5 * torch.sum(lv - mu ** 2 - torch.exp(lv) + 1, 1)).mean()
    kc = 0.88
    ((z * gz[:, :Z]).sum() + kc * kld).backward()
    md = mulv.detach().float().to(dev)
    zc, sd, kd = torch.empty(B, Z + Cc, device=dev), torch.empty(B, Z, device=dev), torch.empty(1, device=dev)
    ops.reparam_kld_fwd(md, eps.float().to(dev), c.float().to(dev), zc, sd, kd)
    assert rel(zc, torch.cat([z, c], 1)) < TOL and rel(kd, kld.view(1)) < TOL
    assert rel(sd, torch.exp(0.5 * lv)) < TOL
    dm = torch.empty(B, 2 * Z, device=dev)
    ops.reparam_kld_bwd(md, eps.float().to(dev), gz.float().to(dev), kc / B, dm)
    assert rel(dm, mulv.grad) < TOL

    # Adam: three steps against torch.optim.Adam (fp32 on CPU): the 16-byte form (aligned arenas), its n % 4 tail, and
    # the scalar form (an unaligned view)
    for n, off in ((1000, 0), (1003, 0), (1001, 1), (3, 0)):
        p = torch.nn.Parameter(rnd(n, seed=40))
        opt = torch.optim.Adam([p], 5e-5)
        pd = torch.zeros(n + off + 8, device=dev)[off:off + n]
        pd.copy_(p.detach())
        m, v = torch.zeros(n + off + 8, device=dev)[off:off + n], torch.zeros(n + off + 8, device=dev)[off:off + n]
        for step in range(1, 4):
            g = rnd(n, seed=40 + step)
            p.grad = g.clone()
            opt.step()
            gd = torch.zeros(n + off + 8, device=dev)[off:off + n]
            gd.copy_(g)
            ops.adam(pd, gd, m, v, 5e-5, step)
            assert (pd.cpu() - p.detach()).abs().max().item() < 1e-7, (n, off)


def test_small_kernels(dev):
    engine, ops = _mods()
    a = rnd(70, 33, seed=50)
    b = rnd(70, 33, seed=51)
    out = torch.empty(70, device=dev)
    ops.rowdot(a.to(dev), b.to(dev), out)
    assert rel(out, (a.double() * b.double()).sum(1)) < TOL
    cs = torch.empty(33, device=dev)
    ops.colsum(a.to(dev), cs)
    assert rel(cs, a.double().sum(0)) < TOL
    big = rnd(5000, 130, seed=52)
    cs2 = torch.ones(130, device=dev)
    ops.colsum(big.to(dev), cs2, beta=1.0)
    assert rel(cs2, big.double().sum(0) + 1) < TOL
    x3 = rnd(6, 5, 40, seed=53)
    st = torch.empty(6, 40, device=dev)
    ops.sum_time(x3.to(dev), st)
    assert rel(st, x3.double().sum(1)) < TOL
    tr = torch.empty(33, 70, device=dev)
    ops.transpose(a.to(dev), tr)
    assert torch.equal(tr.cpu(), a.t().contiguous())
    table = rnd(50, 12, seed=54)
    idx = torch.tensor([3, 49, 0, 7, 60, -1])  # two out-of-range ids -> zero rows
    o = torch.full((6, 20), 5.0, device=dev)
    ops.gather_rows(table.to(dev), idx.to(dev), o[:, 4:16])
    exp = torch.zeros(6, 12)
    exp[:4] = table[idx[:4]]
    assert torch.equal(o[:, 4:16].cpu(), exp) and (o[:, :4] == 5).all() and (o[:, 16:] == 5).all()
    u = torch.tensor([0.1, 0.9, 0.5], device=dev)
    p1 = torch.tensor([0.2, 0.8, 0.5], device=dev)
    c = torch.empty(3, device=dev)
    ops.bernoulli(p1, u, c)
    assert c.tolist() == [1.0, 0.0, 0.0]
    t = ops.lincomb([(79.0, out[0:1]), (0.5, out[1:2])])
    assert abs(t.item() - (79 * out[0].item() + 0.5 * out[1].item())) < 1e-4 * abs(t.item()) + 1e-6


@pytest.mark.parametrize("layout,act", [(0, 0), (1, 1)])
def test_conv_dx_dense_route_matches_sparse_kernel(dev, layout, act):
    """cgn_pool_bwd_scatter_f32 + grouped product + cgn_conv_col2im_f32 == cgn_conv_bwd_dx_f32 (same sums, other order)."""
    engine, ops = _mods()
    B, L, E = 9, 11, 20
    Ks, Cos = [1, 3, 4, 6], [5, 7, 3, 9]
    F_ = sum(Cos)
    g = torch.Generator().manual_seed(11)
    W = [torch.randn(co, k * E, generator=g).to(dev) if layout == 0 else torch.randn(co, E, k, generator=g).to(dev)
         for k, co in zip(Ks, Cos)]
    dfeat = torch.randn(B, F_, generator=g).to(dev)
    feat = torch.tanh(torch.randn(B, F_, generator=g)).to(dev)
    arg = torch.zeros(B, F_, dtype=torch.int32)
    col = 0
    for k, co in zip(Ks, Cos):
        arg[:, col:col + co] = torch.randint(-1 if act == 0 else 0, L - k + 1, (B, co), generator=g).int()
        col += co
    arg = arg.to(dev)
    stride = L * E + 8
    dx_a = torch.full((B * stride,), float("nan"), device=dev)
    dx_b = torch.full((B * stride,), float("nan"), device=dev)
    ops.conv_bwd_dx(Ks, Cos, act, layout, W, dfeat, feat if act else None, arg, B, L, E, dx_a, stride)
    ops.conv_bwd_dx_dense(Ks, Cos, act, layout, W, dfeat, feat if act else None, arg, B, L, E, dx_b, stride)
    a = dx_a.view(B, stride)[:, :L * E]
    b = dx_b.view(B, stride)[:, :L * E]
    assert rel(b, a.double().cpu()) < 1e-6


@pytest.mark.parametrize("mode", [1, 2])
@pytest.mark.parametrize("B,T,H,bidir,init", [(64, 5, 1024, False, False), (256, 4, 1024, True, True),
                                               (200, 3, 1024, False, False), (128, 6, 768, False, True)])
def test_row_tiled_lstm_bf16_operands_in_the_bf16_mode(dev, B, T, H, bidir, init, mode, monkeypatch):
    """The optional bf16 mode's recurrences at H > 512 (cgn_lstm_set_operand_mode). Mode 1: h.W_hh^T and dgates.W_hh from
    two-term bf16 splits of both operands (three partial products): within 1e-4 of the fp32-MFMA launch of the same
    inputs - a hundred times inside the mode's 1e-2 allowance. Mode 2 (the default of the bf16 mode): plain bf16 operands,
    h_t / dgates_t handed over as bf16 through the blocked scratch, the given initial state cast into it: within the
    mode's 1e-2. Neither gives the same bits as the fp32 launch (the other kernel ran). RT 1, 4, 2 and a ragged chunk
    count (H = 768); zero and given initial states."""
    engine, ops = _mods()
    monkeypatch.setattr(ops, "LSTM_BF16_RT", mode)
    nseq = 2 if bidir else 1
    assert ops.lstm_persistent_group(nseq, B, T, H, dev) > 0
    g = torch.Generator().manual_seed(B + H + T)
    xg = [torch.randn(B, T, 4 * H, generator=g).to(dev) for _ in range(nseq)]
    whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev) for _ in range(nseq)]
    dcs = [torch.randn(B, H, generator=g).to(dev) for _ in range(nseq)]
    exts = [(torch.randn(B, T, H, generator=g).to(dev), T * H, H) for _ in range(nseq)]
    h0 = [torch.randn(B, H, generator=g).to(dev) * 0.5 for _ in range(nseq)]
    c0 = [torch.randn(B, H, generator=g).to(dev) * 0.5 for _ in range(nseq)]

    def run():
        seqs = []
        for i in range(nseq):
            s = engine.LstmSeq(B, T, H, dev, bool(i))
            s.xg.copy_(xg[i])
            if init:
                s.set_initial(h0[i], c0[i])
            else:
                s.set_initial()
            seqs.append(s)
        engine.lstm_run_fwd(seqs, whh)
        dgs = engine.lstm_run_bwd(seqs, whh, exts, [None] * nseq, [d.clone() for d in dcs])
        torch.cuda.synchronize()
        assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
        return [s.hbuf.clone() for s in seqs], [s.cbuf.clone() for s in seqs], [d.clone() for d in dgs]

    ref = run()
    prev = ops.set_compute("bf16")
    try:
        got = run()
        got2 = run()
    finally:
        ops.set_compute(prev)
    again = run()
    for a, b in zip(sum(ref, []), sum(again, [])):
        assert torch.equal(a, b)  # the mode is off again
    for a, b in zip(sum(got, []), sum(got2, [])):
        assert torch.equal(a, b)  # deterministic (fresh scratch, either parity)
    bar = 1e-4 if mode == 1 else 1e-2
    for name, r_, g_ in zip(("h", "c", "dgates"), ref, got):
        for a, b in zip(r_, g_):
            assert not torch.equal(a, b), name
            assert torch.isfinite(b).all(), name
            assert rel(b, a.double().cpu()) < bar, name


@pytest.mark.parametrize("B,T,H,bidir,init", [(64, 7, 512, True, True), (40, 5, 256, False, False), (64, 6, 320, True, False),
                                               (256, 3, 512, False, True), (64, 3, 1024, False, False),
                                               (200, 3, 1024, False, True)])
def test_persistent_lstm_handoff_scratch_changes_no_bit(dev, B, T, H, bidir, init, monkeypatch):
    """The persistent recurrences at H <= 512 read h_t / dgates_t of the other workgroups from the blocked fp32 hand-off
    scratch (whole cache lines per load) instead of the history buffers: same values, same summation order - every
    output equals the launch without the scratch bit for bit; zero and given initial states, both directions, an H
    whose chunks do not fill the waves evenly, and the row-tiled kernels (B = 256 at H = 512; H = 1024 with one and with
    four row tiles, fp32 mode)."""
    engine, ops = _mods()
    nseq = 2 if bidir else 1
    assert ops.lstm_persistent_group(nseq, B, T, H, dev) == nseq
    g = torch.Generator().manual_seed(B * 3 + H + T)
    xg = [torch.randn(B, T, 4 * H, generator=g).to(dev) for _ in range(nseq)]
    whh = [(torch.randn(4 * H, H, generator=g) / H ** 0.5).to(dev) for _ in range(nseq)]
    dcs = [torch.randn(B, H, generator=g).to(dev) for _ in range(nseq)]
    exts = [(torch.randn(B, T, H, generator=g).to(dev), T * H, H) for _ in range(nseq)]
    h0 = [torch.randn(B, H, generator=g).to(dev) * 0.5 for _ in range(nseq)]
    c0 = [torch.randn(B, H, generator=g).to(dev) * 0.5 for _ in range(nseq)]

    def run(handoff):
        monkeypatch.setattr(ops, "LSTM_HANDOFF", handoff)
        seqs = []
        for i in range(nseq):
            s = engine.LstmSeq(B, T, H, dev, bool(i))
            s.xg.copy_(xg[i])
            if init:
                s.set_initial(h0[i], c0[i])
            else:
                s.set_initial()
            seqs.append(s)
        engine.lstm_run_fwd(seqs, whh)
        dgs = engine.lstm_run_bwd(seqs, whh, exts, [None] * nseq, [d.clone() for d in dcs])
        torch.cuda.synchronize()
        assert ops.lstm_sticky_timeouts() == 0 and ops.lstm_counters_at_rest(dev)
        return [s.hbuf.clone() for s in seqs] + [s.cbuf.clone() for s in seqs] + [s.gates.clone() for s in seqs] + \
               [d.clone() for d in dgs]

    ref = run(False)
    for _ in range(2):
        got = run(True)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
