"""GPU parity of the fp32 MFMA GEMM (C-ABI cgn_gemm_f32) against fp64 CPU matmul.

Tolerance: the kernel is an exact-fp32 FMA chain, so |err| <= ~K * eps * sum|a*b|; we check a relative Frobenius
error <= 2e-6 and exactness on small-integer operands (which also catches any fragment-layout transposition,
since the operands are asymmetric).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    import ldrs_amd  # noqa: F401
    from ldrs_amd.hip import ops
    return ops


def _rel(a, ref):
    return (a.double().cpu() - ref).norm().item() / max(ref.norm().item(), 1e-30)


def _mk(shape, gen, ints=False):
    if ints:
        return torch.randint(-3, 4, shape, generator=gen).float()
    return torch.randn(shape, generator=gen)


@pytest.mark.parametrize("M,N,K", [(200, 300, 96), (128, 128, 32), (1, 1, 1), (65, 129, 825), (257, 70, 365),
                                   (64, 64, 1024), (33, 2048, 512)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
@pytest.mark.parametrize("tile", [0, 64, 128])
def test_gemm_layouts(dev, M, N, K, layout, tile):
    ops = _ops()
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K)
    ints = (M + N + K) % 2 == 0
    A = _mk((M, K), g, ints)
    B = _mk((K, N), g, ints)
    ref = A.double() @ B.double()
    if layout[0] == "n":
        a = A.to(dev)
    else:
        a = A.t().contiguous().to(dev).t()
    if layout[1] == "n":
        b = B.to(dev)
    else:
        b = B.t().contiguous().to(dev).t()
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(a, b, out, force_tile=tile)
    if ints:
        assert torch.equal(out.cpu().double(), ref)
    else:
        assert _rel(out, ref) < 2e-6


@pytest.mark.parametrize("splitk", [2, 5])
def test_gemm_splitk_bias_beta(dev, splitk):
    ops = _ops()
    g = torch.Generator().manual_seed(7)
    M, N, K = 150, 90, 1000
    A, B = _mk((M, K), g), _mk((K, N), g)
    C0, b0, b1 = _mk((M, N), g), _mk((N,), g), _mk((N,), g)
    ref = 0.5 * (A.double() @ B.double()) + 2.0 * C0.double() + b0.double() + b1.double()
    out = C0.to(dev).clone()
    ops.gemm(A.to(dev), B.to(dev), out, alpha=0.5, beta=2.0, bias0=b0.to(dev), bias1=b1.to(dev),
             force_splitk=splitk)
    assert _rel(out, ref) < 2e-6
    # same through the heuristic path, into a strided output view
    big = torch.zeros(M, N + 13, device=dev)
    view = big[:, 5:5 + N]
    view.copy_(C0.to(dev))
    ops.gemm(A.to(dev), B.to(dev), view, alpha=0.5, beta=2.0, bias0=b0.to(dev), bias1=b1.to(dev))
    assert _rel(view, ref) < 2e-6
    assert big[:, :5].abs().max().item() == 0 and big[:, 5 + N:].abs().max().item() == 0


def test_gemm_conv_windows(dev):
    """Text-CNN as implicit GEMM: row (b,t) of A is the contiguous window x[b, t:t+Kw, :] (discriminator_sentiment.py:64)."""
    ops = _ops()
    from ldrs_amd.hip._lib import Axis
    g = torch.Generator().manual_seed(3)
    Bn, Lx, L, E, Kw, Co = 5, 9, 8, 12, 3, 7
    x = _mk((Bn, Lx, E), g)
    W = _mk((Co, 1, Kw, E), g)
    bias = _mk((Co,), g)
    ref = torch.nn.functional.conv2d(x[:, :L].double().unsqueeze(1), W.double(), bias.double()).squeeze(3)  # [B,Co,L-Kw+1]
    ref = ref.permute(0, 2, 1).reshape(Bn * (L - Kw + 1), Co)
    xd, Wd = x.to(dev), W.to(dev)
    out = torch.empty(Bn * (L - Kw + 1), Co, device=dev)
    ops.gemm(xd, Wd.view(Co, Kw * E).t(), out, M=Bn * (L - Kw + 1), K=Kw * E,
             a_m=Axis(E, L - Kw + 1, Lx * E), a_k=Axis(1, 0, 0), bias0=bias.to(dev))
    assert _rel(out, ref) < 2e-6


def test_gemm_history_buffer_tn(dev):
    """dW_hh = dGates^T · h_prev with h_prev = hbuf[:, :T] of a [B, T+1, H] buffer (K axis two-level on B)."""
    ops = _ops()
    from ldrs_amd.hip._lib import Axis
    g = torch.Generator().manual_seed(4)
    Bn, T, H = 6, 5, 16
    dg = _mk((Bn, T, 4 * H), g)
    hbuf = _mk((Bn, T + 1, H), g)
    ref = dg.double().reshape(Bn * T, 4 * H).t() @ hbuf[:, :T].double().reshape(Bn * T, H)
    out = torch.empty(4 * H, H, device=dev)
    ops.gemm(dg.to(dev).view(Bn * T, 4 * H).t(), hbuf.to(dev), out, N=H, K=Bn * T,
             b_k=Axis(H, T, (T + 1) * H), b_n=Axis(1, 0, 0))
    assert _rel(out, ref) < 2e-6


def test_gemm_softmax_operand_and_backward_epilogue(dev):
    """soft-embedding fwd (A = exp(logits/tau - lse) on the fly) and the fused dlogits epilogue (cgn.py:1086-1108)."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, V, E = 70, 333, 24
    tau = 0.9
    logits = (_mk((M, V), g) * 3).double().requires_grad_(True)
    emb = _mk((V, E), g).double()
    target = torch.randint(0, V, (M,), generator=g)
    dout = _mk((M, E), g).double()
    ls = torch.log_softmax(logits / tau, dim=1)
    ce = torch.nn.functional.nll_loss(ls, target) * 79
    soft = torch.exp(ls) @ emb
    (ce + (soft * dout).sum()).backward()
    ref_soft, ref_dlogits = soft.detach(), logits.grad

    lg = logits.detach().float().to(dev)
    embd = emb.float().to(dev)
    tg = target.to(dev)
    lse, nll, loss = ops.softmax_nll_fwd(lg, tg, 1.0 / tau)
    assert abs(loss.item() * 79 - ce.item()) / abs(ce.item()) < 1e-6
    out = torch.empty(M, E, device=dev)
    ops.gemm(lg, embd, out, a_exp=(1.0 / tau, lse))
    assert _rel(out, ref_soft) < 2e-6
    dod = dout.float().to(dev)
    rowdot = (dod * out).sum(1).contiguous()
    dlogits = torch.empty(M, V, device=dev)
    ops.gemm(dod, embd.t(), dlogits, softmax_bwd=dict(logits=lg, lse=lse, rowdot=rowdot, target=tg,
                                                     inv_tau=1.0 / tau, ce_scale=79.0 / (M * tau)))
    assert _rel(dlogits, ref_dlogits) < 5e-6
    # in place over the logits (what the training step does to save 107 MB at cfg2)
    lg2 = lg.clone()
    ops.gemm(dod, embd.t(), lg2, softmax_bwd=dict(logits=lg2, lse=lse, rowdot=rowdot, target=tg,
                                                 inv_tau=1.0 / tau, ce_scale=79.0 / (M * tau)))
    assert torch.equal(lg2, dlogits)


def test_gemm_rejects_bad_args(dev):
    ops = _ops()
    a = torch.zeros(4, 8, device=dev)
    b = torch.zeros(9, 4, device=dev)
    with pytest.raises(ValueError):
        ops.gemm(a, b, torch.zeros(4, 4, device=dev))
    with pytest.raises(TypeError):
        ops.gemm(a.cpu(), b, torch.zeros(4, 4, device=dev))
    from ldrs_amd.hip._lib import Axis
    with pytest.raises(ValueError):  # window addressing that would run past the buffer
        ops.gemm(a, torch.zeros(16, 4, device=dev), torch.zeros(8, 4, device=dev), M=8, K=16,
                 a_m=Axis(8, 0, 0), a_k=Axis(1, 0, 0))


@pytest.mark.parametrize("layouts", [("nt", "nt", "nt"), ("nn", "nn"), ("nt", "nn", "tn"), ("tn",) * 8])
def test_gemm_group_matches_single_launches(dev, layouts):
    """cgn_gemm_group_f32: several products in one launch (members of different shapes, also of different operand
    orientations: the group then drops to scalar loads) == the same products one by one; exact on integer operands."""
    ops = _ops()
    g = torch.Generator().manual_seed(len(layouts))
    calls, refs, outs = [], [], []
    for i, layout in enumerate(layouts):
        M, N, K = 70 + 33 * i, 40 + 28 * i, 24 + 36 * i
        A = _mk((M, K), g, True)
        B = _mk((K, N), g, True)
        bias = _mk((N,), g, True).to(dev)
        a = A.to(dev) if layout[0] == "n" else A.t().contiguous().to(dev).t()
        b = B.to(dev) if layout[1] == "n" else B.t().contiguous().to(dev).t()
        out = torch.full((M, N), float("nan"), device=dev)
        calls.append((a, b, out, dict(bias0=bias)))
        refs.append(A.double() @ B.double() + bias.cpu().double())
        outs.append(out)
    ops.gemm_group(calls)
    for out, ref in zip(outs, refs):
        assert torch.equal(out.cpu().double(), ref)


# ---- optional bf16 operand mode (cgn_gemm_desc.compute = CGN_COMPUTE_BF16) ----------------------------------------
def _bf(x):
    return x.bfloat16().double()


@pytest.mark.parametrize("M,N,K", [(200, 300, 96), (1, 1, 1), (65, 129, 825), (257, 70, 365), (64, 64, 1024),
                                   (33, 2048, 512)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
@pytest.mark.parametrize("tile", [0, 64, 128])
def test_gemm_bf16_layouts(dev, M, N, K, layout, tile):
    """bf16 operands, fp32 accumulation: equal to the fp64 product of the bf16-ROUNDED operands up to fp32 summation
    error (<= 2e-6 relative), exact on small integers, and within north-star's 1e-2 of the unrounded product."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K + 1)
    ints = (M + N + K) % 2 == 0
    A = _mk((M, K), g, ints)
    B = _mk((K, N), g, ints)
    ref = _bf(A) @ _bf(B)
    a = A.to(dev) if layout[0] == "n" else A.t().contiguous().to(dev).t()
    b = B.to(dev) if layout[1] == "n" else B.t().contiguous().to(dev).t()
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(a, b, out, force_tile=tile, compute="bf16")
    if ints:
        assert torch.equal(out.cpu().double(), ref)
    else:
        assert _rel(out, ref) < 2e-6
        assert _rel(out, A.double() @ B.double()) < 1e-2


def test_gemm_bf16_splitk_windows_group_and_softmax(dev):
    ops = _ops()
    from ldrs_amd.hip._lib import Axis
    g = torch.Generator().manual_seed(11)
    # split-K + bias + beta
    M, N, K = 150, 90, 1000
    A, B = _mk((M, K), g), _mk((K, N), g)
    C0, b0 = _mk((M, N), g), _mk((N,), g)
    ref = 0.5 * (_bf(A) @ _bf(B)) + 2.0 * C0.double() + b0.double()
    out = C0.to(dev).clone()
    ops.gemm(A.to(dev), B.to(dev), out, alpha=0.5, beta=2.0, bias0=b0.to(dev), force_splitk=5, compute="bf16")
    assert _rel(out, ref) < 2e-6
    # overlapping conv windows (two-level A rows)
    Bn, Lx, L, E, Kw, Co = 5, 9, 8, 12, 3, 7
    x = _mk((Bn, Lx, E), g)
    W = _mk((Co, 1, Kw, E), g)
    refc = torch.nn.functional.conv2d(_bf(x[:, :L]).unsqueeze(1), _bf(W)).squeeze(3).permute(0, 2, 1)
    refc = refc.reshape(Bn * (L - Kw + 1), Co)
    outc = torch.empty(Bn * (L - Kw + 1), Co, device=dev)
    ops.gemm(x.to(dev), W.to(dev).view(Co, Kw * E).t(), outc, M=Bn * (L - Kw + 1), K=Kw * E,
             a_m=Axis(E, L - Kw + 1, Lx * E), a_k=Axis(1, 0, 0), compute="bf16")
    assert _rel(outc, refc) < 2e-6
    # grouped launch
    calls, refs, outs = [], [], []
    for i in range(3):
        Mi, Ni, Ki = 70 + 33 * i, 40 + 28 * i, 24 + 36 * i
        Ai, Bi = _mk((Mi, Ki), g), _mk((Ki, Ni), g)
        o = torch.full((Mi, Ni), float("nan"), device=dev)
        calls.append((Ai.to(dev), Bi.to(dev), o, dict(compute="bf16")))
        refs.append(_bf(Ai) @ _bf(Bi))
        outs.append(o)
    ops.gemm_group(calls)
    for o, r in zip(outs, refs):
        assert _rel(o, r) < 2e-6
    # softmax operand + fused softmax-backward epilogue against the UNROUNDED fp64 math: the 1e-2 bar of the bf16 mode
    M, V, E = 70, 333, 24
    tau = 0.9
    logits = (_mk((M, V), g) * 3).double().requires_grad_(True)
    emb = _mk((V, E), g).double()
    target = torch.randint(0, V, (M,), generator=g)
    dout = _mk((M, E), g).double()
    ls = torch.log_softmax(logits / tau, dim=1)
    soft = torch.exp(ls) @ emb
    (torch.nn.functional.nll_loss(ls, target) * 79 + (soft * dout).sum()).backward()
    lg, embd, tg = logits.detach().float().to(dev), emb.float().to(dev), target.to(dev)
    lse, _, _ = ops.softmax_nll_fwd(lg, tg, 1.0 / tau)
    out = torch.empty(M, E, device=dev)
    ops.gemm(lg, embd, out, a_exp=(1.0 / tau, lse), compute="bf16")
    assert _rel(out, soft.detach()) < 5e-3
    dod = dout.float().to(dev)
    rowdot = (dod * out).sum(1).contiguous()
    dlogits = torch.empty(M, V, device=dev)
    ops.gemm(dod, embd.t(), dlogits, compute="bf16",
             softmax_bwd=dict(logits=lg, lse=lse, rowdot=rowdot, target=tg, inv_tau=1.0 / tau, ce_scale=79.0 / (M * tau)))
    assert _rel(dlogits, logits.grad) < 1e-2


# ---- bf16 STORAGE of the logits-sized tensors (cgn_gemm_desc.storage, CGN_STORE_*) ---------------------------------
@pytest.mark.parametrize("M,N,K", [(200, 300, 96), (65, 129, 824), (1, 8, 8), (1344, 512, 2000), (300, 64, 4096)])
@pytest.mark.parametrize("b_layout", ["n", "t"])
@pytest.mark.parametrize("tile", [0, 64, 128])
def test_gemm_bf16_stored_a(dev, M, N, K, b_layout, tile):
    """A held as bfloat16 in memory (read with 16-byte loads straight into the bf16 LDS image), B fp32 rounded inside the
    kernel: equal to the fp64 product of the bf16 values up to fp32 summation order; exact on small integers."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K)
    ints = (M + N + K) % 2 == 0
    A = _mk((M, K), g, ints)
    B = _mk((K, N), g, ints)
    a = A.to(dev).to(torch.bfloat16)
    ref = a.cpu().double() @ _bf(B).double()
    b = B.to(dev) if b_layout == "n" else B.t().contiguous().to(dev).t()
    if N % 4:  # (16-byte loads along n need aligned rows: the library rejects a B it cannot read that way)
        b = b if b_layout == "t" and K % 4 == 0 else None
    if b is None:
        pytest.skip("B not readable with 16-byte loads")
    out = torch.full((M, N), float("nan"), device=dev)
    ops.gemm(a, b, out, force_tile=tile, compute="bf16")
    if ints:
        assert torch.equal(out.cpu().double(), ref)
    else:
        assert _rel(out, ref) < 2e-6


def test_gemm_bf16_stored_a_rows_with_padding_and_split_k(dev):
    """Row stride larger than K (a view into a wider bf16 matrix) and a forced K split."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    M, N, K, ld = 150, 96, 1600, 1608
    big = torch.randn(M, ld, generator=g).to(dev).to(torch.bfloat16)
    a = big[:, :K]
    B = torch.randn(K, N, generator=g)
    ref = a.cpu().double() @ _bf(B).double()
    for sk in (0, 5):
        out = torch.full((M, N), float("nan"), device=dev)
        ops.gemm(a, B.to(dev), out, compute="bf16", force_splitk=sk)
        assert _rel(out, ref) < 2e-6, sk


def test_gemm_bf16_storage_rejects_what_it_cannot_read(dev):
    ops = _ops()
    a = torch.zeros(16, 24, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(24, 16, device=dev)
    out = torch.zeros(16, 16, device=dev)
    with pytest.raises(ValueError):
        ops.gemm(a, b, out, compute="f32")  # bf16 tensors need the bf16 mode
    with pytest.raises(ValueError):
        ops.gemm(a[:, :20], b[:20], out, compute="bf16")  # K % 8
    with pytest.raises(ValueError):
        ops.gemm(a.t().contiguous().t(), b, out, compute="bf16")  # not k-contiguous
    with pytest.raises(ValueError):
        ops.gemm(a, b, out.to(torch.bfloat16), beta=1.0, compute="bf16")


@pytest.mark.parametrize("bb", [False, True])
@pytest.mark.parametrize("M,V,E", [(96, 1200, 300), (130, 2056, 64), (264, 1032, 304), (520, 777 + 7, 300)])
def test_gemm_bf16_stored_logits_chain(dev, M, V, E, bb, monkeypatch):
    """The generator's output layer with the logits-sized tensors held in bfloat16, product by product:
    vocabulary projection -> bf16 logits; softmax/NLL over bf16 logits; soft embedding with the exp loader over bf16
    logits; fused softmax backward reading bf16 logits in place, writing bf16 dlogits AND their transpose; dH and dW from
    the bf16 dlogits; bias gradient. Every step against fp64 math on the SAME bf16 inputs (so only the kernel's own
    rounding is measured): fp32 outputs 2e-5 (exp operand: 1e-2 bar of the mode, measured ~2e-3), bf16 outputs within
    one bf16 ulp (2^-8 relative) of the fp64 value."""
    ops = _ops()
    # bb: every product the bf16 x bf16 kernel takes (M, N >= 128, K >= 64) runs there - projection with a bf16 result,
    # exp loader, fused softmax backward with the transposed copy, dH, dW; otherwise the kernels that round at staging
    monkeypatch.setattr(ops, "BB_MIN_FLOP", 0.0 if bb else 1e30)
    g = torch.Generator().manual_seed(M + V)
    H, tau = 64, 0.7
    h = torch.randn(M, H, generator=g)
    W = torch.randn(V, H, generator=g) * 0.2
    bias = torch.randn(V, generator=g) * 0.1
    tgt = torch.randint(0, V, (M,), generator=g)
    emb = torch.randn(V, E, generator=g) * 0.5
    dout = torch.randn(M, E, generator=g)
    hd, Wd, embd, tgd = h.to(dev), W.to(dev), emb.to(dev), tgt.to(dev)
    # 1. logits = h W^T + b, stored bf16
    logits = torch.full((M, V), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.gemm(hd, Wd.t(), logits, bias0=bias.to(dev), compute="bf16")
    ref_logits = _bf(h).double() @ _bf(W).double().t() + bias.double()
    got = logits.float().cpu().double()
    assert ((got - ref_logits).abs() <= ref_logits.abs() * 2.0 ** -8 + 1e-6).all()
    L = got  # everything downstream is checked against THESE stored values
    # 2. softmax / NLL
    lse, nll, loss = ops.softmax_nll_fwd(logits, tgd, 1.0 / tau)
    ref_lse = torch.logsumexp(L / tau, 1)
    assert (lse.cpu().double() - ref_lse).abs().max().item() < 1e-4
    assert (nll.cpu().double() - (ref_lse - L[torch.arange(M), tgt] / tau)).abs().max().item() < 1e-4
    # 3. soft embedding: softmax(L/tau) @ E with the probabilities formed in the loader from the bf16 logits
    out_emb = torch.full((M, E), float("nan"), device=dev)
    ops.gemm(logits, embd, out_emb, a_exp=(1.0 / tau, lse), compute="bf16")
    P = torch.exp(L / tau - ref_lse[:, None])
    assert _rel(out_emb, P @ emb.double()) < 1e-2
    # 4. fused softmax/CE backward in place over the bf16 logits, plus the transposed copy
    rowdot = torch.empty(M, device=dev)
    ops.rowdot(dout.to(dev), out_emb, rowdot)
    ce_scale = 79.0 / tau / M
    dlt = torch.full((V, M), float("nan"), device=dev, dtype=torch.bfloat16)
    ops.gemm(dout.to(dev), embd.t(), logits, compute="bf16", c_t=dlt,
             softmax_bwd=dict(logits=logits, lse=lse, rowdot=rowdot, target=tgd, inv_tau=1.0 / tau, ce_scale=ce_scale))
    acc = dout.double() @ emb.double().t()
    onehot = torch.zeros(M, V, dtype=torch.float64)
    onehot[torch.arange(M), tgt] = 1.0
    ref_dl = P * (acc - rowdot.cpu().double()[:, None]) / tau + ce_scale * (P - onehot)
    dl = logits.float().cpu().double()
    assert _rel(dl, ref_dl) < 1e-2  # (bf16 operands of the product itself: the mode's bar)
    assert torch.equal(dlt.float().cpu(), logits.float().cpu().t())
    # 5. dH = dlogits @ W (A bf16, K = V) and dW = dlogits^T @ h (A = the transposed copy, K = M)
    if M % 8 == 0:
        dH = torch.full((M, H), float("nan"), device=dev)
        ops.gemm(logits, Wd, dH, compute="bf16")
        assert _rel(dH, dl @ _bf(W).double()) < 2e-5
        dW = torch.full((V, H), float("nan"), device=dev)
        ops.gemm(dlt, hd, dW, compute="bf16")
        assert _rel(dW, dl.t() @ _bf(h).double()) < 2e-5
    # 6. bias gradient
    db = torch.empty(V, device=dev)
    ops.colsum(logits, db)
    assert _rel(db, dl.sum(0)) < 2e-6


@pytest.mark.parametrize("tile", [64, 128])
@pytest.mark.parametrize("M,N,K", [(200, 300, 96), (129, 131, 40), (256, 256, 64)])
def test_gemm_unsplit_epilogue_nests(dev, tile, M, N, K):
    """The in-kernel epilogue nests of an unsplit product (csrc/gemm_core.h: linear, linear + beta): alpha, both biases
    and beta * C against fp64, ragged edges included."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K + tile)
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    b0, b1, C0 = torch.randn(N, generator=g), torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    for beta in (0.0, 0.75):
        out = C0.clone().to(dev)
        ops.gemm(A.to(dev), B.to(dev), out, alpha=0.5, beta=beta, bias0=b0.to(dev), bias1=b1.to(dev), force_tile=tile,
                 force_splitk=1)
        ref = 0.5 * (A.double() @ B.double()) + b0.double() + b1.double() + beta * C0.double()
        assert _rel(out, ref) < 2e-6, beta


# ---- X6 products: fp32 accuracy from the bf16 matrix pipe (include/cgn_hip.h CGN_COMPUTE_F32X6) -------------------------
def _x6_ran(ops, fn):
    """Runs fn() (one ops.gemm call) with the launch profile on and tells whether the library took an X6 kernel."""
    ops.GEMM_PROFILE = {"variant": None, "records": []}
    try:
        fn()
        torch.cuda.synchronize()
        return bool(ops.GEMM_PROFILE["records"][-1][0][3] & 32)
    finally:
        ops.GEMM_PROFILE = None


@pytest.mark.parametrize("M,N,K", [(200, 300, 96), (128, 128, 32), (4, 4, 4), (68, 132, 824), (260, 72, 364),
                                   (64, 64, 1024), (36, 2048, 512), (1344, 512, 2000), (256, 300, 20), (132, 260, 16)])
@pytest.mark.parametrize("layout", ["nt", "nn", "tn"])
@pytest.mark.parametrize("tile", [0, 64, 128])
def test_gemm_x6_layouts(dev, M, N, K, layout, tile):
    """Every layout the X6 translation unit instantiates, ragged tile edges and K tails (K % 16 != 0) included: the X6
    kernel must be the one that runs, and its result meets the fp32-MFMA kernel's bar (2e-6 relative Frobenius against
    fp64; small-integer operands - exact bf16 values, so two of the three terms are zero - bit-exact)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 131 + N * 17 + K)
    ints = (M + N + K) % 8 == 0
    A, B = _mk((M, K), g, ints), _mk((K, N), g, ints)
    ref = A.double() @ B.double()
    a = A.to(dev) if layout[0] == "n" else A.t().contiguous().to(dev).t()
    b = B.to(dev) if layout[1] == "n" else B.t().contiguous().to(dev).t()
    out = torch.full((M, N), float("nan"), device=dev)
    assert _x6_ran(ops, lambda: ops.gemm(a, b, out, force_tile=tile, compute="f32x6"))
    if ints:
        assert torch.equal(out.cpu().double(), ref)
    else:
        assert _rel(out, ref) < 2e-6
    # and no worse than the fp32 MFMA product of the same operands, element by element, against the size of the terms
    out32 = torch.empty(M, N, device=dev)
    assert not _x6_ran(ops, lambda: ops.gemm(a, b, out32, force_tile=tile, compute="f32_mfma"))
    scale = (A.abs().double() @ B.abs().double()).clamp_min(1e-30)
    w6 = ((out.cpu().double() - ref).abs() / scale).max().item()
    w32 = ((out32.cpu().double() - ref).abs() / scale).max().item()
    assert w6 <= max(2.0 * w32, 2.0 ** -22), (w6, w32)


def test_gemm_x6_operand_split_is_exact_over_the_exponent_range(dev):
    """x = p0 + p1 + p2 must hold bit for bit: products with a one-hot operand return the other operand's elements
    exactly (1.0 splits into (1, 0, 0), so only a.b0 + a.b1 + a.b2 terms are non-zero), for magnitudes 2^-60 .. 2^60."""
    ops = _ops()
    g = torch.Generator().manual_seed(99)
    M = N = K = 128
    A = torch.randn(M, K, generator=g) * torch.exp2(torch.randint(-60, 61, (M, K), generator=g).float())
    eye = torch.eye(K)
    out = torch.empty(M, N, device=dev)
    assert _x6_ran(ops, lambda: ops.gemm(A.to(dev), eye.to(dev), out, compute="f32x6"))
    assert torch.equal(out.cpu(), A)
    assert _x6_ran(ops, lambda: ops.gemm(eye.to(dev), A.to(dev), out, compute="f32x6"))
    assert torch.equal(out.cpu(), A)


@pytest.mark.parametrize("splitk", [0, 2, 5])
def test_gemm_x6_splitk_bias_beta_history_and_softmax(dev, splitk):
    ops = _ops()
    from ldrs_amd.hip._lib import Axis
    g = torch.Generator().manual_seed(7 + splitk)
    M, N, K = 152, 92, 1000
    A, B = _mk((M, K), g), _mk((K, N), g)
    C0, b0, b1 = _mk((M, N), g), _mk((N,), g), _mk((N,), g)
    ref = 0.5 * (A.double() @ B.double()) + 2.0 * C0.double() + b0.double() + b1.double()
    out = C0.to(dev).clone()
    assert _x6_ran(ops, lambda: ops.gemm(A.to(dev), B.to(dev), out, alpha=0.5, beta=2.0, bias0=b0.to(dev), bias1=b1.to(dev),
                                         force_splitk=splitk, compute="f32x6"))
    assert _rel(out, ref) < 2e-6
    # dW_hh = dGates^T . h_prev over a [B, T+1, H] history buffer (B's K axis two-level: the kernel's FK = 2 form)
    Bn, T, H = 12, 20, 32
    dg, hbuf = _mk((Bn, T, 4 * H), g), _mk((Bn, T + 1, H), g)
    refw = dg.double().reshape(Bn * T, 4 * H).t() @ hbuf[:, :T].double().reshape(Bn * T, H)
    outw = torch.empty(4 * H, H, device=dev)
    assert _x6_ran(ops, lambda: ops.gemm(dg.to(dev).view(Bn * T, 4 * H).t(), hbuf.to(dev), outw, N=H, K=Bn * T,
                                         b_k=Axis(H, T, (T + 1) * H), b_n=Axis(1, 0, 0), force_splitk=splitk,
                                         compute="f32x6"))
    assert _rel(outw, refw) < 2e-6
    # soft embedding (A = exp(logits / tau - lse) on the fly) and the fused dlogits epilogue (cgn.py:1086-1108)
    Ms, V, E = 72, 336, 24
    tau = 0.9
    logits = (_mk((Ms, V), g) * 3).double().requires_grad_(True)
    emb = _mk((V, E), g).double()
    target = torch.randint(0, V, (Ms,), generator=g)
    dout = _mk((Ms, E), g).double()
    ls = torch.log_softmax(logits / tau, dim=1)
    soft = torch.exp(ls) @ emb
    (torch.nn.functional.nll_loss(ls, target) * 79 + (soft * dout).sum()).backward()
    lg, embd, tg = logits.detach().float().to(dev), emb.float().to(dev), target.to(dev)
    lse, nll, loss = ops.softmax_nll_fwd(lg, tg, 1.0 / tau)
    so = torch.empty(Ms, E, device=dev)
    assert _x6_ran(ops, lambda: ops.gemm(lg, embd, so, a_exp=(1.0 / tau, lse), force_splitk=splitk, compute="f32x6"))
    assert _rel(so, soft.detach()) < 2e-6
    dod = dout.float().to(dev)
    rowdot = (dod * so).sum(1).contiguous()
    dl = torch.empty(Ms, V, device=dev)
    sb = dict(logits=lg, lse=lse, rowdot=rowdot, target=tg, inv_tau=1.0 / tau, ce_scale=79.0 / (Ms * tau))
    for tile in (64, 128):
        assert _x6_ran(ops, lambda: ops.gemm(dod, embd.t(), dl, softmax_bwd=sb, force_tile=tile, compute="f32x6"))
        assert _rel(dl, logits.grad) < 5e-6


def test_gemm_x6_falls_back_where_it_has_no_variant(dev):
    """Scalar-load operands (unaligned rows) run the fp32-MFMA kernel under CGN_COMPUTE_F32X6: same result path as
    compute='f32_mfma', bit for bit."""
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    A, B = _mk((65, 129), g).to(dev), _mk((129, 67), g).to(dev)
    o1, o2 = torch.empty(65, 67, device=dev), torch.empty(65, 67, device=dev)
    assert not _x6_ran(ops, lambda: ops.gemm(A, B, o1, compute="f32x6"))
    ops.gemm(A, B, o2, compute="f32_mfma")
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("M,V,K", [(1344, 20000, 512), (256, 300, 64), (320, 1000, 96), (260, 257, 32)])
def test_gemm_fused_log_softmax_statistics(dev, M, V, K):
    """The vocabulary projection with the CE's first pass in its epilogue (CGN_EPI_LINEAR_STATS, generator.py:61 ->
    cgn.py:1086-1092): logits bit-identical to the plain product's, lse / nll / mean NLL against fp64 log_softmax of
    those logits (2e-6) and against the stand-alone softmax kernel; ragged last chunks and row tiles included."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + V + K)
    tau = 0.7
    h, W, bias = _mk((M, K), g), _mk((V, K), g) * 0.3, _mk((V,), g)
    tgt = torch.randint(0, V, (M,), generator=g)
    hd, Wd, bd, td = h.to(dev), W.to(dev), bias.to(dev), tgt.to(dev)
    plain = torch.empty(M, V, device=dev)
    ops.gemm(hd, Wd.t(), plain, bias0=bd, compute="f32x6", force_tile=128)
    logits = torch.full((M, V), float("nan"), device=dev)
    out, stats = ops.gemm(hd, Wd.t(), logits, bias0=bd, compute="f32x6", lse_stats=1.0 / tau)
    assert stats is not None and stats.shape == (M, (V + 63) // 64, 2)
    assert torch.equal(logits, plain)
    lse, nll, loss = ops.softmax_nll_from_stats(stats, logits, td, 1.0 / tau)
    ls = torch.log_softmax(logits.double().cpu() / tau, dim=1)
    ref_lse = torch.logsumexp(logits.double().cpu() / tau, dim=1)
    ref_nll = -ls[torch.arange(M), tgt]
    assert _rel(lse, ref_lse) < 2e-6 and _rel(nll, ref_nll) < 2e-6
    assert abs(loss.item() - ref_nll.mean().item()) < 2e-6 * abs(ref_nll.mean().item())
    lse2, nll2, loss2 = ops.softmax_nll_fwd(logits, td, 1.0 / tau)
    assert _rel(lse, lse2.double().cpu()) < 1e-6 and abs(loss.item() - loss2.item()) < 2e-6 * abs(loss2.item())
    # twice in a row (the ticket re-arms), with the temperature read from device memory the second time
    dyn = torch.tensor([1.0 / tau, 0.0], device=dev)
    logits.fill_(float("nan"))
    _, stats_b = ops.gemm(hd, Wd.t(), logits, bias0=bd, compute="f32x6", lse_stats=123.0, dyn=dyn)
    lse3, nll3, loss3 = ops.softmax_nll_from_stats(stats_b, logits, td, 123.0, dyn=dyn)
    assert torch.equal(lse3, lse) and torch.equal(nll3, nll) and torch.equal(loss3, loss)


def test_gemm_fused_statistics_are_refused_where_no_kernel_serves_them(dev):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    a, b = _mk((300, 64), g).to(dev), _mk((64, 300), g).to(dev)  # B row-contiguous: not the NT product
    out = torch.empty(300, 300, device=dev)
    _, stats = ops.gemm(a, b, out, compute="f32x6", lse_stats=1.0)
    assert stats is None
    _, stats = ops.gemm(a, b.t().contiguous().t(), out, compute="f32_mfma", lse_stats=1.0)
    assert stats is None
    assert _rel(out, a.double().cpu() @ b.double().cpu()) < 2e-6
    from ldrs_amd.hip import _lib
    import ctypes as C
    d = ops.gemm(a, b, out, compute="f32x6", _desc_only=True)
    d.epilogue = _lib.EPI_LINEAR_STATS
    d.ep_stats, d.ep_stats_ld = out.data_ptr(), 5
    with pytest.raises(RuntimeError, match="invalid argument"):
        _lib.check(_lib.load().cgn_gemm_f32(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.byref(d)), "cgn_gemm_f32")


@pytest.mark.parametrize("kind", ["nt", "nn", "tn", "tn_history", "mixed"])
def test_gemm_x6_group(dev, kind):
    """Sibling X6 products in one 128-tile launch (the encoder's two W_ih products, the per-layer LSTM weight gradients
    over history buffers): each member equals its own single launch bit for bit where the plan is the same, and the fp64
    product to 2e-6 always; a group whose members do not share a kernel variant ("mixed") is issued member by member."""
    ops = _ops()
    from ldrs_amd.hip._lib import Axis
    g = torch.Generator().manual_seed(11)
    calls, refs, outs = [], [], []
    if kind == "tn_history":
        Bn, T, H = 24, 20, 64
        for i in range(3):
            dg, hbuf = _mk((Bn, T, 4 * H), g), _mk((Bn, T + 1, H), g)
            refs.append(dg.double().reshape(Bn * T, 4 * H).t() @ hbuf[:, :T].double().reshape(Bn * T, H))
            out = torch.full((4 * H, H), float("nan"), device=dev)
            outs.append(out)
            calls.append((dg.to(dev).view(Bn * T, 4 * H).t(), hbuf.to(dev), out,
                          dict(N=H, K=Bn * T, b_k=Axis(H, T, (T + 1) * H), b_n=Axis(1, 0, 0))))
    else:
        for i in range(3):
            layout = ("nt", "nn", "tn")[i] if kind == "mixed" else kind
            M, N, K = 256 + 132 * i, 384 - 64 * i, 512 + 96 * i
            A, B = _mk((M, K), g), _mk((K, N), g)
            bias = _mk((N,), g).to(dev)
            a = A.to(dev) if layout[0] == "n" else A.t().contiguous().to(dev).t()
            b = B.to(dev) if layout[1] == "n" else B.t().contiguous().to(dev).t()
            out = torch.full((M, N), float("nan"), device=dev)
            calls.append((a, b, out, dict(bias0=bias)))
            refs.append(A.double() @ B.double() + bias.cpu().double())
            outs.append(out)
    ops.gemm_group([(a, b, o, dict(kw, compute="f32x6")) for a, b, o, kw in calls])
    torch.cuda.synchronize()
    for out, ref in zip(outs, refs):
        assert _rel(out, ref) < 2e-6
    if kind == "mixed":  # issued member by member: exactly the single launches
        for (a, b, o, kw), out in zip(calls, outs):
            single = torch.empty_like(out)
            ops.gemm(a, b, single, compute="f32x6", **kw)
            assert torch.equal(single, out)


# ---- the bf16 x bf16 kernel of the bf16 mode (cgn_gemm_bb, cgn_cast_bf16_f32; csrc/bb_gemm.hip) ----------------------
def test_cast_bf16_plain_history_axis_transposed_and_unaligned(dev):
    """The casts round to nearest even exactly as torch does, follow the two-level row axis of a [B, T+1, H] history
    buffer, zero their padding columns, and take unaligned sources (scalar loads)."""
    ops = _ops()
    from ldrs_amd.hip._lib import Axis
    g = torch.Generator().manual_seed(11)
    src = torch.randn(37, 21, 52, generator=g).to(dev)
    rows = src[:, :20].reshape(-1, 52)
    ax = Axis(52, 20, 21 * 52)
    d = ops.cast_bf16(src, ax, 37 * 20, 52)
    assert d.shape == (740, 56) and torch.equal(d[:, :52], rows.bfloat16()) and bool((d[:, 52:] == 0).all())
    dt = ops.cast_bf16(src, ax, 37 * 20, 52, transpose=True)
    assert dt.shape == (52, 744) and torch.equal(dt[:, :740], rows.t().bfloat16()) and bool((dt[:, 740:] == 0).all())
    m = torch.randn(301, 777, generator=g).to(dev)
    v = m[:, 1:]
    d = ops.cast_bf16(v, Axis(777, 0, 0), 301, 776)
    assert torch.equal(d[:, :776], v.bfloat16())
    dt = ops.cast_bf16(v, Axis(777, 0, 0), 301, 776, transpose=True)
    assert torch.equal(dt[:, :301], v.t().bfloat16()) and bool((dt[:, 301:] == 0).all())
    # values that sit half way between two bf16 numbers go to the even one; infinities and NaN survive
    x = torch.tensor([[1.00390625, 1.01171875, -1.00390625, float("inf"), float("nan"), 0.0, -0.0, 3e38]], device=dev)
    d = ops.cast_bf16(x, Axis(8, 0, 0), 1, 8)
    ref = x.bfloat16()
    assert torch.equal(d.view(torch.int16), ref.view(torch.int16))


@pytest.mark.parametrize("M,N,K,kw", [
    (256, 256, 64, {}), (300, 520, 328, {}), (513, 257, 1000, dict(bias=True, beta=0.5, alpha=0.7)),
    (1000, 300, 2048, dict(splitk=3)), (640, 900, 4096, dict(tile=256, splitk=4, bias=True, beta=1.0)),
    (128, 128, 64, dict(tile=128)), (1344, 2048, 512, dict(tile=128)), (1344, 2048, 512, dict(tile=256)),
    (260, 264, 8, dict(tile=256)), (130, 3000, 40, {}), (1, 1, 8, dict(tile=128)), (257, 129, 72, dict(tile=256)),
    # the 128x384 tile (one column tile, fp32 result): full, ragged and narrow N, K tails, split with bias / beta
    (300, 384, 328, dict(tile=384)), (1344, 300, 2000, dict(tile=384, splitk=5, bias=True, beta=1.0)),
    (129, 130, 72, dict(tile=384)), (640, 304, 8, dict(tile=384, bias=True, beta=0.5, alpha=0.7)),
])
def test_gemm_bb_against_fp64(dev, M, N, K, kw):
    """fp32 C: the fp64 product of the bf16 operands up to fp32 summation error (2e-6 of sum |a||b|), exact on small
    integers; both tiles, K tails (K % 32 != 0), one-step products, edge tiles, split-K with bias / beta / alpha."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    ints = (M + N + K) % 2 == 0
    a, b = _mk((M, K), g, ints).to(dev).bfloat16(), _mk((N, K), g, ints).to(dev).bfloat16()
    c0 = _mk((M, N), g, ints).to(dev)
    out = c0.clone()
    bias = kw.get("bias", False)
    b0 = _mk((N,), g, ints).to(dev) if bias else None
    b1 = _mk((N,), g, ints).to(dev) if bias else None
    alpha, beta = (kw.get("alpha", 1.0), kw.get("beta", 0.0)) if not ints else (1.0, float(int(kw.get("beta", 0.0))))
    ops.gemm_bb(a, b, out, alpha=alpha, beta=beta, bias0=b0, bias1=b1, force_tile=kw.get("tile", 0),
                force_splitk=kw.get("splitk", 0))
    ref = alpha * (a.double() @ b.double().t())
    if bias:
        ref = ref + b0.double() + b1.double()
    ref = ref + beta * c0.double()
    if ints:
        assert torch.equal(out.double(), ref)
    else:
        scale = a.double().abs() @ b.double().abs().t() + 1.0
        assert ((out.double() - ref).abs() / scale).max().item() < 2e-6


@pytest.mark.parametrize("M,N,K,tile", [(200, 1000, 520, 128), (777, 1032, 512, 256), (300, 260, 96, 0), (256, 257, 64, 256)])
def test_gemm_bb_bf16_result(dev, M, N, K, tile):
    """bf16 C (the vocabulary projection's logits): the fp32 result rounded once, through the LDS-staged 16-byte row
    stores and the scalar tail of rows whose length is not a multiple of 8."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a, b = torch.randn(M, K, generator=g).to(dev).bfloat16(), torch.randn(N, K, generator=g).to(dev).bfloat16()
    bias = torch.randn(N, generator=g).to(dev)
    ld = (N + 7) // 8 * 8
    for buf in (torch.zeros(M, ld, device=dev, dtype=torch.bfloat16), torch.zeros(M, N + 3, device=dev, dtype=torch.bfloat16)):
        out = buf[:, :N]
        ops.gemm_bb(a, b, out, bias0=bias, force_tile=tile)
        ref32 = torch.empty(M, N, device=dev)
        ops.gemm_bb(a, b, ref32, bias0=bias, force_tile=tile)
        assert torch.equal(out, ref32.bfloat16())
        assert bool((buf[:, N:] == 0).all())  # nothing written past the row's end


def test_gemm_bb_refuses_what_it_cannot_read(dev):
    ops = _ops()
    a = torch.zeros(64, 72, device=dev, dtype=torch.bfloat16)
    b = torch.zeros(64, 72, device=dev, dtype=torch.bfloat16)
    out = torch.zeros(64, 64, device=dev)
    with pytest.raises(ValueError):
        ops.gemm_bb(a, b, out, K=60)  # K % 8
    with pytest.raises(RuntimeError):
        ops.gemm_bb(a[:, 1:65], b[:, :64], out, K=64)  # rows not 16-byte aligned
    with pytest.raises(RuntimeError):
        ops.gemm_bb(a, b, out.bfloat16(), K=64, beta=1.0)  # a bf16 result cannot be accumulated into
    with pytest.raises(RuntimeError):
        ops.gemm_bb(a, b, out, K=64, force_tile=64)
    with pytest.raises(RuntimeError):
        ops.gemm_bb(a, b, out.bfloat16(), K=64, force_tile=384)  # the 128x384 tile writes fp32 only
    wide = torch.zeros(392, 72, device=dev, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError):
        ops.gemm_bb(a, wide, torch.zeros(64, 392, device=dev), K=64, force_tile=384)  # more than one column tile


@pytest.mark.parametrize("layout", ["nt", "nn", "tn", "tt"])
def test_bf16_mode_routes_large_products_through_the_bb_kernel(dev, layout, monkeypatch):
    """ops.gemm in the bf16 mode: with CGN_BB_GEMM (default) a large plain product is cast once and multiplied by the
    bf16 x bf16 kernel; the result equals the kernel that rounds at LDS staging up to fp32 summation order, for every
    operand orientation, with bias and beta, and for a history-buffer A operand."""
    ops = _ops()
    from ldrs_amd.hip._lib import Axis
    g = torch.Generator().manual_seed(5)
    M, N, K = 520, 392, 300
    A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    a = A.to(dev) if layout[0] == "n" else A.t().contiguous().to(dev).t()
    b = B.to(dev) if layout[1] == "n" else B.t().contiguous().to(dev).t()
    bias = torch.randn(N, generator=g).to(dev)
    c0 = torch.randn(M, N, generator=g).to(dev)
    monkeypatch.setattr(ops, "BB_MIN_FLOP", 0.0)
    res = []
    for bb in (False, True):
        monkeypatch.setattr(ops, "BB_GEMM", bb)
        before = len(ops.KEEP) if ops.KEEP is not None else None
        out = c0.clone()
        ops.gemm(a, b, out, bias0=bias, beta=0.5, compute="bf16")
        res.append(out)
    assert _rel(res[1], res[0].double().cpu()) < 2e-6
    assert _rel(res[1], (_bf(A) @ _bf(B) + bias.double().cpu() + 0.5 * c0.double().cpu())) < 2e-6
    if layout == "nt":  # A = the rows (b, t < T) of a [B, T+1, H] history buffer
        Bn, T, H = 26, 20, K
        hist = torch.randn(Bn, T + 1, H, generator=g).to(dev)
        ax = Axis(H, T, (T + 1) * H)
        outs = []
        for bb in (False, True):
            monkeypatch.setattr(ops, "BB_GEMM", bb)
            out = torch.empty(Bn * T, N, device=dev)
            ops.gemm(hist, b, out, M=Bn * T, K=K, a_m=ax, a_k=Axis(1, 0, 0), compute="bf16")
            outs.append(out)
        assert _rel(outs[1], outs[0].double().cpu()) < 2e-6
        ref = _bf(hist[:, :T].reshape(-1, H).cpu()) @ _bf(B)
        assert _rel(outs[1], ref) < 2e-6


@pytest.mark.parametrize("tile", [192, 256])
@pytest.mark.parametrize("M,V,E,ldpad", [(264, 1032, 304, 0), (300, 791, 96, 1), (130, 520, 64, 0)])
def test_gemm_bb_softmax_backward_and_exp_operand(dev, M, V, E, ldpad, tile):
    """The two logits-sized products over bf16 logits on both tiles that carry them: fused softmax / CE backward in place
    with the transposed copy (16-byte row paths and, with odd leading dimensions, the scalar ones), and the exp loader.
    Against fp64 math on the same bf16 inputs."""
    ops = _ops()
    g = torch.Generator().manual_seed(M * 3 + V)
    tau = 0.8
    L0 = (torch.randn(M, V, generator=g) * 2).to(dev).bfloat16()
    buf = torch.zeros(M, V + ldpad, device=dev, dtype=torch.bfloat16)
    logits = buf[:, :V]
    logits.copy_(L0)
    Ld = L0.double()
    lse = torch.logsumexp(Ld / tau, 1)
    lse32 = lse.float()
    P = torch.exp(Ld / tau - lse32.double()[:, None])
    tgt = torch.randint(0, V, (M,), generator=g).to(dev)
    emb = (torch.randn(V, E, generator=g) * 0.5).to(dev).bfloat16()      # [N = V rows, K = E]
    dout = torch.randn(M, E, generator=g).to(dev).bfloat16()
    # exp loader: softmax(L / tau) @ emb (B operand [N = E, K = V])
    if E >= 128:
        # (forced tile, then the planner's own choice: the 128x384 tile for a product of one column tile)
        for ft in (tile, 0):
            out_emb = torch.full((M, E), float("nan"), device=dev)
            ops.gemm_bb(logits if ldpad == 0 else L0, emb.t().contiguous(), out_emb, a_exp=(1.0 / tau, lse32),
                        force_tile=ft)
            assert _rel(out_emb, (P @ emb.double()).cpu()) < 1e-2
    rowdot = torch.randn(M, generator=g).to(dev)
    ce_scale = 79.0 / tau / M
    ct_buf = torch.full((V, M + 8 + ldpad), float("nan"), device=dev, dtype=torch.bfloat16)
    ct = ct_buf[:, :M]
    ops.gemm_bb(dout, emb, logits, c_t=ct, force_tile=tile,
                softmax_bwd=dict(logits=logits, lse=lse32, rowdot=rowdot, target=tgt, inv_tau=1.0 / tau, ce_scale=ce_scale))
    acc = dout.double() @ emb.double().t()
    onehot = torch.zeros(M, V, dtype=torch.float64, device=dev)
    onehot[torch.arange(M, device=dev), tgt] = 1.0
    ref = P * (acc - rowdot.double()[:, None]) / tau + ce_scale * (P - onehot)
    got = logits.double()
    assert ((got - ref).abs() <= ref.abs() * 2.0 ** -7 + 1e-5 * ref.abs().max()).all()
    assert torch.equal(ct, logits.t())
    assert bool((buf[:, V:] == 0).all()) and bool(torch.isnan(ct_buf[:, M:].float()).all())
