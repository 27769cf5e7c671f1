"""GPU parity of the fused launch-bound chains (csrc/fused.hip over the C-ABI) against fp64 CPU PyTorch autograd of the
modules they stand for: the code head (model/cgn.py:83-103, :1076-1080, generator.py:54-56), the c-reconstruction head
behind the frozen text-CNN (cgn.py:1114-1118, discriminator_sentiment.py:66-77) and col2im + zero last step + row dot
(cgn.py:1103-1108). fp32 tolerance 1e-5 relative Frobenius error per output, as the other building blocks."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _ops():
    import ldrs_amd  # noqa: F401
    from ldrs_amd.hip import ops
    return ops


def rel(a, ref):
    a = a.detach().double().cpu()
    ref = ref.detach().double()
    return (a - ref).norm().item() / max(ref.norm().item(), 1e-30)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed * 7919 + sum(shape))
    return torch.randn(*shape, generator=g) * scale


# (B, He, Z, C, Hg, T): BASELINE configs[1] and [3] shapes, a ragged one, one with C = 0 columns absent from nothing
CODE_HEAD_SHAPES = [(64, 512, 64, 1, 512, 21), (256, 1024, 128, 1, 1024, 5), (5, 36, 10, 3, 20, 4), (3, 16, 8, 1, 8, 9)]


@pytest.mark.parametrize("B,He,Z,Cc,Hg,T", CODE_HEAD_SHAPES)
def test_code_head_forward_and_backward(dev, B, He, Z, Cc, Hg, T):
    ops = _ops()
    G4, ZC, E = 4 * Hg, Z + Cc, 12
    cn = rnd(B, 2, He, seed=1).double().requires_grad_(True)           # [B, dir, He] final cell states
    w_mu, w_lv = rnd(Z, 2 * He, seed=2, scale=0.05).double(), rnd(Z, 2 * He, seed=3, scale=0.05).double()
    b_mu, b_lv = rnd(Z, seed=4, scale=0.1).double(), rnd(Z, seed=5, scale=0.1).double()
    eps = rnd(B, Z, seed=6).double()
    c = (rnd(B, Cc, seed=7) > 0).double()
    w_ih = rnd(G4, E + ZC, seed=8, scale=0.1).double()                  # generator W_ih_l0, [z,c] columns behind E
    dg0 = rnd(B, T, G4, seed=9, scale=0.1).double()
    kld_w = 0.37

    # reference: autograd through context -> mu, logvar -> z -> [z,c] -> gates (same [z,c] at every timestep) + KLD
    ctx = torch.cat([cn[:, 0], cn[:, 1]], 1)
    mu, lv = ctx @ w_mu.t() + b_mu, ctx @ w_lv.t() + b_lv
    std = torch.exp(0.5 * lv)
    z = eps * std + mu
    zc_live = torch.cat([z, c], 1)
    zg = zc_live @ w_ih[:, E:].t()
    kld = (-0.5 * torch.sum(lv - mu.pow(2) - lv.exp() + 1, 1)).mean()
    mu.retain_grad(), lv.retain_grad()
    ((zg.unsqueeze(1) * dg0).sum() + kld_w * B * kld).backward()
    dzc_ref = dg0.sum(1) @ w_ih[:, E:]

    f = lambda t: t.detach().float().to(dev).contiguous()
    cstate = f(cn)                                                      # rows of 2*He: strided views as the engine has
    cf, cr = cstate[:, 0], cstate[:, 1]
    w2, b2 = f(torch.cat([w_mu, w_lv], 0)), f(torch.cat([b_mu, b_lv], 0))
    wih_d = torch.zeros(G4, ops.pad4(E + ZC) + 4, device=dev)[:, :E + ZC]
    wih_d.copy_(f(w_ih))
    wzc = wih_d[:, E:]
    assert ops.code_head_eligible(B, He, Z, Cc, w2, wzc)
    out = {n: torch.full(s, float("nan"), device=dev) for n, s in
           (("ctx", (B, 2 * He)), ("mulv", (B, 2 * Z)), ("zc", (B, ZC)), ("stdv", (B, Z)), ("kld", (1,)), ("zg", (B, G4)))}
    for rep in range(2):  # twice: the ticket re-arms itself
        ops.code_head_fwd(cf, cr, w2, b2, f(eps), f(c), wzc, out["ctx"], out["mulv"], out["zc"], out["stdv"], out["kld"],
                          out["zg"])
        torch.cuda.synchronize()
        assert rel(out["ctx"], ctx) == 0.0
        assert rel(out["mulv"], torch.cat([mu, lv], 1)) < TOL
        assert rel(out["zc"], zc_live) < TOL and rel(out["stdv"], std) < TOL
        assert rel(out["zg"], zg) < TOL
        assert rel(out["kld"], kld.reshape(1)) < TOL

    bo = {n: torch.full(s, float("nan"), device=dev) for n, s in
          (("dsum", (B, G4)), ("dzc", (B, ZC)), ("dmulv", (B, 2 * Z)), ("dcf", (B, He)), ("dcr", (B, He)))}
    ops.code_head_bwd(f(dg0), wzc, out["mulv"], f(eps), w2, kld_w, bo["dsum"], bo["dzc"], bo["dmulv"], bo["dcf"], bo["dcr"])
    torch.cuda.synchronize()
    assert rel(bo["dsum"], dg0.sum(1)) < TOL
    assert rel(bo["dzc"], dzc_ref) < TOL
    assert rel(bo["dmulv"], torch.cat([mu.grad, lv.grad], 1)) < TOL
    assert rel(bo["dcf"], cn.grad[:, 0]) < TOL and rel(bo["dcr"], cn.grad[:, 1]) < TOL
    # the replayed graph reads the KLD weight from device memory
    dyn = torch.tensor([kld_w], device=dev)
    bo2 = {n: torch.empty_like(t) for n, t in bo.items()}
    ops.code_head_bwd(f(dg0), wzc, out["mulv"], f(eps), w2, 123.0, bo2["dsum"], bo2["dzc"], bo2["dmulv"], bo2["dcf"],
                      bo2["dcr"], dyn=dyn)
    torch.cuda.synchronize()
    for n in bo:
        assert torch.equal(bo[n], bo2[n]), n


@pytest.mark.parametrize("B,L,Ks,Cos", [(64, 20, (3, 4, 5), (100, 100, 100)), (7, 9, (2, 3), (5, 12)),
                                          (256, 20, (3, 4, 5), (100, 100, 100)), (1, 6, (6,), (9,))])
def test_textcnn_head(dev, B, L, Ks, Cos):
    ops = _ops()
    Fdim = sum(Cos)
    Ps = [L - K + 1 for K in Ks]
    convs = [rnd(B, P, Co, seed=10 + i).double().requires_grad_(True) for i, (P, Co) in enumerate(zip(Ps, Cos))]
    W, bias = rnd(2, Fdim, seed=20, scale=0.1).double(), rnd(2, seed=21, scale=0.1).double()
    y = (rnd(B, seed=22) > 0).long()
    w_ce = 0.1
    feat = torch.cat([F.relu(cv).max(1)[0] for cv in convs], 1)
    ce = F.cross_entropy(feat @ W.t() + bias, y)
    (w_ce * ce).backward()

    f = lambda t: t.detach().float().to(dev).contiguous()
    zc = torch.zeros(B, 3, device=dev)                                  # labels = a strided float column, as [z,c]'s
    zc[:, 2] = y.float().to(dev)
    feat_d = torch.empty(B, Fdim, device=dev)
    arg_d = torch.empty(B, Fdim, device=dev, dtype=torch.int32)
    stats = torch.full((1,), float("nan"), device=dev)
    for rep in range(2):
        dconvs = ops.textcnn_head([f(cv).view(B * P, Co) for cv, P, Co in zip(convs, Ps, Cos)], Ps, B, f(W), f(bias),
                                  zc[:, 2], w_ce, feat_d, arg_d, stats)
        torch.cuda.synchronize()
        assert rel(feat_d, feat) == 0.0
        assert rel(stats, ce.reshape(1)) < TOL
        for dc, cv, P, Co in zip(dconvs, convs, Ps, Cos):
            assert rel(dc.view(B, P, Co), cv.grad) < TOL


@pytest.mark.parametrize("B,L,T,E,Ks", [(64, 20, 21, 300, (3, 4, 5)), (5, 7, 8, 37, (2, 3)), (3, 6, 6, 1100, (1, 6)),
                                        (256, 20, 21, 300, (3, 4, 5))])
@pytest.mark.parametrize("beta", [0.0, 1.0])
def test_col2im_rowdot(dev, B, L, T, E, Ks, beta):
    """dx rows t < L: the windows' gradients summed back (the transpose of unfold); rows L <= t < T: zero (+ beta dx)."""
    ops = _ops()
    dcols = [rnd(B * (L - K + 1), K * E, seed=30 + K).double() for K in Ks]
    dx0 = rnd(B * T, E, seed=40).double()
    other = rnd(B * T, E + 3, seed=41).double()[:, :E]
    ref = torch.zeros(B, T, E, dtype=torch.float64)
    for K, dl in zip(Ks, dcols):
        d4 = dl.view(B, L - K + 1, K, E)
        for k in range(K):
            ref[:, k:k + L - K + 1] += d4[:, :, k]
    ref = ref.view(B * T, E) + beta * dx0
    f = lambda t: t.float().to(dev)
    dx = f(dx0).contiguous()
    oth = f(rnd(B * T, E + 3, seed=41))[:, :E]
    rowdot = torch.full((B * T,), float("nan"), device=dev)
    ops.col2im_rowdot(list(Ks), [f(d).contiguous() for d in dcols], B, L, T, E, dx, oth, rowdot, beta=beta)
    torch.cuda.synchronize()
    assert rel(dx, ref) < TOL
    assert rel(rowdot, (ref * other).sum(1)) < TOL
    if T > L and beta == 0.0:
        assert dx.view(B, T, E)[:, L:].abs().max().item() == 0.0
