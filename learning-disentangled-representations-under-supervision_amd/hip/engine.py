"""Forward/backward building blocks of the CGN step, composed from the C-ABI kernels (hip/ops.py).

Every block is a pair  fwd(...) -> (outputs, saved)  /  bwd(saved, upstream grads, grad buffers) and calls only
hand-written HIP kernels; torch supplies device buffers. Parameter gradients are WRITTEN (beta=0) or ACCUMULATED
(beta=1) into caller-provided tensors (views of the flat gradient arenas, hip/arena.py), which is what lets the
training step reproduce the reference's backward order and optimizer grouping without autograd.

Layouts are batch-major like the reference API ([B,L,Din] inputs, [B,T,V] logits); LSTM histories are [B,T+1,H]
buffers addressed in place by the GEMMs (two-level axes), never copied.
"""
import torch

from . import ops
from ._lib import ACT_RELU, ACT_TANH, Axis


def _new(shape, dev, dtype=torch.float32):
    return ops.alloc(shape, dev, dtype)


LIN = Axis(1, 0, 0)


# ------------------------------------------------------------------------------------------------------------------
# linear
# ------------------------------------------------------------------------------------------------------------------
def linear_fwd(x2d, W, b, out):
    """out = x2d @ W^T + b   (nn.Linear)."""
    return ops.gemm(x2d, W.t(), out, bias0=b)


def linear_bwd(x2d, W, dy, dW, db, dx=None, beta=0.0, dx_beta=0.0):
    ops.gemm(dy.t(), x2d, dW, beta=beta)
    if db is not None:
        ops.colsum(dy, db, beta=beta)
    if dx is not None:
        ops.gemm(dy, W, dx, beta=dx_beta)


# ------------------------------------------------------------------------------------------------------------------
# char-TDNN (selfModules/tdnn.py:18-48) incl. the char-embedding lookup (selfModules/embedding.py:44-51)
# ------------------------------------------------------------------------------------------------------------------
def tdnn_conv_fwd(xw, N, Wl, kernels):
    """xw [N*W, Cc] embedded characters -> feat [N, S] (contiguous). kernels: list of [od, Cc, kW]."""
    Cc = xw.shape[1]
    dev = xw.device
    S = sum(k.shape[0] for k in kernels)
    feat = _new((N, S), dev)
    arg = _new((N, S), dev, torch.int32)
    col = 0
    calls, convs, Ps, cols = [], [], [], []
    for k in kernels:
        od, _, kW = k.shape
        if kW > Wl:
            raise RuntimeError("TDNN kernel width %d exceeds max_word_len %d" % (kW, Wl))
        P = Wl - kW + 1
        conv = _new((N * P, od), dev)
        # row (n,p) = xw[n, p:p+kW, :] (contiguous kW*Cc floats, order (j,c)); weight element (o, c, j) -> k = j*Cc + c
        # compute="f32" also in the bf16 mode: these products feed a max-over-time selection, whose argmax would flip
        # under bf16 operand rounding (0.3 % of the step's FLOPs)
        calls.append((xw, k, conv, dict(M=N * P, N=od, K=kW * Cc, a_m=Axis(Cc, P, Wl * Cc), a_k=LIN,
                                        b_k=Axis(kW, Cc, 1), b_n=Axis(Cc * kW, 0, 0), compute="f32")))
        convs.append(conv), Ps.append(P), cols.append(col)
        col += od
    # all widths in one grouped GEMM launch + one pooling launch (the reference loops over F.conv1d, tdnn.py:41-44)
    ops.gemm_group(calls)
    for i in range(0, len(convs), 8):
        ops.act_maxpool_group(convs[i:i + 8], N, Ps[i:i + 8], ACT_TANH, feat, cols[i:i + 8], arg)
    return feat, dict(xw=xw, feat=feat, arg=arg, N=N, W=Wl, Cc=Cc)


def tdnn_conv_bwd(saved, kernels, dfeat, dkernels, need_dx, beta=0.0):
    """dfeat [N,S] contiguous -> kernel grads (into dkernels) and, if asked, d(xw) [N*W, Cc]."""
    N, Wl, Cc = saved["N"], saved["W"], saved["Cc"]
    Ks = [k.shape[2] for k in kernels]
    Cos = [k.shape[0] for k in kernels]
    if not need_dx:
        if dkernels is not None:
            ops.conv_bwd_dw(Ks, Cos, ACT_TANH, 1, dkernels, None, dfeat, saved["feat"], saved["arg"], saved["xw"],
                            Wl * Cc, N, Wl, Cc, beta=beta)
        return None
    # both gradients from the scattered conv-output gradients: two grouped MFMA launches (the sparse dW kernel walks
    # [N, S] gradient columns with a stride of S floats per lane: 66 us at cfg2 against ~20 us for the dense product)
    dxw = _new((N * Wl, Cc), dfeat.device)
    ops.conv_bwd_dx_dense(Ks, Cos, ACT_TANH, 1, list(kernels), dfeat, saved["feat"], saved["arg"], N, Wl, Cc, dxw,
                          Wl * Cc, dW=dkernels, x=saved["xw"], beta=beta)
    return dxw


def tdnn_fwd(char_idx, char_table, kernels):
    """char_idx [N, W] int64 -> feat [N, S]: char-embedding lookup (selfModules/embedding.py:44-49) + TDNN."""
    N, Wl = char_idx.shape
    xw = _new((N * Wl, char_table.shape[1]), char_table.device)
    flat = char_idx.reshape(-1)
    ops.gather_rows(char_table, flat, xw)
    feat, saved = tdnn_conv_fwd(xw, N, Wl, kernels)
    saved["char_idx"] = flat
    return feat, saved


def tdnn_bwd(saved, kernels, dfeat, dkernels, dchar_table, beta=0.0):
    dxw = tdnn_conv_bwd(saved, kernels, dfeat, dkernels, need_dx=True, beta=beta)
    ops.embedding_bwd(saved["char_idx"], dxw, dchar_table, beta=beta)


# ------------------------------------------------------------------------------------------------------------------
# highway (selfModules/highway.py:27-46), f = relu as used by the encoder (model/encoder.py:15)
# ------------------------------------------------------------------------------------------------------------------
def highway_fwd(x, layers):
    """x [N,S]; layers: list of dict(nw, nb, lw, lb, gw, gb[, wcat, bcat]). Returns y and the per-layer saved
    (input, pre). Activations live in buffers whose rows/blocks start on 16-byte boundaries (S = 825 is odd in the
    reference). With wcat [3*Sp, S] / bcat [3*Sp] (the three weights stacked in the arena, hip/arena.py `fused`) the
    N|L|G pre-activations are ONE GEMM."""
    N, S = x.shape
    Sp = ops.pad4(S)
    saved = []
    for p in layers:
        pre = _new((N, 3 * Sp), x.device)
        # compute="f32" also in the bf16 mode: the nonlinear branch feeds a ReLU whose gates flip under bf16 operand
        # rounding, and the gradients below the highway (char-TDNN, char embedding) amplify the perturbation of the
        # other two branches: measured against the oracle (tests/diag_bf16_sensitivity.py, tests/test_step_gpu.py) they go
        # 3e-2 with all three branches in bf16, 1.6e-2 with the ReLU branch alone in fp32, 6e-3 with all three in fp32
        if p.get("wcat") is not None:
            ops.gemm(x, p["wcat"].t(), pre, bias0=p["bcat"], compute="f32")
        else:
            ops.gemm(x, p["nw"].t(), pre[:, 0:S], bias0=p["nb"], compute="f32")
            ops.gemm(x, p["lw"].t(), pre[:, Sp:Sp + S], bias0=p["lb"], compute="f32")
            ops.gemm(x, p["gw"].t(), pre[:, 2 * Sp:2 * Sp + S], bias0=p["gb"], compute="f32")
        y = ops.new2d(N, S, x.device)
        ops.highway_gate_fwd(pre, S, y)
        saved.append((x, pre))
        x = y
    return x, saved


def highway_bwd(saved, layers, dy, grads, need_dx, beta=0.0, later=None):
    """grads: list of dict(nw, nb, lw, lb, gw, gb[, wcat, bcat]) gradient tensors (may be None to skip parameter
    grads). later: optional callable(fn) that runs the parameter-gradient work fn() now or later, on another stream
    (CGNStep.later); default: at once.
    The gate backward writes zeros into the pad columns of dpre, so the stacked forms (one dW GEMM, one column sum,
    one dX GEMM over K = 3*Sp) see exact zeros for the pad rows of wcat."""
    later = later or (lambda fn: fn())
    dx = None
    for li in range(len(layers) - 1, -1, -1):
        x, pre = saved[li]
        p = layers[li]
        N, S = x.shape
        Sp = pre.shape[1] // 3
        dpre = _new((N, 3 * Sp), x.device)
        ops.highway_gate_bwd(pre, S, dy, dpre)
        views = (dpre[:, 0:S], dpre[:, Sp:Sp + S], dpre[:, 2 * Sp:2 * Sp + S])
        if grads is not None:
            g = grads[li]

            def param_grads(g=g, dpre=dpre, x=x, views=views):
                if g.get("wcat") is not None:
                    ops.gemm(dpre.t(), x, g["wcat"], beta=beta)
                    ops.colsum(dpre, g["bcat"], beta=beta)
                else:
                    for v, wn, bn in zip(views, ("nw", "lw", "gw"), ("nb", "lb", "gb")):
                        ops.gemm(v.t(), x, g[wn], beta=beta)
                        ops.colsum(v, g[bn], beta=beta)

            later(param_grads)
        if li > 0 or need_dx:
            dx = ops.new2d(N, S, x.device)
            if p.get("wcat") is not None:
                ops.gemm(dpre, p["wcat"], dx)
            else:
                ops.gemm(views[0], p["nw"], dx)
                ops.gemm(views[1], p["lw"], dx, beta=1.0)
                ops.gemm(views[2], p["gw"], dx, beta=1.0)
            dy = dx
    return dx


# ------------------------------------------------------------------------------------------------------------------
# LSTM (nn.LSTM, batch_first, zero or given initial state)
# ------------------------------------------------------------------------------------------------------------------
class LstmSeq:
    """One direction of one layer: buffers + addressing helpers."""

    def __init__(self, B, T, H, dev, reverse, need_xg=True):
        self.B, self.T, self.H, self.reverse = B, T, H, reverse
        self.xg = _new((B, T, 4 * H), dev) if need_xg else None  # layer 1 of a fused two-layer launch has none
        self.hbuf = _new((B, T + 1, H), dev)
        self.cbuf = _new((B, T + 1, H), dev)
        self.gates = _new((B, T, 4 * H), dev)
        self.xg_extra = None
        self.init_zero = False
        self.init_slot = T if reverse else 0
        self.final_slot = 0 if reverse else T

    def set_initial(self, h0=None, c0=None):
        if h0 is None and c0 is None:
            self.init_zero = True  # the LSTM entry point zeroes the initial slots itself (no extra launches)
            return
        self.init_zero = False
        for buf, init in ((self.hbuf, h0), (self.cbuf, c0)):
            slot = buf[:, self.init_slot, :]
            if init is None:
                ops.copy2d(slot, slot, scale=0.0)  # zero the slot with a kernel (no torch arithmetic on the path)
            else:
                ops.copy2d(init, slot)

    # emitted h as a GEMM operand with rows (b,t): element (m, k) = hbuf[b, t + off, k]
    def hout_operand(self):
        off = 0 if self.reverse else 1
        base = self.hbuf.view(-1)[off * self.H:]
        return base, Axis(self.H, self.T, (self.T + 1) * self.H)

    def hprev_operand(self):
        off = 1 if self.reverse else 0
        base = self.hbuf.view(-1)[off * self.H:]
        return base, Axis(self.H, self.T, (self.T + 1) * self.H)

    def h_final(self):
        return self.hbuf[:, self.final_slot, :]

    def c_final(self):
        return self.cbuf[:, self.final_slot, :]

    def fwd_desc(self, w_hh):
        return dict(xg=self.xg, xg_extra=self.xg_extra, w_hh=w_hh, hbuf=self.hbuf, cbuf=self.cbuf, gates=self.gates,
                    reverse=self.reverse, init_zero=self.init_zero)


def lstm_run_fwd(seqs, w_hhs):
    s0 = seqs[0]
    ops.lstm_seq_fwd([s.fwd_desc(w) for s, w in zip(seqs, w_hhs)], s0.B, s0.T, s0.H)


def lstm_run_bwd(seqs, w_hhs, dh_exts, dh_seeds, dc_states):
    """Returns dgates per sequence. dh_exts: list of None | (tensor, batch_stride, time_stride)."""
    s0 = seqs[0]
    dev = s0.xg.device
    descs, outs = [], []
    persistent = ops.lstm_persistent_group(len(seqs), s0.B, s0.T, s0.H, dev) > 0
    for s, w, ext, seed, dc in zip(seqs, w_hhs, dh_exts, dh_seeds, dc_states):
        if persistent and w.is_contiguous():
            wt = None  # the persistent kernel loads its W_hh slice once, straight from the untransposed weight
        else:
            wt = _new((s.H, 4 * s.H), dev)
            ops.transpose(w, wt)
        dg = _new((s.B, s.T, 4 * s.H), dev)
        d = dict(w_hh_t=wt, w_hh=w if wt is None else None, cbuf=s.cbuf, gates=s.gates, dgates=dg, dc_state=dc, dh_seed=seed, reverse=s.reverse)
        if ext is not None:
            d["dh_ext"], d["dh_bs"], d["dh_ts"] = ext
        descs.append(d)
        outs.append(dg)
    ops.lstm_seq_bwd(descs, s0.B, s0.T, s0.H)
    return outs


def lstm_param_grads(seq, dgates, x_operand, dW_ih, dW_hh, db_ih, db_hh, beta=0.0):
    """x_operand: (tensor, k_axis) describing x rows (b,t) as the B operand [K=B*T, Din] of dW_ih = dG^T x."""
    B, T, H = seq.B, seq.T, seq.H
    dg2 = dgates.view(B * T, 4 * H)
    xt, xk = x_operand
    hp, hk = seq.hprev_operand()
    with ops.bb_memo():  # (bf16 mode: the transposed bf16 copy of dG serves both products)
        if dW_ih is not None:
            ops.gemm(dg2.t(), xt, dW_ih, N=dW_ih.shape[1], K=B * T, b_k=xk, b_n=LIN, beta=beta)
        ops.gemm(dg2.t(), hp, dW_hh, N=H, K=B * T, b_k=hk, b_n=LIN, beta=beta)
    ops.colsum(dg2, db_ih, beta=beta)
    if beta == 0.0:
        ops.copy2d(db_ih.view(1, -1), db_hh.view(1, -1))  # both biases enter the gates additively: same gradient
    else:
        ops.colsum(dg2, db_hh, beta=beta)
    return dg2


def lstm_param_grads_multi(items, beta=0.0):
    """lstm_param_grads for several sequences / layers at once: items = [(seq, dgates, x_operand, dW_ih, dW_hh, db_ih,
    db_hh), ...]. The weight-gradient products of one kind (over plain x rows; over [B,T+1,H] history buffers) go out as
    ONE grouped launch each (cgn_gemm_group_f32: an X6 128-tile group where every member qualifies, else the library
    issues them one by one): a 2048 x 512 x 1344 product alone has 64-128 workgroups for 256 CUs."""
    plain, hist = [], []
    for seq, dgates, (xt, xk), dW_ih, dW_hh, db_ih, db_hh in items:
        B, T, H = seq.B, seq.T, seq.H
        dg2 = dgates.view(B * T, 4 * H)
        if dW_ih is not None:
            (hist if xk.rpg else plain).append((dg2.t(), xt, dW_ih, dict(N=dW_ih.shape[1], K=B * T, b_k=xk, b_n=LIN,
                                                                         beta=beta)))
        hp, hk = seq.hprev_operand()
        hist.append((dg2.t(), hp, dW_hh, dict(N=H, K=B * T, b_k=hk, b_n=LIN, beta=beta)))
    for calls in (hist, plain):
        if len(calls) == 1:
            a, b, out, kw = calls[0]
            ops.gemm(a, b, out, **kw)
        elif calls:
            ops.gemm_group(calls)
    for seq, dgates, _, _, _, db_ih, db_hh in items:
        dg2 = dgates.view(seq.B * seq.T, 4 * seq.H)
        ops.colsum(dg2, db_ih, beta=beta)
        if beta == 0.0:
            ops.copy2d(db_ih.view(1, -1), db_hh.view(1, -1))
        else:
            ops.colsum(dg2, db_hh, beta=beta)


# ------------------------------------------------------------------------------------------------------------------
# text-CNN discriminator (model/discriminator_sentiment.py:56-78)
# ------------------------------------------------------------------------------------------------------------------
def textcnn_conv_fwd(x, B, L, E, sent_stride, conv_w, conv_b):
    """The convolution products only (all filter widths in one grouped launch): returns ([B*P_w, Co_w] per width, Ps)."""
    dev = x.device
    calls, convs, Ps = [], [], []
    for w, b in zip(conv_w, conv_b):
        Co, _, K, Ew = w.shape
        if K > L:
            raise RuntimeError("Kernel size can't be greater than actual input size (K=%d, L=%d)" % (K, L))
        P = L - K + 1
        conv = _new((B * P, Co), dev)
        calls.append((x, w.view(Co, K * E).t(), conv, dict(M=B * P, K=K * E, a_m=Axis(E, P, sent_stride), a_k=LIN,
                                                            bias0=b, compute="f32")))
        convs.append(conv), Ps.append(P)
    ops.gemm_group(calls)
    return convs, Ps


def textcnn_features_fwd(x, B, L, E, sent_stride, conv_w, conv_b, grouped=True):
    """x: tensor whose element (b,t,e) sits at b*sent_stride + t*E + e. Returns feat [B,F], arg [B,F]."""
    dev = x.device
    Cos = [w.shape[0] for w in conv_w]
    F = sum(Cos)
    feat = _new((B, F), dev)
    arg = _new((B, F), dev, torch.int32)
    col = 0
    calls, convs, Ps, cols = [], [], [], []
    for w, b in zip(conv_w, conv_b):
        Co, _, K, Ew = w.shape
        if K > L:
            raise RuntimeError("Kernel size can't be greater than actual input size (K=%d, L=%d)" % (K, L))
        P = L - K + 1
        conv = _new((B * P, Co), dev)
        # fp32 products in both compute modes: ReLU + max-over-time selections follow (see tdnn_conv_fwd)
        calls.append((x, w.view(Co, K * E).t(), conv, dict(M=B * P, K=K * E, a_m=Axis(E, P, sent_stride), a_k=LIN,
                                                            bias0=b, compute="f32")))
        convs.append(conv), Ps.append(P), cols.append(col)
        col += Co
    if grouped:
        # all filter widths in one grouped GEMM launch + one pooling launch (discriminator_sentiment.py:64-70)
        ops.gemm_group(calls)
        for i in range(0, len(convs), 8):
            ops.act_maxpool_group(convs[i:i + 8], B, Ps[i:i + 8], ACT_RELU, feat, cols[i:i + 8], arg)
    else:
        for (a_, b_, c_, kw), P, c0 in zip(calls, Ps, cols):
            ops.gemm(a_, b_, c_, **kw)
            ops.act_maxpool(c_, B, P, ACT_RELU, feat, c0, arg)
    return feat, arg
