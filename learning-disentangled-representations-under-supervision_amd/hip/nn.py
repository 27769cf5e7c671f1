"""nn.Module parameter containers with HIP forward/backward (torch.autograd.Function over hip/engine.py).

These give the reference's modules (the nn.Linear / nn.LSTM / nn.Conv2d users) the same parameter names, shapes and
default initialisation, so `state_dict()` keys match the reference's checkpoints, while every forward and backward
runs the hand-written kernels. The fused training step (hip/step.py) bypasses autograd and calls the same engine
blocks directly.
"""
import math

import torch
import torch.nn as nn

from . import engine, ops
from ._lib import Axis

LIN = Axis(1, 0, 0)


def need_cuda(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s: the CGN modules run only on an MI355X (HIP) device; there is no CPU path. "
                           "Move the module and its inputs with .cuda()." % what)


def f32c(t):
    """Contiguous fp32 tensor; strided 2-D/3-D views are packed by a HIP copy kernel."""
    if t.dtype != torch.float32:
        raise TypeError("fp32 tensors expected, got %s" % t.dtype)
    if t.is_contiguous():
        return t
    out = torch.empty(t.shape, dtype=torch.float32, device=t.device)
    if t.dim() == 2 and t.stride(1) == 1:
        return ops.copy2d(t, out)
    if t.dim() == 3 and t.stride(2) == 1:
        return ops.copy3d(t, out)
    raise ValueError("unsupported memory layout %s / %s" % (tuple(t.shape), t.stride()))


def new(shape, dev):
    return torch.empty(shape, dtype=torch.float32, device=dev)


# ------------------------------------------------------------------------------------------------------------------
class _LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, W, b):
        need_cuda(x, "Linear")
        x2 = f32c(x.reshape(-1, x.shape[-1]))
        out = new((x2.shape[0], W.shape[0]), x.device)
        engine.linear_fwd(x2, W, b, out)
        ctx.save_for_backward(x2, W)
        ctx.xshape = x.shape
        return out.view(*x.shape[:-1], W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        dy2 = f32c(dy.reshape(-1, dy.shape[-1]))
        dW, db = torch.empty_like(W), new((W.shape[0],), W.device)
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        engine.linear_bwd(x2, W, dy2, dW, db, dx)
        return (dx.view(ctx.xshape) if dx is not None else None), dW, db


class Linear(nn.Module):
    """nn.Linear replacement (same parameter names / default init)."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return _LinearFn.apply(x, self.weight, self.bias)


# ------------------------------------------------------------------------------------------------------------------
class _LSTMFn(torch.autograd.Function):
    """Multi-layer, optionally bidirectional, batch_first LSTM with optional initial state.
    args: x, h0, c0 (tensors or None), num_layers, bidirectional, then (w_ih, w_hh, b_ih, b_hh) per layer & direction.
    Gradients flow to x and the weights (not to h0/c0: the reference never differentiates through them)."""

    @staticmethod
    def forward(ctx, x, h0, c0, num_layers, bidirectional, *weights):
        need_cuda(x, "LSTM")
        B, T, _ = x.shape
        ndir = 2 if bidirectional else 1
        H = weights[1].shape[1]
        dev = x.device
        inp = f32c(x).view(B * T, -1)
        layers, inputs = [], []
        h_n, c_n = new((num_layers * ndir, B, H), dev), new((num_layers * ndir, B, H), dev)
        out = None
        for l in range(num_layers):
            seqs, whh = [], []
            for d in range(ndir):
                k = l * ndir + d
                w_ih, w_hh, b_ih, b_hh = weights[k * 4:k * 4 + 4]
                s = engine.LstmSeq(B, T, H, dev, reverse=(d == 1))
                ops.gemm(inp, w_ih.t(), s.xg.view(B * T, 4 * H), bias0=b_ih, bias1=b_hh)
                s.set_initial(None if h0 is None else f32c(h0[k]), None if c0 is None else f32c(c0[k]))
                seqs.append(s)
                whh.append(w_hh)
            engine.lstm_run_fwd(seqs, whh)
            out = new((B, T, ndir * H), dev)
            for d, s in enumerate(seqs):
                ops.copy3d(s.hbuf[:, :T] if s.reverse else s.hbuf[:, 1:], out[:, :, d * H:(d + 1) * H])
                ops.copy2d(s.h_final(), h_n[l * ndir + d])
                ops.copy2d(s.c_final(), c_n[l * ndir + d])
            layers.append(seqs)
            inputs.append(inp)
            inp = out.view(B * T, ndir * H)
        ctx.layers, ctx.inputs, ctx.weights = layers, inputs, weights
        ctx.dims = (B, T, H, ndir, num_layers)
        return out, h_n, c_n

    @staticmethod
    def backward(ctx, d_out, d_hn, d_cn):
        B, T, H, ndir, num_layers = ctx.dims
        weights = ctx.weights
        dev = weights[0].device
        grads = [None] * len(weights)
        d_out = f32c(d_out) if d_out is not None else None
        d_hn = f32c(d_hn) if d_hn is not None else None
        d_cn = f32c(d_cn) if d_cn is not None else None
        for l in range(num_layers - 1, -1, -1):
            seqs = ctx.layers[l]
            whh = [weights[(l * ndir + d) * 4 + 1] for d in range(ndir)]
            exts, seeds, dcs = [], [], []
            for d in range(ndir):
                k = l * ndir + d
                exts.append(None if d_out is None else (d_out.view(-1)[d * H:], T * ndir * H, ndir * H))
                seeds.append(d_hn[k] if d_hn is not None else None)
                dc = new((B, H), dev)
                if d_cn is not None:
                    ops.copy2d(d_cn[k], dc)
                else:
                    ops.fill(dc, 0.0)
                dcs.append(dc)
            dgs = engine.lstm_run_bwd(seqs, whh, exts, seeds, dcs)
            inp = ctx.inputs[l]
            Din = inp.shape[1]
            dx = new((B * T, Din), dev)
            for d, (s, dg) in enumerate(zip(seqs, dgs)):
                base = (l * ndir + d) * 4
                g = [torch.empty_like(weights[base + i]) for i in range(4)]
                dg2 = engine.lstm_param_grads(s, dg, (inp, Axis(Din, 0, 0)), g[0], g[1], g[2], g[3])
                grads[base:base + 4] = g
                ops.gemm(dg2, weights[base], dx, beta=(1.0 if d > 0 else 0.0))
            d_out = dx.view(B, T, Din)
        return (d_out, None, None, None, None) + tuple(grads)


class LSTM(nn.Module):
    """nn.LSTM replacement: batch_first, `num_layers`, optional bidirectional; parameter names weight_ih_l{k}[_reverse]
    etc. and U(-1/sqrt(H), 1/sqrt(H)) init as torch. forward(x, state=None) -> (out, (h_n, c_n))."""

    def __init__(self, input_size, hidden_size, num_layers=1, batch_first=True, bidirectional=False):
        super().__init__()
        if not batch_first:
            raise NotImplementedError("the reference uses batch_first=True everywhere")
        if hidden_size % 4:
            raise ValueError("hidden_size must be a multiple of 4")
        self.input_size, self.hidden_size = input_size, hidden_size
        self.num_layers, self.bidirectional = num_layers, bidirectional
        ndir = 2 if bidirectional else 1
        k = 1.0 / math.sqrt(hidden_size)
        for l in range(num_layers):
            din = input_size if l == 0 else hidden_size * ndir
            for d in range(ndir):
                sfx = "_l%d%s" % (l, "_reverse" if d == 1 else "")
                for name, shape in (("weight_ih", (4 * hidden_size, din)), ("weight_hh", (4 * hidden_size, hidden_size)),
                                    ("bias_ih", (4 * hidden_size,)), ("bias_hh", (4 * hidden_size,))):
                    self.register_parameter(name + sfx, nn.Parameter(torch.empty(shape).uniform_(-k, k)))

    def flat_weights(self):
        ndir = 2 if self.bidirectional else 1
        ws = []
        for l in range(self.num_layers):
            for d in range(ndir):
                sfx = "_l%d%s" % (l, "_reverse" if d == 1 else "")
                ws += [getattr(self, n + sfx) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
        return ws

    def forward(self, x, state=None):
        h0, c0 = (None, None) if state is None else state
        out, h_n, c_n = _LSTMFn.apply(x, h0, c0, self.num_layers, self.bidirectional, *self.flat_weights())
        return out, (h_n, c_n)


# ------------------------------------------------------------------------------------------------------------------
class Conv2dParams(nn.Module):
    """Parameter holder for nn.Conv2d(1, Co, (K, D)) (same names/shapes/init); the convolution runs as an implicit
    GEMM inside the text-CNN block."""

    def __init__(self, Co, K, D):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(Co, 1, K, D))
        self.bias = nn.Parameter(torch.empty(Co))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1 / math.sqrt(K * D)
        nn.init.uniform_(self.bias, -bound, bound)


class _TextCNNFn(torch.autograd.Function):
    """x [B,L,E] -> logits [B,2]; args after x: mask (or None), fc_w, fc_b, then (conv_w, conv_b) per width."""

    @staticmethod
    def forward(ctx, x, mask, fc_w, fc_b, *convs):
        need_cuda(x, "Sentiment_CNN")
        B, L, E = x.shape
        xc = f32c(x)
        cw, cb = list(convs[0::2]), list(convs[1::2])
        feat, arg = engine.textcnn_features_fwd(xc, B, L, E, L * E, cw, cb)
        logits = new((B, 2), x.device)
        m = f32c(mask) if mask is not None else None
        ops.disc_head(feat, fc_w, fc_b, logits, mask=m)
        ctx.saved = (xc, feat, arg, m, fc_w, cw)
        ctx.dims = (B, L, E)
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        xc, feat, arg, m, fc_w, cw = ctx.saved
        B, L, E = ctx.dims
        dev = xc.device
        dl = f32c(dlogits)
        F_ = feat.shape[1]
        dW, db, dfeat = torch.empty_like(fc_w), new((2,), dev), new((B, F_), dev)
        ops.disc_head_bwd(feat, fc_w, dl, mask=m, dW=dW, db=db, dfeat=dfeat)
        Ks, Cos = [w.shape[2] for w in cw], [w.shape[0] for w in cw]
        dcw = [torch.empty_like(w) for w in cw]
        dcb = [new((w.shape[0],), dev) for w in cw]
        ops.conv_bwd_dw(Ks, Cos, 0, 0, dcw, dcb, dfeat, None, arg, xc, L * E, B, L, E)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = new((B, L, E), dev)
            ops.conv_bwd_dx(Ks, Cos, 0, 0, cw, dfeat, None, arg, B, L, E, dx, L * E)
        out = [dx, None, dW, db]
        for a, b in zip(dcw, dcb):
            out += [a, b]
        return tuple(out)


# ------------------------------------------------------------------------------------------------------------------
class _HighwayFn(torch.autograd.Function):
    """x [N,S]; weights: per layer (nw, nb, lw, lb, gw, gb)."""

    @staticmethod
    def forward(ctx, x, *ws):
        need_cuda(x, "Highway")
        layers = [dict(zip(("nw", "nb", "lw", "lb", "gw", "gb"), ws[i:i + 6])) for i in range(0, len(ws), 6)]
        y, saved = engine.highway_fwd(f32c(x), layers)
        ctx.layers, ctx.hw_saved = layers, saved
        return y

    @staticmethod
    def backward(ctx, dy):
        layers = ctx.layers
        grads = [{k: torch.empty_like(v) for k, v in l.items()} for l in layers]
        dx = engine.highway_bwd(ctx.hw_saved, layers, f32c(dy), grads, need_dx=ctx.needs_input_grad[0])
        flat = []
        for g in grads:
            flat += [g[k] for k in ("nw", "nb", "lw", "lb", "gw", "gb")]
        return (dx,) + tuple(flat)


class _TDNNFn(torch.autograd.Function):
    """x [N, W, Cc] already-embedded characters -> [N, S]."""

    @staticmethod
    def forward(ctx, x, *kernels):
        need_cuda(x, "TDNN")
        N, Wl, Cc = x.shape
        feat, saved = engine.tdnn_conv_fwd(f32c(x).view(N * Wl, Cc), N, Wl, list(kernels))
        ctx.tsaved, ctx.kernels = saved, kernels
        return feat

    @staticmethod
    def backward(ctx, dfeat):
        kernels = ctx.kernels
        dk = [torch.empty_like(k) for k in kernels]
        dxw = engine.tdnn_conv_bwd(ctx.tsaved, list(kernels), f32c(dfeat), dk, need_dx=ctx.needs_input_grad[0])
        s = ctx.tsaved
        return (dxw.view(s["N"], s["W"], s["Cc"]) if dxw is not None else None,) + tuple(dk)


class _EmbeddingFn(torch.autograd.Function):
    """Row gather with a table gradient (char embedding). idx any shape -> [..., E]."""

    @staticmethod
    def forward(ctx, idx, table):
        need_cuda(table, "Embedding")
        flat = idx.reshape(-1).contiguous()
        out = new((flat.numel(), table.shape[1]), table.device)
        ops.gather_rows(table, flat, out)
        ctx.save_for_backward(flat)
        ctx.tshape = table.shape
        return out.view(*idx.shape, table.shape[1])

    @staticmethod
    def backward(ctx, dout):
        (flat,) = ctx.saved_tensors
        dt = new(ctx.tshape, dout.device)
        ops.embedding_bwd(flat, f32c(dout.reshape(-1, ctx.tshape[1])), dt)
        return None, dt


def embedding_lookup(idx, table):
    """nn.Embedding forward. Frozen tables (requires_grad False) take the gather-only path."""
    if table.requires_grad and torch.is_grad_enabled():
        return _EmbeddingFn.apply(idx, table)
    need_cuda(table, "Embedding")
    flat = idx.reshape(-1).contiguous()
    out = new((flat.numel(), table.shape[1]), table.device)
    ops.gather_rows(table.detach(), flat, out)
    return out.view(*idx.shape, table.shape[1])
