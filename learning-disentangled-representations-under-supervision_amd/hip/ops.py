"""Tensor-level wrappers over the C-ABI (include/cgn_hip.h). torch is used only for device memory and streams.

Every function validates device/dtype/shape on the host before a kernel is launched (a faulting kernel can reset the
whole GPU node), launches on torch's current stream and returns torch tensors that alias caller-visible memory.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import Axis, GemmDesc

_workspaces = {}
WORKSPACE_BYTES = 96 << 20


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError("%s must be a float32 CUDA(HIP) tensor" % name)


def _chk_i64(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous int64 CUDA(HIP) tensor" % name)


def workspace(device):
    """Per-device split-K workspace (allocated once; the C library never allocates)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    ws = _workspaces.get(key)
    if ws is None:
        ws = torch.empty(WORKSPACE_BYTES // 4, dtype=torch.float32, device=torch.device("cuda", key))
        _workspaces[key] = ws
    return ws


def _span_ok(t, base_off, axes_extents):
    """Host-side bounds check: the largest element offset reached must lie inside the tensor's storage."""
    hi = base_off
    for ax, n in axes_extents:
        if n <= 0:
            return True
        i = n - 1
        if ax.rpg > 0:
            g = i // ax.rpg
            last = g * ax.gs + (i % ax.rpg) * ax.stride
            prev = (g - 1) * ax.gs + (ax.rpg - 1) * ax.stride if g > 0 else 0
            hi += max(last, prev)
        else:
            hi += i * ax.stride
    nelem = t.untyped_storage().nbytes() // t.element_size()
    return 0 <= hi < nelem


def gemm(a, b, out, *, M=None, N=None, K=None, alpha=1.0, beta=0.0, bias0=None, bias1=None,
         a_m=None, a_k=None, b_k=None, b_n=None, a_exp=None, softmax_bwd=None, force_tile=0, force_splitk=0):
    """out[M,N] = epilogue(alpha * A @ B + beta*out + bias0 + bias1).

    `a` is viewed as A[M,K] and `b` as B[K,N]; arbitrary 2-D strides (so `w.t()` is free). The optional cgn Axis
    overrides give two-level addressing (see include/cgn_hip.h) for overlapping windows / history buffers; when an
    override is given, M/K/N must be given explicitly and `a`/`b` only provide the base pointer.
    a_exp=(scale, lse): A := exp(A*scale - lse[m]).   softmax_bwd=dict(logits, lse, rowdot, target, inv_tau, ce_scale).
    """
    lib = _lib.load()
    _chk_f32(a, "a"), _chk_f32(b, "b"), _chk_f32(out, "out")
    if a_m is None or a_k is None:
        if a.dim() != 2:
            raise ValueError("a must be 2-D unless a_m/a_k are given")
        a_m = Axis(a.stride(0), 0, 0)
        a_k = Axis(a.stride(1), 0, 0)
        M = a.shape[0] if M is None else M
        K = a.shape[1] if K is None else K
    if b_k is None or b_n is None:
        if b.dim() != 2:
            raise ValueError("b must be 2-D unless b_k/b_n are given")
        b_k = Axis(b.stride(0), 0, 0)
        b_n = Axis(b.stride(1), 0, 0)
        if K is None:
            K = b.shape[0]
        elif b.shape[0] != K:
            raise ValueError("inner dimensions differ: %d vs %d" % (K, b.shape[0]))
        N = b.shape[1] if N is None else N
    if M is None or N is None or K is None:
        raise ValueError("M, N, K must be known")
    if out.dim() != 2 or out.shape[0] != M or out.shape[1] != N or out.stride(1) != 1:
        raise ValueError("out must be [M,N] with unit column stride, got %s" % (tuple(out.shape),))
    for ax in (a_m, a_k, b_k, b_n):
        if ax.stride < 0 or ax.gs < 0 or ax.rpg < 0:
            raise ValueError("negative strides are not supported")
    if not _span_ok(a, a.storage_offset(), [(a_m, M), (a_k, K)]):
        raise ValueError("A addressing runs outside its storage")
    if not _span_ok(b, b.storage_offset(), [(b_k, K), (b_n, N)]):
        raise ValueError("B addressing runs outside its storage")
    dev = out.device
    if a.device != dev or b.device != dev:
        raise ValueError("operands on different devices")
    d = GemmDesc()
    d.A, d.B, d.C = _ptr(a), _ptr(b), _ptr(out)
    for name, t in (("bias0", bias0), ("bias1", bias1)):
        if t is not None:
            _chk_f32(t, name)
            if t.numel() != N or not t.is_contiguous():
                raise ValueError("%s must be contiguous with N elements" % name)
        setattr(d, name, _ptr(t))
    d.a_m, d.a_k, d.b_k, d.b_n = a_m, a_k, b_k, b_n
    d.M, d.N, d.K = M, N, K
    d.ldc = out.stride(0)
    d.alpha, d.beta = alpha, beta
    d.a_transform = _lib.ATRANS_NONE
    keep = []
    if a_exp is not None:
        scale, lse = a_exp
        _chk_f32(lse, "lse")
        if lse.numel() != M or not lse.is_contiguous():
            raise ValueError("lse must have M elements")
        d.a_transform = _lib.ATRANS_EXP
        d.a_scale = scale
        d.a_row_param = _ptr(lse)
    d.epilogue = _lib.EPI_LINEAR
    if softmax_bwd is not None:
        lg, lse, rd, tg = (softmax_bwd[k] for k in ("logits", "lse", "rowdot", "target"))
        _chk_f32(lg, "logits"), _chk_f32(lse, "lse"), _chk_f32(rd, "rowdot"), _chk_i64(tg, "target")
        if lg.dim() != 2 or lg.shape[0] != M or lg.shape[1] != N or lg.stride(1) != 1:
            raise ValueError("softmax_bwd logits must be [M,N]")
        if lse.numel() != M or rd.numel() != M or tg.numel() != M or not (lse.is_contiguous() and rd.is_contiguous()):
            raise ValueError("softmax_bwd row vectors must have M elements")
        d.epilogue = _lib.EPI_SOFTMAX_BWD
        d.ep_logits, d.ep_lse, d.ep_rowdot, d.ep_target = _ptr(lg), _ptr(lse), _ptr(rd), _ptr(tg)
        d.ep_ld = lg.stride(0)
        d.ep_inv_tau = softmax_bwd["inv_tau"]
        d.ep_ce_scale = softmax_bwd["ce_scale"]
    ws = workspace(dev)
    d.workspace = _ptr(ws)
    d.workspace_bytes = ws.numel() * 4
    d.force_tile, d.force_splitk = force_tile, force_splitk
    with torch.cuda.device(dev):
        _lib.check(lib.cgn_gemm_f32(_stream(), C.byref(d)), "cgn_gemm_f32")
    del keep
    return out


def softmax_nll_fwd(logits, target, inv_tau):
    """Returns (lse[M], nll[M], loss[1]) for logits [M,V] (row stride free), model/cgn.py:1086-1092."""
    lib = _lib.load()
    _chk_f32(logits, "logits"), _chk_i64(target, "target")
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("logits must be [M,V] with unit column stride")
    M, V = logits.shape
    if target.numel() != M:
        raise ValueError("target must have M elements")
    lse = torch.empty(M, dtype=torch.float32, device=logits.device)
    nll = torch.empty(M, dtype=torch.float32, device=logits.device)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    with torch.cuda.device(logits.device):
        _lib.check(lib.cgn_softmax_nll_fwd_f32(_stream(), _ptr(logits), logits.stride(0), _ptr(target), M, V,
                                               inv_tau, _ptr(lse), _ptr(nll), _ptr(loss)), "cgn_softmax_nll_fwd_f32")
    return lse, nll, loss
