"""Tensor-level wrappers over the C-ABI (include/cgn_hip.h). torch is used only for device memory and streams.

Every function validates device/dtype/shape on the host before a kernel is launched (a faulting kernel can reset the
whole GPU node), launches on torch's current stream and returns torch tensors that alias caller-visible memory.
"""
import contextlib
import ctypes as C
import gc
import os

import torch

from . import _lib
from ._lib import Axis, GemmDesc

_workspaces = {}
WORKSPACE_BYTES = 96 << 20

# Operand type of every MFMA product (cgn_gemm_desc.compute): "f32" (default, the parity mode) or "bf16" (north-star's
# optional mode: operands rounded to bf16 inside the kernel, fp32 accumulation, fp32 everywhere else). Read when a
# product is launched, so a captured hipGraph keeps the mode it was captured in.
COMPUTE = _lib.COMPUTE_BF16 if os.environ.get("CGN_COMPUTE", "f32").lower() == "bf16" else _lib.COMPUTE_F32


# How an fp32 product is formed (cgn_gemm_desc.compute for the "f32" mode): True (default) = CGN_COMPUTE_F32X6, operands
# split exactly into three bf16 terms inside the kernel, six partial products on the bf16 matrix pipe, fp32 accumulation -
# fp32 accuracy at 2.67x less matrix-pipe time; False (CGN_GEMM_X6=0) = v_mfma_f32_32x32x2_f32. Read at launch time.
GEMM_X6 = os.environ.get("CGN_GEMM_X6", "1") != "0"


def _compute_code(compute):
    """None = the module-level mode | 'f32' (X6 or fp32 MFMA, by GEMM_X6) | 'f32_mfma' | 'f32x6' | 'bf16'."""
    if compute is None:
        compute = "bf16" if COMPUTE == _lib.COMPUTE_BF16 else "f32"
    if compute == "bf16":
        return _lib.COMPUTE_BF16
    if compute == "f32":
        return _lib.COMPUTE_F32X6 if GEMM_X6 else _lib.COMPUTE_F32
    if compute == "f32_mfma":
        return _lib.COMPUTE_F32
    if compute == "f32x6":
        return _lib.COMPUTE_F32X6
    raise ValueError("compute must be None, 'f32', 'f32_mfma', 'f32x6' or 'bf16'")


# bf16 mode only: hold the logits-sized tensors of the generator's output layer in bf16 (hip/step.py; CGN_STORE_* in
# include/cgn_hip.h). CGN_BF16_STORAGE=0 keeps every tensor fp32 (operands are then rounded inside the kernels only).
BF16_STORAGE = os.environ.get("CGN_BF16_STORAGE", "1") != "0"


# bf16 mode: the row-tiled recurrences at H > 512 take bf16 operands (cgn_lstm_set_operand_mode): CGN_LSTM_BF16_RT=2
# (default) plain bf16 operands with the hand-off loads pipelined, 1 two-term operands (~2^-16 per product), 0 keeps them
# on the fp32 MFMA (A/B switch)
LSTM_BF16_RT = int(os.environ.get("CGN_LSTM_BF16_RT", "2"))


def _sync_lstm_operand_mode():
    try:
        lib = _lib.load()
    except RuntimeError:
        return  # (no library: every op raises on its own)
    _lib.check(lib.cgn_lstm_set_operand_mode(int(LSTM_BF16_RT) if COMPUTE == _lib.COMPUTE_BF16 else 0),
               "cgn_lstm_set_operand_mode")


def set_compute(mode):
    """'f32' | 'bf16'; returns the previous mode."""
    global COMPUTE
    if mode not in ("f32", "bf16"):
        raise ValueError("compute mode must be 'f32' or 'bf16'")
    prev = "bf16" if COMPUTE == _lib.COMPUTE_BF16 else "f32"
    COMPUTE = _lib.COMPUTE_BF16 if mode == "bf16" else _lib.COMPUTE_F32
    _sync_lstm_operand_mode()
    return prev


def get_compute():
    return "bf16" if COMPUTE == _lib.COMPUTE_BF16 else "f32"


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _chk_f32(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise TypeError("%s must be a float32 CUDA(HIP) tensor" % name)


def _chk_f32_or_bf16(t, name):
    """bf16 STORAGE (the logits-sized tensors of the optional bf16 mode) where the entry point has a bf16 form."""
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype in (torch.float32, torch.bfloat16)):
        raise TypeError("%s must be a float32 or bfloat16 CUDA(HIP) tensor" % name)
    return t.dtype == torch.bfloat16


def _chk_i64(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int64 and t.is_contiguous()):
        raise TypeError("%s must be a contiguous int64 CUDA(HIP) tensor" % name)


def workspace(device, nbytes=None):
    """Split-K / column-sum workspace, one per (device, stream): kernels launched on different streams may overlap, so
    they must not share slabs. Allocated once per key (the C library never allocates). nbytes: a second, larger set for
    the bf16 x bf16 products (the fp32 planner's choices depend on its workspace's size and stay as they are)."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream, nbytes)
    ws = _workspaces.get(key)
    if ws is None:
        with torch.cuda.stream(torch.cuda.default_stream(idx)):  # owned by the default stream's pool, never freed
            ws = torch.empty((nbytes or WORKSPACE_BYTES) // 4, dtype=torch.float32, device=torch.device("cuda", idx))
        _workspaces[key] = ws
    return ws


# Step-scoped temporaries. While KEEP is a list every tensor handed out by alloc() is also appended to it, so nothing is
# returned to torch's caching allocator before the step ends: kernels of the same step running on different streams
# (and the nodes of a captured hipGraph) can then never see a buffer recycled under them.
KEEP = None


@contextlib.contextmanager
def capture_without_gc():
    """Stream capture must not be interrupted by Python's cyclic garbage collector: collecting an old hipGraph, event or
    stream object (e.g. a dropped engine's captured decode graphs) issues HIP calls that are illegal while this thread
    captures, and the runtime aborts the process. Collect first, then keep the collector off for the capture."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


def alloc(shape, dev, dtype=torch.float32):
    t = torch.empty(shape, dtype=dtype, device=dev)
    if KEEP is not None:
        KEEP.append(t)
    return t


def _span_ok(t, base_off, axes_extents):
    """Host-side bounds check: the largest element offset reached must lie inside the tensor's storage. Where the
    library may read an axis in 16-byte quads although its extent is not a multiple of 4 (csrc/gemm.hip axis_vec_ok:
    unit stride, plain axes, the other axis' stride covers the rounded-up extent) the extent counts rounded up: the
    last row's quad must lie inside the storage too."""
    hi = base_off
    quads = t.data_ptr() % 16 == 0  # the library never issues 16-byte loads from an unaligned base
    for j, (ax, n) in enumerate(axes_extents):
        if n <= 0:
            return True
        i = n - 1
        other = axes_extents[1 - j][0] if len(axes_extents) == 2 else None
        if (quads and n % 4 and ax.stride == 1 and ax.rpg == 0 and other is not None and other.rpg == 0
                and other.stride % 4 == 0 and other.stride >= pad4(n)):
            i = pad4(n) - 1
        if ax.rpg > 0:
            g = i // ax.rpg
            last = g * ax.gs + (i % ax.rpg) * ax.stride
            prev = (g - 1) * ax.gs + (ax.rpg - 1) * ax.stride if g > 0 else 0
            hi += max(last, prev)
        else:
            hi += i * ax.stride
    nelem = t.untyped_storage().nbytes() // t.element_size()
    return 0 <= hi < nelem


# Plan hint for products issued where the GPU is shared by several streams (hip/step.py sets it for the step's tail):
# {"splitk": n, "tile": t} stand in for force_splitk / force_tile when the caller gives none. None = the library's plan.
PLAN_HINT = None


def gemm(a, b, out, *, M=None, N=None, K=None, alpha=1.0, beta=0.0, bias0=None, bias1=None,
         a_m=None, a_k=None, b_k=None, b_n=None, a_exp=None, softmax_bwd=None, force_tile=0, force_splitk=0,
         dyn=None, compute=None, c_t=None, lse_stats=None, _desc_only=False):
    """out[M,N] = epilogue(alpha * A @ B + beta*out + bias0 + bias1).

    `a` is viewed as A[M,K] and `b` as B[K,N]; arbitrary 2-D strides (so `w.t()` is free). The optional cgn Axis
    overrides give two-level addressing (see include/cgn_hip.h) for overlapping windows / history buffers; when an
    override is given, M/K/N must be given explicitly and `a`/`b` only provide the base pointer.
    a_exp=(scale, lse): A := exp(A*scale - lse[m]).   softmax_bwd=dict(logits, lse, rowdot, target, inv_tau, ce_scale).
    compute: None (the module-level mode, see set_compute) | 'f32' | 'bf16' | 'f32_mfma' | 'f32x6' (_compute_code).
    lse_stats=inv_tau: ask for the fused log-softmax statistics of inv_tau * out (CGN_EPI_LINEAR_STATS: the vocabulary
    projection with the first pass of the CE behind it); the call then returns (out, stats) with stats a [M, chunks, 2]
    tensor for softmax_nll_from_stats(), or (out, None) where the library cannot serve this product that way.
    bf16 STORAGE (bf16 compute only, include/cgn_hip.h CGN_STORE_*): `a` may be a bfloat16 [M,K] matrix with unit k stride
    and rows 16 bytes aligned (K % 8 == 0); `out` and softmax_bwd's logits may be bfloat16; c_t: optional bfloat16 [N, >=M]
    receiving the transpose of a bfloat16 softmax_bwd result.
    """
    lib = _lib.load()
    a16, c16 = _chk_f32_or_bf16(a, "a"), _chk_f32_or_bf16(out, "out")
    _chk_f32(b, "b")
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
    if (BB_GEMM and not _desc_only and lse_stats is None and not force_tile and not force_splitk
            and _compute_code(compute) == _lib.COMPUTE_BF16):
        if _gemm_bb_route(a, b, out, M, N, K, a_m, a_k, b_k, b_n, alpha, beta, bias0, bias1, a_exp, softmax_bwd, c_t, dyn):
            return out
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
        ep16 = _chk_f32_or_bf16(lg, "logits")
        _chk_f32(lse, "lse"), _chk_f32(rd, "rowdot"), _chk_i64(tg, "target")
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
    if PLAN_HINT is not None and a_exp is None and softmax_bwd is None:
        force_tile = force_tile or PLAN_HINT.get("tile", 0)
        force_splitk = force_splitk or PLAN_HINT.get("splitk", 0)
    d.force_tile, d.force_splitk = force_tile, force_splitk
    d.dyn_scalars = _ptr(dyn)
    d.compute = _compute_code(compute)
    stats = None
    if lse_stats is not None and softmax_bwd is None and not (a16 or c16):
        d.ep_inv_tau = float(lse_stats)
        nch = C.c_int32()
        _lib.check(lib.cgn_gemm_stats_chunks(C.byref(d), C.byref(nch)), "cgn_gemm_stats_chunks")
        if nch.value > 0:
            stats = alloc((M, nch.value, 2), dev)
            d.epilogue = _lib.EPI_LINEAR_STATS
            d.ep_stats, d.ep_stats_ld = _ptr(stats), nch.value
    d.storage = ((_lib.STORE_A_BF16 if a16 else 0) | (_lib.STORE_C_BF16 if c16 else 0) |
                 (_lib.STORE_EP_BF16 if (softmax_bwd is not None and ep16) else 0))
    if d.storage and d.compute != _lib.COMPUTE_BF16:
        raise ValueError("bfloat16 tensors need the bf16 compute mode")
    if a16 and not (a_k.stride == 1 and a_k.rpg == 0 and a_m.rpg == 0 and a_m.stride % 8 == 0 and K % 8 == 0 and K > 0
                    and a.data_ptr() % 16 == 0):
        raise ValueError("a bfloat16 A must be [M,K] with unit k stride, K % 8 == 0 and 16-byte aligned rows")
    if c16 and beta != 0.0:
        raise ValueError("a bfloat16 out cannot be accumulated into (beta must be 0)")
    if c_t is not None:
        if not (c16 and softmax_bwd is not None and isinstance(c_t, torch.Tensor) and c_t.is_cuda
                and c_t.dtype == torch.bfloat16 and c_t.dim() == 2 and c_t.shape[0] == N and c_t.shape[1] == M
                and c_t.stride(1) == 1 and c_t.device == dev):
            raise ValueError("c_t must be a bfloat16 [N,M] matrix beside a bfloat16 softmax_bwd result")
        d.c_t, d.ld_ct = _ptr(c_t), c_t.stride(0)
    if _desc_only:
        return d
    prof = GEMM_PROFILE
    if prof is not None:
        t_, s_, ak, bk = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(lib.cgn_gemm_plan(C.byref(d), C.byref(t_), C.byref(s_), C.byref(ak), C.byref(bk)), "cgn_gemm_plan")
        key = (t_.value, ak.value, bk.value)
        if prof["variant"] is None or prof["variant"] == key:
            # events directly around the MFMA kernel (cgn_gemm_timed_f32), on the stream it is launched on
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.device(dev):
                e0.record(), e1.record()  # torch creates the hipEvent_t lazily, at the first record
                vec = C.c_int32()
                _lib.check(lib.cgn_gemm_timed_f32(_stream(), C.byref(d), e0.cuda_event, e1.cuda_event, C.byref(vec)),
                           "cgn_gemm_timed_f32")
            # (tile, A k-contiguous, B k-contiguous, vector mode, exp A-loader, softmax-bwd epilogue, bf16 operands)
            sym = key + (vec.value, d.a_transform == _lib.ATRANS_EXP, softmax_bwd is not None,
                         d.compute == _lib.COMPUTE_BF16)
            prof["records"].append((sym, s_.value, M, N, K, e0, e1))
            return out if lse_stats is None else (out, stats)
    with torch.cuda.device(dev):
        _lib.check(lib.cgn_gemm_f32(_stream(), C.byref(d)), "cgn_gemm_f32")
    return out if lse_stats is None else (out, stats)


# bf16 mode: products of plain / history-buffer operands with M, N >= 128 run the bf16 x bf16 kernel (csrc/bb_gemm.hip)
# on operands rounded ONCE by a cast kernel (transposing where the operand is not k-contiguous); CGN_BB_GEMM=0 keeps
# them on the kernels that round fp32 operands at LDS staging (A/B switch). Below BB_MIN_FLOP the casts do not pay.
BB_GEMM = os.environ.get("CGN_BB_GEMM", "1") != "0"
BB_WORKSPACE_BYTES = 384 << 20  # split-K slabs of the tall-K products at configs[3] shapes (dH1: 3 x 43 MB)
BB_MIN_FLOP = float(os.environ.get("CGN_BB_MIN_FLOP", "2e10"))


def pad8(n):
    return (n + 7) // 8 * 8


def pad16(n):
    return (n + 15) // 16 * 16


# While a bb_memo() block is open, ops.gemm's bf16 copies of fp32 operands are kept and served again to later products
# of the block that name the same operand (same base, axes, extents). The caller promises that the operands are not
# written inside the block.
_BB_MEMO = None


@contextlib.contextmanager
def bb_memo():
    global _BB_MEMO
    prev, _BB_MEMO = _BB_MEMO, ({} if _BB_MEMO is None else _BB_MEMO)
    try:
        yield
    finally:
        _BB_MEMO = prev


def cast_bf16(src, rows, n_rows, n_cols, transpose=False):
    """bf16 copy of the fp32 matrix whose n_rows rows sit at the two-level Axis `rows` of `src` (unit column stride):
    [n_rows, pad8(n_cols)], or transposed [n_cols, pad8(n_rows)]; padding columns are zeros (cgn_cast_bf16_f32)."""
    _chk_f32(src, "src")
    shape = (n_cols, pad8(n_rows)) if transpose else (n_rows, pad8(n_cols))
    dst = alloc(shape, src.device, torch.bfloat16)
    _call("cgn_cast_bf16_f32", src.device, _ptr(src), rows, n_rows, n_cols, _ptr(dst), shape[1], 1 if transpose else 0)
    return dst


def gemm_bb(a16, b16, out, *, K=None, alpha=1.0, beta=0.0, bias0=None, bias1=None, force_tile=0, force_splitk=0,
            a_exp=None, softmax_bwd=None, c_t=None, dyn=None):
    """out[M,N] = alpha * a16 @ b16^T + beta*out + bias0 + bias1 with bfloat16 a16 [M, >=K], b16 [N, >=K] (unit k
    stride, row strides % 8 == 0, K % 8 == 0) and a float32 or bfloat16 out (cgn_gemm_bb). force_tile: 256 | 192 (the
    256x128 tile) | 128 | 384 (128x384: float32 out, N <= 384). a_exp=(scale, lse): a16 := exp(a16*scale - lse[m]); softmax_bwd=dict(logits (bfloat16), lse,
    rowdot, target, inv_tau, ce_scale) with a bfloat16 out and the optional transposed copy c_t [N, >=M], as gemm()."""
    lib = _lib.load()
    for t, name in ((a16, "a16"), (b16, "b16")):
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 2
                and t.stride(1) == 1):
            raise TypeError("%s must be a 2-D bfloat16 CUDA(HIP) tensor with unit column stride" % name)
    c16 = _chk_f32_or_bf16(out, "out")
    K = a16.shape[1] if K is None else K
    M, N = a16.shape[0], b16.shape[0]
    if K > a16.shape[1] or K > b16.shape[1] or K % 8:
        raise ValueError("K must be a multiple of 8 inside both operands' rows")
    if out.dim() != 2 or out.shape[0] != M or out.shape[1] != N or out.stride(1) != 1:
        raise ValueError("out must be [M,N] with unit column stride")
    dev = out.device
    d = _lib.GemmBBDesc()
    d.A, d.B, d.C = _ptr(a16), _ptr(b16), _ptr(out)
    for name, t in (("bias0", bias0), ("bias1", bias1)):
        if t is not None:
            _chk_f32(t, name)
            if t.numel() != N or not t.is_contiguous():
                raise ValueError("%s must be contiguous with N elements" % name)
        setattr(d, name, _ptr(t))
    ws = workspace(dev, BB_WORKSPACE_BYTES)
    d.workspace, d.workspace_bytes = _ptr(ws), ws.numel() * 4
    d.lda, d.ldb, d.ldc = a16.stride(0), b16.stride(0), out.stride(0)
    d.M, d.N, d.K = M, N, K
    d.alpha, d.beta = alpha, beta
    d.c_bf16 = 1 if c16 else 0
    d.force_tile, d.force_splitk = force_tile, force_splitk
    d.a_transform, d.epilogue = _lib.ATRANS_NONE, _lib.EPI_LINEAR
    if a_exp is not None:
        scale, lse = a_exp
        _chk_f32(lse, "lse")
        if lse.numel() != M or not lse.is_contiguous():
            raise ValueError("lse must have M elements")
        d.a_transform, d.a_scale, d.a_row_param = _lib.ATRANS_EXP, scale, _ptr(lse)
    if softmax_bwd is not None:
        lg, lse, rd, tg = (softmax_bwd[k] for k in ("logits", "lse", "rowdot", "target"))
        _chk_f32(lse, "lse"), _chk_f32(rd, "rowdot"), _chk_i64(tg, "target")
        if not (c16 and lg.dtype == torch.bfloat16 and lg.is_cuda and lg.dim() == 2 and lg.shape[0] == M
                and lg.shape[1] == N and lg.stride(1) == 1):
            raise ValueError("softmax_bwd here takes bfloat16 logits [M,N] and a bfloat16 out")
        if lse.numel() != M or rd.numel() != M or tg.numel() != M or not (lse.is_contiguous() and rd.is_contiguous()):
            raise ValueError("softmax_bwd row vectors must have M elements")
        d.epilogue = _lib.EPI_SOFTMAX_BWD
        d.ep_logits, d.ep_lse, d.ep_rowdot, d.ep_target = _ptr(lg), _ptr(lse), _ptr(rd), _ptr(tg)
        d.ep_ld = lg.stride(0)
        d.ep_inv_tau, d.ep_ce_scale = softmax_bwd["inv_tau"], softmax_bwd["ce_scale"]
    if c_t is not None:
        if not (softmax_bwd is not None and isinstance(c_t, torch.Tensor) and c_t.is_cuda and c_t.dtype == torch.bfloat16
                and c_t.dim() == 2 and c_t.shape[0] == N and c_t.shape[1] == M and c_t.stride(1) == 1
                and c_t.device == dev):
            raise ValueError("c_t must be a bfloat16 [N,M] matrix beside a bfloat16 softmax_bwd result")
        d.c_t, d.ld_ct = _ptr(c_t), c_t.stride(0)
    d.dyn_scalars = _ptr(dyn)
    prof = GEMM_PROFILE
    tile, sk = C.c_int32(), C.c_int32()
    with torch.cuda.device(dev):
        if prof is not None and prof["variant"] in (None, ("bb",)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(), e1.record()
            _lib.check(lib.cgn_gemm_bb(_stream(), C.byref(d), e0.cuda_event, e1.cuda_event, C.byref(tile), C.byref(sk)),
                       "cgn_gemm_bb")
            prof["records"].append((("bb", tile.value, c16, a_exp is not None, softmax_bwd is not None), sk.value,
                                    M, N, K, e0, e1))
        else:
            _lib.check(lib.cgn_gemm_bb(_stream(), C.byref(d), None, None, C.byref(tile), C.byref(sk)), "cgn_gemm_bb")
    return out


def _bb_operand(t, row_ax, k_ax, rows, K, is16):
    """The bf16 [rows, pad8(K)] k-contiguous form of a GEMM operand given by its row axis and K axis, or None when the
    addressing is not one the cast kernels take."""
    if is16:
        ok = (k_ax.stride == 1 and k_ax.rpg == 0 and row_ax.rpg == 0 and t.dim() == 2 and t.shape[0] == rows
              and t.stride(0) == row_ax.stride and t.shape[1] >= K)
        return t if ok else None
    kc = k_ax.stride == 1 and k_ax.rpg == 0  # k-contiguous: rows along the row axis
    if not kc and not (row_ax.stride == 1 and row_ax.rpg == 0):
        return None
    ax = row_ax if kc else k_ax  # (row-contiguous: the source's rows are the K axis, cast with a transpose)
    key = (t.data_ptr(), ax.stride, ax.rpg, ax.gs, rows, K, kc)
    if _BB_MEMO is not None and key in _BB_MEMO:
        return _BB_MEMO[key]
    r = cast_bf16(t, ax, rows, K) if kc else cast_bf16(t, ax, K, rows, transpose=True)
    if _BB_MEMO is not None:
        _BB_MEMO[key] = r
    return r


def _gemm_bb_route(a, b, out, M, N, K, a_m, a_k, b_k, b_n, alpha, beta, bias0, bias1, a_exp=None, softmax_bwd=None,
                   c_t=None, dyn=None):
    lib = _lib.load()
    # (the two logits-sized products cast only their small operand: they pay from a fifth of the size on)
    floor = BB_MIN_FLOP * (0.2 if (a_exp is not None or softmax_bwd is not None) else 1.0)
    if not lib.cgn_gemm_bb_wants(M, N, K) or 2.0 * M * N * K < floor:
        return False
    a16, c16 = a.dtype == torch.bfloat16, out.dtype == torch.bfloat16
    if a_exp is not None and not a16:  # (the exp loader reads bf16 logits: the bf16 storage of the mode)
        return False
    if softmax_bwd is not None and not (c16 and softmax_bwd["logits"].dtype == torch.bfloat16 and a_exp is None):
        return False
    if c_t is not None and softmax_bwd is None:
        return False
    if c16 and (beta != 0.0 or out.stride(0) % 8 or out.data_ptr() % 16):
        return False
    # (a cast reads 4-byte elements through a 64-bit offset: any span; the product addresses each operand in 31 bits)
    if max(M, N) * pad8(K) * 2 >= (1 << 31) - 64:
        return False
    if a16 and not (a_k.stride == 1 and a_k.rpg == 0 and a_m.rpg == 0 and a.dim() == 2 and K % 8 == 0
                    and a_m.stride % 8 == 0 and a.data_ptr() % 16 == 0 and M * a_m.stride * 2 < (1 << 31) - 64):
        return False
    ok = lambda row_ax, k_ax: ((k_ax.stride == 1 and k_ax.rpg == 0) or (row_ax.stride == 1 and row_ax.rpg == 0))
    if (not a16 and not ok(a_m, a_k)) or not ok(b_n, b_k):
        return False
    A = _bb_operand(a, a_m, a_k, M, K, a16)
    if A is None:
        return False
    B = _bb_operand(b, b_n, b_k, N, K, False)
    Kp = pad8(K)
    gemm_bb(A if a16 else A[:M], B[:N], out, K=Kp if not a16 else K, alpha=alpha, beta=beta, bias0=bias0, bias1=bias1,
            a_exp=a_exp, softmax_bwd=softmax_bwd, c_t=c_t, dyn=dyn)
    return True


def gemm_group(calls):
    """Several independent small products in ONE launch (cgn_gemm_group_f32): calls = [(a, b, out, kwargs), ...] with
    the arguments of gemm(); 64x64 tiles, no split-K, linear epilogue, equal operand orientations (host-checked by the
    library). Used for the per-width convolution products of the char-TDNN and the text-CNN."""
    n = len(calls)
    if n == 0:
        return
    if n > 8:
        gemm_group(calls[:8])
        gemm_group(calls[8:])
        return
    lib = _lib.load()
    arr = (GemmDesc * n)()
    dev = calls[0][2].device
    for i, (a, b, out, kw) in enumerate(calls):
        if out.device != dev:
            raise ValueError("grouped products must live on one device")
        arr[i] = gemm(a, b, out, _desc_only=True, **kw)
    if GEMM_PROFILE is not None:  # bench.py's step-level MFMA accounting (FLOPs only; grouped launches are not timed)
        GEMM_PROFILE["group_flops"] = GEMM_PROFILE.get("group_flops", 0.0) + sum(
            2.0 * arr[i].M * arr[i].N * arr[i].K for i in range(n))
    with torch.cuda.device(dev):
        _lib.check(lib.cgn_gemm_group_f32(_stream(), arr, n), "cgn_gemm_group_f32")


def act_maxpool_group(convs, B, Ps, act, feat, col0s, arg):
    """act_maxpool for several widths in one launch: convs[i] [B*Ps[i], Co_i] contiguous -> feat/arg columns
    col0s[i] .. col0s[i]+Co_i."""
    n = len(convs)
    _req(1 <= n <= 8 and len(Ps) == n and len(col0s) == n, "1..8 widths")
    ldf = _rows2d(feat, "feat")
    _req(arg.is_cuda and arg.dtype == torch.int32 and arg.is_contiguous() and arg.shape == feat.shape, "arg must be int32")
    ptrs = (C.c_void_p * n)()
    P_, Co_, c0_ = (C.c_int32 * n)(), (C.c_int32 * n)(), (C.c_int32 * n)()
    for i, cv in enumerate(convs):
        _chk_f32(cv, "conv")
        _req(cv.is_contiguous() and cv.dim() == 2 and cv.shape[0] == B * Ps[i] and feat.shape[0] == B
             and feat.shape[1] >= col0s[i] + cv.shape[1], "act_maxpool_group shapes")
        ptrs[i], P_[i], Co_[i], c0_[i] = cv.data_ptr(), Ps[i], cv.shape[1], col0s[i]
    _call("cgn_act_maxpool_group_f32", feat.device, n, ptrs, P_, Co_, c0_, B, act, _ptr(feat), ldf, _ptr(arg),
          arg.stride(0))


# bench.py sets this to {"variant": (tile, a_kcontig, b_kcontig) or None, "records": []} to time GEMM launches with
# HIP events on the launch stream; None (default) = no instrumentation.
GEMM_PROFILE = None


def softmax_nll_fwd(logits, target, inv_tau, dyn=None):
    """Returns (lse[M], nll[M], loss[1]) for logits [M,V] (row stride free), model/cgn.py:1086-1092.
    dyn: optional device float tensor whose element 0 overrides inv_tau at run time (graph replay)."""
    lib = _lib.load()
    in16 = _chk_f32_or_bf16(logits, "logits")
    _chk_i64(target, "target")
    if logits.dim() != 2 or logits.stride(1) != 1:
        raise ValueError("logits must be [M,V] with unit column stride")
    M, V = logits.shape
    if target.numel() != M:
        raise ValueError("target must have M elements")
    lse, nll, loss = alloc((M,), logits.device), alloc((M,), logits.device), alloc((1,), logits.device)
    name = "cgn_softmax_nll_fwd_bf16" if in16 else "cgn_softmax_nll_fwd_f32"
    with torch.cuda.device(logits.device):
        _lib.check(getattr(lib, name)(_stream(), _ptr(logits), logits.stride(0), _ptr(target), M, V,
                                      inv_tau, _ptr(dyn), _ptr(lse), _ptr(nll), _ptr(loss)), name)
    return lse, nll, loss


# The vocabulary projection leaves the softmax statistics of its own output (gemm(lse_stats=...)): CGN_FUSED_LSE=0 keeps
# the separate one-read softmax / NLL kernel (A/B switch; the bf16 mode and the fp32-MFMA products always use it)
FUSED_LSE = os.environ.get("CGN_FUSED_LSE", "1") != "0"


def softmax_nll_from_stats(stats, logits, target, inv_tau, dyn=None):
    """(lse[M], nll[M], loss[1]) as softmax_nll_fwd defines them, from the statistics a gemm(lse_stats=inv_tau) product
    left (cgn_softmax_nll_from_stats_f32): the logits are only read at the M target positions."""
    _chk_f32(stats, "stats"), _chk_f32(logits, "logits"), _chk_i64(target, "target")
    M, V = logits.shape
    _req(stats.is_contiguous() and stats.dim() == 3 and stats.shape[0] == M and stats.shape[2] == 2
         and stats.shape[1] == (V + 63) // 64 and logits.stride(1) == 1 and target.numel() == M
         and (M + 3) // 4 + 1 <= FUSED_SCRATCH_WORDS, "softmax_nll_from_stats shapes")
    lse, nll, loss = alloc((M,), logits.device), alloc((M,), logits.device), alloc((1,), logits.device)
    ws = _fused_scratch_ws(logits.device)
    _call("cgn_softmax_nll_from_stats_f32", logits.device, _ptr(stats), stats.shape[1], stats.shape[1], _ptr(logits),
          logits.stride(0), _ptr(target), M, V, float(inv_tau), _ptr(dyn), _ptr(lse), _ptr(nll), _ptr(loss), _ptr(ws),
          ws.numel() * 4)
    return lse, nll, loss


def log_softmax_rows(logits, lse, rows):
    """out[i] = logits[rows[i]] - lse[rows[i]] for logits [R,V] (contiguous), lse [R], rows int64 [n] (device): the
    log_softmax rows of model/cgn.py:803 picked after the fact (sampler learning=True)."""
    _chk_f32(logits, "logits"), _chk_f32(lse, "lse"), _chk_i64(rows, "rows")
    if logits.dim() != 2 or not logits.is_contiguous() or not lse.is_contiguous() or lse.numel() != logits.shape[0]:
        raise ValueError("logits must be contiguous [R,V] and lse [R]")
    if rows.dim() != 1 or not rows.is_contiguous():
        raise ValueError("rows must be a contiguous int64 vector")
    R, V = logits.shape
    out = alloc((rows.numel(), V), logits.device)
    with torch.cuda.device(logits.device):
        _lib.check(_lib.load().cgn_log_softmax_rows_f32(_stream(), _ptr(logits), V, _ptr(lse), _ptr(rows), rows.numel(),
                                                        R, V, _ptr(out), V), "cgn_log_softmax_rows_f32")
    return out


# ------------------------------------------------------------------------------------------------------------------
# thin wrappers over the remaining entry points; each checks shapes on the host before launching
# ------------------------------------------------------------------------------------------------------------------
def _call(name, dev, *args):
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(getattr(lib, name)(_stream(), *args), name)


def _req(cond, msg):
    if not cond:
        raise ValueError(msg)


def _rows2d(t, name):
    """[R, C] view with unit column stride; returns (tensor, ld)."""
    _chk_f32(t, name)
    _req(t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), "%s must be 2-D with unit column stride" % name)
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def fill(t, v):
    _chk_f32(t, "t")
    _req(t.is_contiguous(), "fill needs a contiguous tensor")
    _call("cgn_fill_f32", t.device, _ptr(t), t.numel(), float(v))
    return t


def copy2d(src, dst, scale=1.0, beta=0.0):
    lds, ldd = _rows2d(src, "src"), _rows2d(dst, "dst")
    _req(src.shape == dst.shape, "copy2d shape mismatch")
    _call("cgn_copy2d_f32", dst.device, _ptr(src), lds, _ptr(dst), ldd, src.shape[0], src.shape[1], scale, beta)
    return dst


def transpose(src, dst):
    _chk_f32(src, "src"), _chk_f32(dst, "dst")
    _req(src.dim() == 2 and src.is_contiguous() and dst.is_contiguous() and dst.shape == (src.shape[1], src.shape[0]),
         "transpose needs contiguous [R,C] -> [C,R]")
    _call("cgn_transpose_f32", dst.device, _ptr(src), src.shape[0], src.shape[1], _ptr(dst))
    return dst


def gather_rows(table, idx, out, mask=None):
    """out[i,:] = table[idx[i],:] (* mask[i,:]). out is [n, E] (row stride free)."""
    _chk_f32(table, "table"), _chk_i64(idx, "idx")
    ldo = _rows2d(out, "out")
    V, E = table.shape
    n = idx.numel()
    _req(table.is_contiguous() and out.shape == (n, E), "gather_rows shapes")
    if mask is not None:
        _chk_f32(mask, "mask")
        _req(mask.is_contiguous() and mask.numel() == n * E, "mask must be [n,E]")
    _call("cgn_gather_rows_f32", out.device, _ptr(table), V, E, _ptr(idx), n, _ptr(mask), _ptr(out), ldo)
    return out


def pad4(n):
    return (n + 3) // 4 * 4


def new2d(rows, cols, dev):
    """[rows, cols] fp32 view whose row stride is a multiple of 4 floats (16-byte aligned rows for float4 GEMM loads)."""
    return alloc((rows, pad4(cols)), dev)[:, :cols]


def highway_gate_fwd(pre, S, out):
    """pre [N, 3*Sp] contiguous with Sp = pre.shape[1] // 3 >= S; out [N, S]."""
    _chk_f32(pre, "pre")
    N, Sp3 = pre.shape
    Sp = Sp3 // 3
    ldo = _rows2d(out, "out")
    _req(pre.is_contiguous() and Sp3 == 3 * Sp and Sp >= S and out.shape == (N, S), "highway_gate_fwd shapes")
    _call("cgn_highway_gate_fwd_f32", out.device, _ptr(pre), N, S, Sp, _ptr(out), ldo)
    return out


def highway_gate_bwd(pre, S, dy, dpre):
    _chk_f32(pre, "pre"), _chk_f32(dpre, "dpre")
    N, Sp3 = pre.shape
    Sp = Sp3 // 3
    ldy = _rows2d(dy, "dy")
    _req(pre.is_contiguous() and dpre.is_contiguous() and dpre.shape == pre.shape and Sp >= S and dy.shape == (N, S),
         "highway_gate_bwd shapes")
    _call("cgn_highway_gate_bwd_f32", dpre.device, _ptr(pre), N, S, Sp, _ptr(dy), ldy, _ptr(dpre))
    return dpre


def reparam_kld_fwd(mulv, eps, c, zc, stdv, kld):
    B, Z2 = mulv.shape
    Z = Z2 // 2
    Cc = zc.shape[1] - Z
    for t, n in ((mulv, "mulv"), (eps, "eps"), (zc, "zc"), (stdv, "stdv"), (kld, "kld")):
        _chk_f32(t, n)
        _req(t.is_contiguous(), n + " must be contiguous")
    _req(eps.shape == (B, Z) and stdv.shape == (B, Z) and zc.shape[0] == B and Cc >= 0 and kld.numel() == 1,
         "reparam_kld_fwd shapes")
    if Cc > 0:
        _chk_f32(c, "c")
        _req(c.is_contiguous() and c.numel() == B * Cc, "c must be [B,C]")
    _call("cgn_reparam_kld_fwd_f32", zc.device, _ptr(mulv), _ptr(eps), _ptr(c if Cc > 0 else None), B, Z, Cc, _ptr(zc),
          _ptr(stdv), _ptr(kld))


def reparam_kld_bwd(mulv, eps, g, kld_weight, dmulv, dyn=None):
    B, Z2 = mulv.shape
    Z = Z2 // 2
    for t, n in ((mulv, "mulv"), (eps, "eps"), (dmulv, "dmulv")):
        _chk_f32(t, n)
        _req(t.is_contiguous(), n + " must be contiguous")
    _req(eps.shape == (B, Z) and dmulv.shape == mulv.shape, "reparam_kld_bwd shapes")
    ldg = 0
    if g is not None:
        ldg = _rows2d(g, "g")
        _req(g.shape[0] == B and g.shape[1] >= Z, "g must be [B, >=Z]")
    _call("cgn_reparam_kld_bwd_f32", dmulv.device, _ptr(mulv), _ptr(eps), _ptr(g), ldg, kld_weight, _ptr(dyn), B, Z,
          _ptr(dmulv))
    return dmulv


def colsum(x, out, beta=0.0):
    if isinstance(x, torch.Tensor) and x.dtype == torch.bfloat16:  # the bf16 mode's dlogits (bf16 storage)
        _req(x.is_cuda and x.dim() == 2 and x.stride(1) == 1, "colsum x must be [R,C] with unit column stride")
        _chk_f32(out, "out")
        R, Cn = x.shape
        _req(out.is_contiguous() and out.numel() == Cn, "colsum out must have C elements")
        ws = workspace(x.device)
        _call("cgn_colsum_bf16", x.device, _ptr(x), x.stride(0), R, Cn, beta, _ptr(out), _ptr(ws), ws.numel() * 4)
        return out
    ld = _rows2d(x, "x")
    _chk_f32(out, "out")
    R, Cn = x.shape
    _req(out.is_contiguous() and out.numel() == Cn, "colsum out must have C elements")
    ws = workspace(x.device)
    _call("cgn_colsum_f32", x.device, _ptr(x), ld, R, Cn, beta, _ptr(out), _ptr(ws), ws.numel() * 4)
    return out


def sum_time(x, out):
    _chk_f32(x, "x"), _chk_f32(out, "out")
    B, T, Cn = x.shape
    _req(x.is_contiguous() and out.is_contiguous() and out.shape == (B, Cn), "sum_time shapes")
    _call("cgn_sum_time_f32", x.device, _ptr(x), B, T, Cn, _ptr(out))
    return out


def rowdot(a, b, out):
    lda, ldb = _rows2d(a, "a"), _rows2d(b, "b")
    _chk_f32(out, "out")
    _req(a.shape == b.shape and out.is_contiguous() and out.numel() == a.shape[0], "rowdot shapes")
    _call("cgn_rowdot_f32", a.device, _ptr(a), lda, _ptr(b), ldb, a.shape[0], a.shape[1], _ptr(out))
    return out


# ------------------------------------------------------------------------------------------------------------------
# fused launch-bound chains (csrc/fused.hip)
# ------------------------------------------------------------------------------------------------------------------
FUSED_CHAINS = os.environ.get("CGN_FUSED_CHAINS", "1") != "0"  # 0: the round-2 chains of small kernels (A/B, fallback test)
_fused_scratch = {}
FUSED_SCRATCH_WORDS = 4096 + 64


def _fused_scratch_ws(device):
    """Ticket + per-sentence slots of the fused kernels' batch sums: zeroed once, one per (device, stream)."""
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _fused_scratch.get(key)
    if ws is None:
        with torch.cuda.stream(torch.cuda.default_stream(idx)):
            ws = torch.zeros(FUSED_SCRATCH_WORDS, dtype=torch.int32, device=torch.device("cuda", idx))
        _fused_scratch[key] = ws
    return ws


def code_head_eligible(B, He, Z, Cc, w2, wzc):
    """Shapes / layouts the fused code head takes (cgn_code_head_fwd/bwd_f32); else the unfused chain runs."""
    return (FUSED_CHAINS and w2 is not None and 2 * He <= 2048 and He % 2 == 0 and Z <= 128 and Z + Cc <= 140
            and B + 1 <= FUSED_SCRATCH_WORDS and w2.stride(1) == 1 and w2.stride(0) % 4 == 0 and w2.data_ptr() % 16 == 0
            and (wzc is None or (wzc.stride(1) == 1 and wzc.stride(0) % 4 == 0 and wzc.data_ptr() % 16 == 0
                                 and wzc.stride(0) >= pad4(Z + Cc) and wzc.shape[0] <= 4096)))


def code_head_fwd(cf, cr, w2, b2, eps, c, wzc, ctx, mulv, zc, stdv, kld, zg):
    """cf, cr: [B, He] final cell states (equal row strides); w2 [2Z, 2He] stacked mu|logvar weights; wzc: generator
    W_ih_l0[:, E:] view [4Hg, Z+C] or None. Fills ctx, mulv, zc, stdv, kld (and zg). include/cgn_hip.h."""
    B, He = cf.shape
    Z = eps.shape[1]
    Cc = zc.shape[1] - Z
    for t, n in ((cf, "cf"), (cr, "cr"), (w2, "w2"), (b2, "b2"), (eps, "eps"), (ctx, "ctx"), (mulv, "mulv"), (zc, "zc"),
                 (stdv, "stdv"), (kld, "kld")):
        _chk_f32(t, n)
    _req(cr.shape == cf.shape and cf.stride(1) == 1 and cr.stride(1) == 1 and cf.stride(0) == cr.stride(0),
         "cf / cr must be [B, He] views with equal row strides")
    _req(tuple(w2.shape) == (2 * Z, 2 * He) and b2.is_contiguous() and b2.numel() == 2 * Z, "w2 / b2 shapes")
    _req(eps.is_contiguous() and eps.shape == (B, Z) and ctx.is_contiguous() and ctx.shape == (B, 2 * He)
         and mulv.is_contiguous() and mulv.shape == (B, 2 * Z) and zc.is_contiguous() and zc.shape[0] == B
         and stdv.is_contiguous() and stdv.shape == (B, Z) and kld.numel() == 1, "code_head_fwd shapes")
    _req(code_head_eligible(B, He, Z, Cc, w2, wzc), "shape not eligible for the fused code head")
    d = _lib.CodeHeadFwd()
    d.cf, d.cr, d.ldc = cf.data_ptr(), cr.data_ptr(), cf.stride(0)
    d.W2, d.ldw2, d.b2, d.eps = w2.data_ptr(), w2.stride(0), b2.data_ptr(), eps.data_ptr()
    if Cc > 0:
        _chk_f32(c, "c")
        _req(c.is_contiguous() and c.numel() == B * Cc, "c must be [B, C]")
        d.c = c.data_ptr()
    if wzc is not None:
        _chk_f32(wzc, "wzc"), _chk_f32(zg, "zg")
        G4 = wzc.shape[0]
        _req(wzc.shape[1] == Z + Cc and zg.is_contiguous() and zg.shape == (B, G4), "wzc / zg shapes")
        # the kernel reads whole 16-byte quads of a weight row: the last row's pad columns must lie inside the storage
        nelem = wzc.untyped_storage().nbytes() // 4 - wzc.storage_offset()
        _req((G4 - 1) * wzc.stride(0) + pad4(Z + Cc) <= nelem, "wzc rows must be readable up to a multiple of 4 columns")
        d.Wzc, d.ldzc, d.zg, d.G4 = wzc.data_ptr(), wzc.stride(0), zg.data_ptr(), G4
    d.ctx, d.mulv, d.zc, d.stdv, d.kld = ctx.data_ptr(), mulv.data_ptr(), zc.data_ptr(), stdv.data_ptr(), kld.data_ptr()
    ws = _fused_scratch_ws(cf.device)
    d.scratch, d.scratch_bytes = ws.data_ptr(), ws.numel() * 4
    d.B, d.He, d.Z, d.C = B, He, Z, Cc
    _call("cgn_code_head_fwd_f32", cf.device, C.byref(d))


def code_head_bwd(dg0, wzc, mulv, eps, w2, kld_weight, dsum, dzc, dmulv, dcf, dcr, dyn=None):
    """Backward of code_head_fwd from the generator's layer-0 gate gradients dg0 [B, T, 4Hg]. include/cgn_hip.h."""
    B, T, G4 = dg0.shape
    Z = eps.shape[1]
    Cc = dzc.shape[1] - Z
    He = dcf.shape[1]
    for t, n in ((dg0, "dg0"), (wzc, "wzc"), (mulv, "mulv"), (eps, "eps"), (w2, "w2"), (dsum, "dsum"), (dzc, "dzc"),
                 (dmulv, "dmulv"), (dcf, "dcf"), (dcr, "dcr")):
        _chk_f32(t, n)
    for t, n in ((dg0, "dg0"), (mulv, "mulv"), (eps, "eps"), (dsum, "dsum"), (dzc, "dzc"), (dmulv, "dmulv"), (dcf, "dcf"),
                 (dcr, "dcr")):
        _req(t.is_contiguous(), n + " must be contiguous")
    _req(tuple(wzc.shape) == (G4, Z + Cc) and wzc.stride(1) == 1 and tuple(w2.shape) == (2 * Z, 2 * He)
         and w2.stride(1) == 1 and mulv.shape == (B, 2 * Z) and eps.shape == (B, Z) and dsum.shape == (B, G4)
         and dzc.shape[0] == B and dmulv.shape == (B, 2 * Z) and dcf.shape == (B, He) and dcr.shape == (B, He),
         "code_head_bwd shapes")
    _req(G4 <= 4096 and Z <= 128 and Z + Cc <= 144, "shape not eligible for the fused code head backward")
    d = _lib.CodeHeadBwd()
    d.dg0, d.Wzc, d.ldzc, d.mulv, d.eps = dg0.data_ptr(), wzc.data_ptr(), wzc.stride(0), mulv.data_ptr(), eps.data_ptr()
    d.dyn_kld_w = dyn.data_ptr() if dyn is not None else None
    d.W2, d.ldw2 = w2.data_ptr(), w2.stride(0)
    d.dsum, d.dzc, d.dmulv, d.dcf, d.dcr = dsum.data_ptr(), dzc.data_ptr(), dmulv.data_ptr(), dcf.data_ptr(), dcr.data_ptr()
    d.kld_w = float(kld_weight)
    d.B, d.T, d.He, d.Z, d.C, d.G4 = B, T, He, Z, Cc, G4
    _call("cgn_code_head_bwd_f32", dg0.device, C.byref(d))


def textcnn_head(convs, Ps, B, W, bias, c, w_ce, feat, arg, stats):
    """ReLU + max over time + fc1 + CE(c.long()) + the way back into dconv matrices, one launch (cgn_textcnn_head_f32).
    convs[i]: [B*Ps[i], Co_i] contiguous; c: [B] float view (any stride). Returns the list of dconv matrices."""
    n = len(convs)
    _req(1 <= n <= 8 and len(Ps) == n, "1..8 widths")
    F = sum(cv.shape[1] for cv in convs)
    _chk_f32(W, "W"), _chk_f32(bias, "bias"), _chk_f32(c, "c"), _chk_f32(feat, "feat"), _chk_f32(stats, "stats")
    _req(W.is_contiguous() and W.shape == (2, F) and bias.numel() == 2 and c.dim() == 1 and c.numel() == B
         and feat.is_contiguous() and feat.shape == (B, F) and arg.is_cuda and arg.dtype == torch.int32
         and arg.is_contiguous() and arg.shape == (B, F) and stats.is_contiguous() and stats.numel() >= 1
         and F <= 1024 and B + 1 <= FUSED_SCRATCH_WORDS, "textcnn_head shapes")
    dconvs = []
    cp, dp = (C.c_void_p * n)(), (C.c_void_p * n)()
    P_, Co_ = (C.c_int32 * n)(), (C.c_int32 * n)()
    for i, cv in enumerate(convs):
        _chk_f32(cv, "conv")
        _req(cv.is_contiguous() and cv.dim() == 2 and cv.shape[0] == B * Ps[i], "conv %d must be [B*P, Co]" % i)
        dc = alloc(tuple(cv.shape), cv.device)
        dconvs.append(dc)
        cp[i], dp[i], P_[i], Co_[i] = cv.data_ptr(), dc.data_ptr(), Ps[i], cv.shape[1]
    ws = _fused_scratch_ws(feat.device)
    _call("cgn_textcnn_head_f32", feat.device, n, cp, dp, P_, Co_, B, _ptr(W), _ptr(bias), _ptr(c),
          c.stride(0) if B > 1 else 1, float(w_ce), _ptr(feat), _ptr(arg), _ptr(stats), _ptr(ws), ws.numel() * 4)
    return dconvs


def conv_dx_from_dconv(Ks, dconvs, W, B, L, E):
    """dcols_w = dconv_w @ W_w (layout 0 weights [Co, K*E]) for every width in ONE grouped MFMA launch; returns dcols."""
    dev = dconvs[0].device
    calls, dcols = [], []
    for K, dc, w in zip(Ks, dconvs, W):
        Co = dc.shape[1]
        P = L - K + 1
        _req(w.numel() == Co * K * E and w.is_contiguous() and dc.shape[0] == B * P, "conv weight / dconv shapes")
        dl = alloc((B * P, K * E), dev)
        calls.append((dc, w, dl, dict(M=B * P, N=K * E, K=Co, a_m=Axis(Co, 0, 0), a_k=Axis(1, 0, 0),
                                      b_k=Axis(K * E, 0, 0), b_n=Axis(1, 0, 0), compute="f32")))
        dcols.append(dl)
    gemm_group(calls)
    return dcols


def col2im_rowdot(Ks, dcols, B, L, T, E, dx, other, rowdot, beta=0.0):
    """dx [B*T, E] (contiguous): rows t < L = summed windows (+ beta*dx), rows t >= L zero (+ beta*dx);
    rowdot[r] = dx[r] . other[r] (cgn_conv_col2im_rowdot_f32)."""
    n = len(Ks)
    _chk_f32(dx, "dx"), _chk_f32(other, "other"), _chk_f32(rowdot, "rowdot")
    _req(dx.is_contiguous() and dx.shape == (B * T, E) and other.dim() == 2 and other.shape == (B * T, E)
         and other.stride(1) == 1 and rowdot.is_contiguous() and rowdot.numel() == B * T, "col2im_rowdot shapes")
    ptrs = (C.c_void_p * n)()
    Karr = (C.c_int32 * n)(*Ks)
    for i, (K, dl) in enumerate(zip(Ks, dcols)):
        _chk_f32(dl, "dcols")
        _req(dl.is_contiguous() and dl.shape == (B * (L - K + 1), K * E), "dcols %d shape" % i)
        ptrs[i] = dl.data_ptr()
    _call("cgn_conv_col2im_rowdot_f32", dx.device, n, ptrs, Karr, B, L, T, E, float(beta), _ptr(dx), _ptr(other),
          other.stride(0), _ptr(rowdot))


# Phase stamps (measurement aid, tools/step_stamps.py): while STAMPS is a dict {"buf": int64 device tensor, "names": []}
# every stamp(name) call appends a 1-thread kernel that writes the device wall clock into the next slot when the CURRENT
# stream gets there - also inside a captured graph, so an un-profiled replay can be timed phase by phase. None = no-op.
STAMPS = None


def stamp(name):
    st = STAMPS
    if st is None:
        return
    i = len(st["names"])
    if i >= st["buf"].numel():
        return
    st["names"].append(name)
    _call("cgn_stamp_u64", st["buf"].device, C.c_void_p(st["buf"].data_ptr() + 8 * i))


def incr(counter):
    """counter[0] += 1 on the current stream (device-resident Adam step count)."""
    _req(counter.is_cuda and counter.dtype == torch.int32 and counter.numel() == 1, "counter must be one device int32")
    _call("cgn_incr_i32", counter.device, _ptr(counter))


def adam(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, dyn_step=None):
    n = p.numel()
    for t, nme in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _chk_f32(t, nme)
        _req(t.is_contiguous() and t.numel() == n, "adam arenas must be contiguous and equally sized")
    _call("cgn_adam_f32", p.device, _ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, beta1, beta2, eps, int(step),
          _ptr(dyn_step), grad_scale)


def disc_head(feat, W, bias, logits, *, mask=None, y=None, w_ce=0.0, w_ent=0.0, p1=None, dlogits=None, stats=None):
    ldf = _rows2d(feat, "feat")
    B, F = feat.shape
    _chk_f32(W, "W"), _chk_f32(bias, "bias"), _chk_f32(logits, "logits")
    _req(W.is_contiguous() and W.shape == (2, F) and bias.numel() == 2 and logits.is_contiguous()
         and logits.shape == (B, 2), "disc_head shapes")
    if mask is not None:
        _chk_f32(mask, "mask")
        _req(mask.is_contiguous() and mask.shape == (B, F), "mask must be [B,F]")
    if y is not None:
        _chk_i64(y, "y")
        _req(y.numel() == B, "y must have B elements")
    for t, n, ne in ((p1, "p1", B), (dlogits, "dlogits", 2 * B), (stats, "stats", 2)):
        if t is not None:
            _chk_f32(t, n)
            _req(t.is_contiguous() and t.numel() == ne, n + " size")
    _call("cgn_disc_head_f32", feat.device, _ptr(feat), ldf, _ptr(mask), _ptr(W), _ptr(bias), _ptr(y), B, F, w_ce, w_ent,
          _ptr(logits), _ptr(p1), _ptr(dlogits), _ptr(stats))


def disc_head_bwd(feat, W, dlogits, *, mask=None, beta=0.0, dW=None, db=None, dfeat=None):
    ldf = _rows2d(feat, "feat")
    B, F = feat.shape
    _chk_f32(W, "W"), _chk_f32(dlogits, "dlogits")
    _req(W.is_contiguous() and W.shape == (2, F) and dlogits.is_contiguous() and dlogits.numel() == 2 * B,
         "disc_head_bwd shapes")
    for t, n, ne in ((dW, "dW", 2 * F), (db, "db", 2), (dfeat, "dfeat", B * F), (mask, "mask", B * F)):
        if t is not None:
            _chk_f32(t, n)
            _req(t.is_contiguous() and t.numel() == ne, n + " size")
    _call("cgn_disc_head_bwd_f32", feat.device, _ptr(feat), ldf, _ptr(mask), _ptr(W), _ptr(dlogits), B, F, beta,
          _ptr(dW), _ptr(db), _ptr(dfeat))


def bernoulli(p1, u, c):
    for t, n in ((p1, "p1"), (u, "u"), (c, "c")):
        _chk_f32(t, n)
        _req(t.is_contiguous() and t.numel() == p1.numel(), n + " size")
    _call("cgn_bernoulli_f32", c.device, _ptr(p1), _ptr(u), p1.numel(), _ptr(c))
    return c


def act_maxpool(conv, B, P, act, feat, col0, arg):
    """conv [B*P, Co] contiguous -> feat[:, col0:col0+Co], arg[:, col0:col0+Co] (int32)."""
    _chk_f32(conv, "conv")
    Co = conv.shape[1]
    ldf = _rows2d(feat, "feat")
    _req(conv.is_contiguous() and conv.shape[0] == B * P and feat.shape[0] == B and feat.shape[1] >= col0 + Co,
         "act_maxpool shapes")
    _req(arg.is_cuda and arg.dtype == torch.int32 and arg.is_contiguous() and arg.shape == feat.shape, "arg must be int32")
    _call("cgn_act_maxpool_f32", feat.device, _ptr(conv), B, P, Co, act, _ptr(feat), ldf, col0, _ptr(arg), arg.stride(0))


def _conv_widths(Ks, Cos, act, layout, W=None, dW=None, db=None):
    cw = _lib.ConvWidths()
    n = len(Ks)
    _req(1 <= n <= 8, "1..8 widths")
    col = 0
    for i in range(n):
        cw.K[i], cw.Co[i], cw.col0[i] = Ks[i], Cos[i], col
        col += Cos[i]
        for arr, name, ne in ((W, "W", None), (dW, "dW", None), (db, "db", Cos[i])):
            if arr is not None and arr[i] is not None:
                _chk_f32(arr[i], name)
                _req(arr[i].is_contiguous(), name + " must be contiguous")
                getattr(cw, name)[i] = arr[i].data_ptr()
    cw.n, cw.F, cw.act, cw.layout = n, col, act, layout
    return cw


def conv_bwd_dx(Ks, Cos, act, layout, W, dfeat, feat, arg, B, L, E, dx, sent_stride):
    """dx[b, 0:L, 0:E] (sentence stride given) from the pooled-feature gradient dfeat [B,F]."""
    for i, w in enumerate(W):
        _req(w.numel() == Cos[i] * Ks[i] * E and Ks[i] <= L, "conv weight %d size" % i)
    cw = _conv_widths(Ks, Cos, act, layout, W=W)
    _chk_f32(dfeat, "dfeat"), _chk_f32(dx, "dx")
    _req(dfeat.is_contiguous() and dfeat.shape == (B, cw.F) and arg.dtype == torch.int32 and arg.is_contiguous()
         and arg.shape == (B, cw.F), "conv_bwd_dx dfeat/arg shapes")
    if feat is not None:
        _req(feat.is_contiguous() and feat.shape == (B, cw.F), "feat shape")
    nelem = dx.untyped_storage().nbytes() // 4 - dx.storage_offset()
    _req(sent_stride >= L * E and (B - 1) * sent_stride + L * E <= nelem, "dx too small")
    _call("cgn_conv_bwd_dx_f32", dx.device, C.byref(cw), _ptr(dfeat), _ptr(feat), _ptr(arg), B, L, E, _ptr(dx),
          sent_stride)


def conv_bwd_dw(Ks, Cos, act, layout, dW, db, dfeat, feat, arg, x, sent_stride, B, L, E, beta=0.0):
    for i, w in enumerate(dW):
        _req(w.numel() == Cos[i] * Ks[i] * E and Ks[i] <= L, "conv dW %d size" % i)
    cw = _conv_widths(Ks, Cos, act, layout, dW=dW, db=db)
    _chk_f32(dfeat, "dfeat"), _chk_f32(x, "x")
    _req(dfeat.is_contiguous() and dfeat.shape == (B, cw.F) and arg.dtype == torch.int32 and arg.is_contiguous()
         and arg.shape == (B, cw.F), "conv_bwd_dw dfeat/arg shapes")
    if feat is not None:
        _req(feat.is_contiguous() and feat.shape == (B, cw.F), "feat shape")
    nelem = x.untyped_storage().nbytes() // 4 - x.storage_offset()
    _req(sent_stride >= L * E and (B - 1) * sent_stride + L * E <= nelem, "x too small")
    _call("cgn_conv_bwd_dw_f32", x.device, C.byref(cw), _ptr(dfeat), _ptr(feat), _ptr(arg), _ptr(x), sent_stride, B, L,
          E, beta)


def embedding_bwd(idx, dx, dtable, beta=0.0):
    _chk_i64(idx, "idx"), _chk_f32(dx, "dx"), _chk_f32(dtable, "dtable")
    V, E = dtable.shape
    n = idx.numel()
    _req(dx.is_contiguous() and dx.numel() == n * E and dtable.is_contiguous(), "embedding_bwd shapes")
    ws = workspace(dx.device)
    _call("cgn_embedding_bwd_f32", dx.device, _ptr(idx), _ptr(dx), n, V, E, beta, _ptr(dtable), _ptr(ws), ws.numel() * 4)


# Persistent LSTM launches (one launch per sequence pass, csrc/lstm.hip) need a small scratch for their arrival counters,
# zeroed ONCE here (every launch leaves its counters at zero again): one per (device, stream), like the split-K
# workspace. LSTM_PERSISTENT=False (or CGN_LSTM_PERSISTENT=0) forces the one-launch-per-timestep kernels, which take
# every shape.
import os as _os
LSTM_PERSISTENT = _os.environ.get("CGN_LSTM_PERSISTENT", "1") != "0"
_lstm_sync = {}
LSTM_SYNC_BYTES = 64 << 10


def _lstm_sync_ws(device):
    idx = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream)
    ws = _lstm_sync.get(key)
    if ws is None:
        with torch.cuda.stream(torch.cuda.default_stream(idx)):
            ws = torch.zeros(LSTM_SYNC_BYTES // 4, dtype=torch.int32, device=torch.device("cuda", idx))
        _lstm_sync[key] = ws
    return ws


def lstm_persistent_bytes(nseq, B, T, H, dev):
    """Scratch bytes a persistent launch of this shape needs on `dev`, 0 = not eligible (use the step kernels)."""
    if not LSTM_PERSISTENT:
        return 0
    n = C.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().cgn_lstm_persistent_query(nseq, B, T, H, C.byref(n)), "cgn_lstm_persistent_query")
    return n.value if n.value <= LSTM_SYNC_BYTES else 0


def lstm_persistent_group(nseq, B, T, H, dev):
    """Sequences per persistent launch: nseq when they all fit the CUs together, 1 when only one sequence at a time does
    (row-tiled kernels at H=1024 / B=256 use every CU for one sequence), 0 = the per-timestep kernels."""
    if lstm_persistent_bytes(nseq, B, T, H, dev):
        return nseq
    return 1 if nseq > 1 and lstm_persistent_bytes(1, B, T, H, dev) else 0


def lstm_counters_at_rest(dev=None):
    """True when every arrival / exit counter of every persistent-LSTM scratch buffer (of `dev`, or of all devices) reads
    zero - the state each launch must leave behind (csrc/lstm.hip persist_checkout), timed-out launches included.
    Synchronises; for tests."""
    idx = None if dev is None else (torch.device(dev).index if torch.device(dev).index is not None
                                    else torch.cuda.current_device())
    torch.cuda.synchronize()
    return all(int(ws[:-32].abs().max().item()) == 0 for (i, _), ws in _lstm_sync.items() if idx is None or i == idx)


def lstm_sticky_timeouts():
    """Number of persistent-LSTM scratch buffers (one per device and stream) whose STICKY timeout flag is set: non-zero
    means some persistent launch since start-up ran out of a bounded wait and produced invalid results (a co-scheduled
    kernel kept its workgroups from becoming resident together). One device read per buffer; synchronises."""
    return sum(int(ws[-32].item() != 0) for ws in _lstm_sync.values())


class LstmTimeoutError(RuntimeError):
    pass


def raise_on_lstm_timeout():
    """Product-path check (train_functions.py, the step closures' logging points, the sampler, checkpoint save): a
    persistent LSTM launch whose bounded wait ran out has produced invalid activations, and every step since then has
    trained on them. Raises LstmTimeoutError; the caller decides (ops.LSTM_PERSISTENT = False takes the per-timestep
    kernels). Synchronises (one word per scratch buffer): call it where the host already waits for the device."""
    if lstm_sticky_timeouts():
        raise LstmTimeoutError(
            "a persistent LSTM launch did not get its workgroups resident together and ran out of a bounded wait: "
            "activations since then are invalid. Reload the last checkpoint and rerun with CGN_LSTM_PERSISTENT=0 "
            "(or set ldrs_amd.hip.ops.LSTM_PERSISTENT = False).")


def clear_lstm_sticky_timeouts():
    for ws in _lstm_sync.values():
        ws[-32:].zero_()


# the persistent recurrences hand h_t / dgates_t over through a blocked scratch (cgn_lstm_fwd_seq.handoff: whole cache
# lines per load); CGN_LSTM_HANDOFF=0 keeps the loads on the history buffers (A/B switch; same bits at H <= 512)
LSTM_HANDOFF = os.environ.get("CGN_LSTM_HANDOFF", "1") != "0"


def _lstm_handoff_wanted(H):
    """Every persistent launch brings a scratch (include/cgn_hip.h cgn_lstm_fwd_seq.handoff): the library holds fp32 in
    it, or bf16 in the bf16 mode's plain-bf16 recurrences at H > 512 (LSTM_BF16_RT == 2, H % 256 == 0)."""
    return LSTM_HANDOFF


_lstm_handoff = {}


def _lstm_handoff_scratch(dev, kind, i, nelem):
    """bf16 hand-off scratch of sequence i of a launch. Inside a step (KEEP is a list) a fresh step-scoped tensor, like
    every other temporary; otherwise one buffer per (device, stream, kind, i) that lives as long as the process: a
    temporary would go back to torch's allocator before the launch has run and could be handed out again right away
    (the first call did exactly that: the LSTM counter workspace, allocated next, got the block)."""
    if KEEP is not None:
        return alloc((nelem,), dev, torch.bfloat16)
    idx = torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device()
    key = (idx, torch.cuda.current_stream(idx).cuda_stream, kind, i)
    t = _lstm_handoff.get(key)
    if t is None or t.numel() < nelem:
        t = _lstm_handoff[key] = torch.empty(nelem, dtype=torch.bfloat16, device=torch.device("cuda", idx))
    return t


def lstm_seq_fwd(seqs, B, T, H):
    """seqs: list of dicts(xg, xg_extra=None, w_hh, hbuf, cbuf, gates, reverse)."""
    arr = (_lib.LstmFwdSeq * len(seqs))()
    dev = seqs[0]["hbuf"].device
    for i, s in enumerate(seqs):
        for k, shape in (("xg", (B, T, 4 * H)), ("w_hh", (4 * H, H)), ("hbuf", (B, T + 1, H)), ("cbuf", (B, T + 1, H)),
                         ("gates", (B, T, 4 * H))):
            _chk_f32(s[k], k)
            _req(s[k].is_contiguous() and tuple(s[k].shape) == shape, "lstm %s must be contiguous %s" % (k, shape))
        ex = s.get("xg_extra")
        if ex is not None:
            _chk_f32(ex, "xg_extra")
            _req(ex.is_contiguous() and tuple(ex.shape) == (B, 4 * H), "xg_extra must be [B,4H]")
        arr[i].xg, arr[i].xg_extra, arr[i].w_hh = s["xg"].data_ptr(), (ex.data_ptr() if ex is not None else None), \
            s["w_hh"].data_ptr()
        arr[i].hbuf, arr[i].cbuf, arr[i].gates = s["hbuf"].data_ptr(), s["cbuf"].data_ptr(), s["gates"].data_ptr()
        arr[i].reverse = 1 if s.get("reverse") else 0
        arr[i].init_zero = 1 if s.get("init_zero") else 0
    grp = lstm_persistent_group(len(seqs), B, T, H, dev)
    if grp:
        if _lstm_handoff_wanted(H):
            for i in range(len(seqs)):  # bf16 hand-off scratch [2][H/8][B][8] per sequence (operand mode 2)
                arr[i].handoff = _lstm_handoff_scratch(dev, "fwd", i, 4 * pad16(B) * H).data_ptr()  # (2*B16*H*4 bytes)
        ws = _lstm_sync_ws(dev)
        for i in range(0, len(seqs), grp):
            sub = (_lib.LstmFwdSeq * grp).from_buffer(arr, i * C.sizeof(_lib.LstmFwdSeq))
            _call("cgn_lstm_seq_fwd_persistent_f32", dev, sub, grp, B, T, H, _ptr(ws), ws.numel() * 4)
    else:
        _call("cgn_lstm_seq_fwd_f32", dev, arr, len(seqs), B, T, H)


def lstm_seq_bwd(seqs, B, T, H):
    """seqs: list of dicts(w_hh_t | w_hh, cbuf, gates, dgates, dc_state, dh_ext=None (+dh_bs, dh_ts), dh_seed=None,
    reverse). w_hh (untransposed [4H,H]) is enough when the shape takes the persistent kernel."""
    arr = (_lib.LstmBwdSeq * len(seqs))()
    dev = seqs[0]["dgates"].device
    grp = nb = lstm_persistent_group(len(seqs), B, T, H, dev)
    for i, s in enumerate(seqs):
        w, wt = s.get("w_hh"), s.get("w_hh_t")
        _req(wt is not None or (w is not None and nb), "lstm_seq_bwd needs w_hh_t (or w_hh on the persistent path)")
        if w is not None:
            _chk_f32(w, "w_hh")
            _req(w.is_contiguous() and tuple(w.shape) == (4 * H, H), "w_hh must be contiguous [4H,H]")
        if wt is not None:
            _chk_f32(wt, "w_hh_t")
            _req(wt.is_contiguous() and tuple(wt.shape) == (H, 4 * H), "w_hh_t must be contiguous [H,4H]")
        arr[i].w_hh = w.data_ptr() if w is not None else None
        arr[i].w_hh_t = wt.data_ptr() if wt is not None else None
        for k, shape in (("cbuf", (B, T + 1, H)), ("gates", (B, T, 4 * H)),
                         ("dgates", (B, T, 4 * H)), ("dc_state", (B, H))):
            _chk_f32(s[k], k)
            _req(s[k].is_contiguous() and tuple(s[k].shape) == shape, "lstm %s must be contiguous %s" % (k, shape))
        ext, seed = s.get("dh_ext"), s.get("dh_seed")
        if ext is not None:
            _chk_f32(ext, "dh_ext")
            bs, ts = int(s["dh_bs"]), int(s["dh_ts"])
            nelem = ext.untyped_storage().nbytes() // 4 - ext.storage_offset()
            _req(bs >= 0 and ts >= 0 and (B - 1) * bs + (T - 1) * ts + H <= nelem, "dh_ext addressing out of range")
            arr[i].dh_bs, arr[i].dh_ts = bs, ts
        if seed is not None:
            _chk_f32(seed, "dh_seed")
            _req(seed.is_contiguous() and tuple(seed.shape) == (B, H), "dh_seed must be [B,H]")
        arr[i].cbuf, arr[i].gates = s["cbuf"].data_ptr(), s["gates"].data_ptr()
        arr[i].dh_ext = ext.data_ptr() if ext is not None else None
        arr[i].dh_seed = seed.data_ptr() if seed is not None else None
        arr[i].dgates, arr[i].dc_state = s["dgates"].data_ptr(), s["dc_state"].data_ptr()
        arr[i].reverse = 1 if s.get("reverse") else 0
    if grp:
        if _lstm_handoff_wanted(H):
            for i in range(len(seqs)):
                arr[i].handoff = _lstm_handoff_scratch(dev, "bwd", i, 4 * pad16(B) * 4 * H).data_ptr()
        ws = _lstm_sync_ws(dev)
        for i in range(0, len(seqs), grp):
            sub = (_lib.LstmBwdSeq * grp).from_buffer(arr, i * C.sizeof(_lib.LstmBwdSeq))
            _call("cgn_lstm_seq_bwd_persistent_f32", dev, sub, grp, B, T, H, _ptr(ws), ws.numel() * 4)
    else:
        _call("cgn_lstm_seq_bwd_f32", dev, arr, len(seqs), B, T, H)


LSTM_FUSE_LAYERS = _os.environ.get("CGN_LSTM_FUSE_LAYERS", "1") != "0"


def lstm2_persistent_bytes(B, T, H, dev):
    """Scratch bytes of the two-layer wavefront launch for this shape on `dev`; 0 = not eligible (or switched off)."""
    if not (LSTM_PERSISTENT and LSTM_FUSE_LAYERS):
        return 0
    n = C.c_int64(0)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().cgn_lstm2_persistent_query(B, T, H, C.byref(n)), "cgn_lstm2_persistent_query")
    return n.value if n.value <= LSTM_SYNC_BYTES else 0


def _chk_shape(t, shape, name):
    _chk_f32(t, name)
    _req(t.is_contiguous() and tuple(t.shape) == tuple(shape), "lstm2 %s must be contiguous %s" % (name, tuple(shape)))


def lstm2_fwd(xg0, xg0_extra, w_hh0, w_ih1, w_hh1, b_ih1, b_hh1, hbuf0, cbuf0, gates0, hbuf1, cbuf1, gates1, B, T, H):
    """Both layers of a two-layer LSTM (zero initial state) in one persistent launch (cgn_lstm2_fwd_persistent_f32)."""
    dev = hbuf0.device
    _req(lstm2_persistent_bytes(B, T, H, dev) > 0, "shape not eligible for the two-layer launch")
    d = _lib.Lstm2Fwd()
    for name, t, shape in (("xg0", xg0, (B, T, 4 * H)), ("w_hh0", w_hh0, (4 * H, H)), ("w_ih1", w_ih1, (4 * H, H)),
                           ("w_hh1", w_hh1, (4 * H, H)), ("b_ih1", b_ih1, (4 * H,)), ("b_hh1", b_hh1, (4 * H,)),
                           ("hbuf0", hbuf0, (B, T + 1, H)), ("cbuf0", cbuf0, (B, T + 1, H)),
                           ("gates0", gates0, (B, T, 4 * H)), ("hbuf1", hbuf1, (B, T + 1, H)),
                           ("cbuf1", cbuf1, (B, T + 1, H)), ("gates1", gates1, (B, T, 4 * H))):
        _chk_shape(t, shape, name)
        setattr(d, name, t.data_ptr())
    if xg0_extra is not None:
        _chk_shape(xg0_extra, (B, 4 * H), "xg0_extra")
        d.xg0_extra = xg0_extra.data_ptr()
    ws = _lstm_sync_ws(dev)
    _call("cgn_lstm2_fwd_persistent_f32", dev, C.byref(d), B, T, H, _ptr(ws), ws.numel() * 4)


def lstm2_bwd(w_hh0, w_ih1, w_hh1, cbuf0, gates0, cbuf1, gates1, dh_ext1, dh_bs, dh_ts, dgates0, dgates1, B, T, H):
    """BPTT through both layers in one persistent launch (cgn_lstm2_bwd_persistent_f32)."""
    dev = dgates0.device
    _req(lstm2_persistent_bytes(B, T, H, dev) > 0, "shape not eligible for the two-layer launch")
    d = _lib.Lstm2Bwd()
    for name, t, shape in (("w_hh0", w_hh0, (4 * H, H)), ("w_ih1", w_ih1, (4 * H, H)), ("w_hh1", w_hh1, (4 * H, H)),
                           ("cbuf0", cbuf0, (B, T + 1, H)), ("gates0", gates0, (B, T, 4 * H)),
                           ("cbuf1", cbuf1, (B, T + 1, H)), ("gates1", gates1, (B, T, 4 * H)),
                           ("dgates0", dgates0, (B, T, 4 * H)), ("dgates1", dgates1, (B, T, 4 * H))):
        _chk_shape(t, shape, name)
        setattr(d, name, t.data_ptr())
    if dh_ext1 is not None:
        _chk_f32(dh_ext1, "dh_ext1")
        nelem = dh_ext1.untyped_storage().nbytes() // 4 - dh_ext1.storage_offset()
        _req(dh_bs >= 0 and dh_ts >= 0 and (B - 1) * dh_bs + (T - 1) * dh_ts + H <= nelem, "dh_ext1 addressing out of range")
        d.dh_ext1, d.dh_bs, d.dh_ts = dh_ext1.data_ptr(), int(dh_bs), int(dh_ts)
    if LSTM_HANDOFF:  # blocked copies of both layers' gate gradients (2 * B*T*4H fp32 = 4 * B*T*4H bf16-sized elements)
        d.handoff = _lstm_handoff_scratch(dev, "bwd2", 0, 4 * B * T * 4 * H).data_ptr()
    ws = _lstm_sync_ws(dev)
    _call("cgn_lstm2_bwd_persistent_f32", dev, C.byref(d), B, T, H, _ptr(ws), ws.numel() * 4)


def lincomb(terms, out=None):
    """out[0] = sum coef*src[0] over terms [(coef, tensor), ...] (<= 4): scalar loss totals, on device."""
    n = len(terms)
    _req(1 <= n <= 4, "1..4 terms")
    dev = terms[0][1].device
    if out is None:
        out = alloc((1,), dev)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for _, t in terms])
    coefs = (C.c_float * n)(*[float(c) for c, _ in terms])
    for _, t in terms:
        _chk_f32(t, "term")
        _req(t.numel() >= 1, "term must hold a scalar")
    _call("cgn_lincomb_f32", dev, n, ptrs, coefs, _ptr(out))
    return out


def copy3d(src, dst, beta=0.0):
    """dst[b,t,:] = src[b,t,:] + beta*dst[b,t,:] for [B,T,H] views with unit last stride (any batch/time strides)."""
    _chk_f32(src, "src"), _chk_f32(dst, "dst")
    _req(src.dim() == 3 and src.shape == dst.shape and src.stride(2) == 1 and dst.stride(2) == 1, "copy3d shapes")
    B, T, H = src.shape
    _call("cgn_copy3d_f32", dst.device, _ptr(src), src.stride(0), src.stride(1), _ptr(dst), dst.stride(0),
          dst.stride(1), B, T, H, beta)
    return dst


def add3d(src, dst):
    return copy3d(src, dst, beta=1.0)


def float_to_index(x, out):
    _chk_f32(x, "x"), _chk_i64(out, "out")
    _req(x.is_contiguous() and out.numel() == x.numel(), "float_to_index sizes")
    _call("cgn_float_to_index_f32", x.device, _ptr(x), x.numel(), _ptr(out))
    return out


def mahalanobis(mulv1, mulv2, weight, z_loss, dmulv2):
    B, Z2 = mulv1.shape
    for t, n in ((mulv1, "mulv1"), (mulv2, "mulv2"), (dmulv2, "dmulv2"), (z_loss, "z_loss")):
        _chk_f32(t, n)
        _req(t.is_contiguous(), n + " must be contiguous")
    _req(mulv2.shape == mulv1.shape and dmulv2.shape == mulv1.shape and z_loss.numel() == 1, "mahalanobis shapes")
    _call("cgn_mahalanobis_f32", mulv1.device, _ptr(mulv1), _ptr(mulv2), B, Z2 // 2, weight, _ptr(z_loss), _ptr(dmulv2))


def mul2d(a, b, out):
    for t, n in ((a, "a"), (b, "b"), (out, "out")):
        _chk_f32(t, n)
        _req(t.is_contiguous() and t.numel() == a.numel(), n + " must be contiguous, same size")
    _call("cgn_mul_f32", out.device, _ptr(a), _ptr(b), _ptr(out), a.numel())
    return out


def beam_topk(logits, lse, S, K, first, eos, pad, scores, done, length, tokens, origin, src_rows, hist_tok, hist_org):
    """One beam-search step (cgn_beam_topk_f32): see include/cgn_hip.h. All state tensors live on the device."""
    _chk_f32(logits, "logits"), _chk_f32(lse, "lse"), _chk_f32(scores, "scores")
    R = S * K
    _req(logits.dim() == 2 and logits.shape[0] == R and logits.stride(1) == 1, "logits must be [S*K, V]")
    V = logits.shape[1]
    _req(lse.is_contiguous() and lse.numel() == R and scores.is_contiguous() and scores.numel() == R, "lse/scores size")
    for t_, n, dt, ne in ((done, "done", torch.int32, S), (length, "length", torch.int32, S),
                          (tokens, "tokens", torch.int64, R), (origin, "origin", torch.int32, R),
                          (src_rows, "src_rows", torch.int64, R), (hist_tok, "hist_tok", torch.int64, R),
                          (hist_org, "hist_org", torch.int32, R)):
        _req(t_.is_cuda and t_.dtype == dt and t_.is_contiguous() and t_.numel() == ne, "%s must be %s[%d]" % (n, dt, ne))
    _req(1 <= K <= 64 and 0 <= eos < V and 0 <= pad < V, "beam_topk arguments")
    cand_val = alloc((R * K,), logits.device)
    cand_idx = alloc((R * K,), logits.device, torch.int32)
    _call("cgn_beam_topk_f32", logits.device, _ptr(logits), logits.stride(0), _ptr(lse), S, K, V, 1 if first else 0,
          int(eos), int(pad), _ptr(scores), _ptr(done), _ptr(length), _ptr(tokens), _ptr(origin), _ptr(src_rows),
          _ptr(hist_tok), _ptr(hist_org), _ptr(cand_val), _ptr(cand_idx))


def conv_bwd_dx_dense(Ks, Cos, act, layout, W, dfeat, feat, arg, B, L, E, dx, sent_stride, dW=None, x=None, beta=0.0):
    """Same result as conv_bwd_dx through the dense route (include/cgn_hip.h): scatter the pooled gradient over the
    convolution outputs, ONE grouped MFMA launch dcols_w = dconv_w @ W_w over the widths, col2im.
    W[i]: weight of width i, [Co, K*E] rows (layout 0) or [Co, E, K] (layout 1, torch conv1d).
    With dW (list, laid out like W) and x (the convolution's input, element (b,t,e) at b*sent_stride + t*E + e) the
    weight gradients dW_w = beta*dW_w + dconv_w^T @ windows_w(x) come from a second grouped MFMA launch over the SAME
    scattered matrices (same result as conv_bwd_dw without bias gradients; K = B*P is split inside the launch)."""
    n = len(Ks)
    cw = _conv_widths(Ks, Cos, act, layout)
    _chk_f32(dfeat, "dfeat"), _chk_f32(dx, "dx")
    _req(dfeat.is_contiguous() and dfeat.shape == (B, cw.F) and arg.dtype == torch.int32 and arg.is_contiguous()
         and arg.shape == (B, cw.F), "conv_bwd_dx_dense dfeat/arg shapes")
    if feat is not None:
        _req(feat.is_contiguous() and feat.shape == (B, cw.F), "feat shape")
    nelem = dx.untyped_storage().nbytes() // 4 - dx.storage_offset()
    _req(sent_stride >= L * E and (B - 1) * sent_stride + L * E <= nelem, "dx too small")
    dev = dx.device
    dconvs, dcols, calls = [], [], []
    for i in range(n):
        K, Co = Ks[i], Cos[i]
        _req(W[i].numel() == Co * K * E and K <= L and W[i].is_contiguous(), "conv weight %d" % i)
        P = L - K + 1
        dc = alloc((B * P, Co), dev)
        dl = alloc((B * P, K * E), dev)
        if layout == 0:
            kw = dict(M=B * P, N=K * E, K=Co, b_k=Axis(K * E, 0, 0), b_n=Axis(1, 0, 0))
        else:  # element (o, c, j) of [Co, E, K] -> column n = j*E + c
            kw = dict(M=B * P, N=K * E, K=Co, b_k=Axis(K * E, 0, 0), b_n=Axis(K, E, 1))
        kw.update(a_m=Axis(Co, 0, 0), a_k=Axis(1, 0, 0), compute="f32")  # convolutions stay fp32 in the bf16 mode
        calls.append((dc, W[i], dl, kw))
        dconvs.append(dc), dcols.append(dl)
    ptrs = (C.c_void_p * n)(*[t_.data_ptr() for t_ in dconvs])
    _call("cgn_pool_bwd_scatter_f32", dev, C.byref(cw), _ptr(dfeat), _ptr(feat), _ptr(arg), B, L, ptrs)
    if dW is not None:
        _chk_f32(x, "x")
        nx = x.untyped_storage().nbytes() // 4 - x.storage_offset()
        _req((B - 1) * sent_stride + L * E <= nx, "x too small")
        wcalls = []
        for i in range(n):
            K, Co = Ks[i], Cos[i]
            P = L - K + 1
            _req(dW[i].numel() == Co * K * E and dW[i].is_contiguous(), "conv dW %d" % i)
            # A(m = map, k = (b,p)) = dconv[(b,p), map];  B(k = (b,p), n) = x[b, p + n // E, n % E] for layout 0,
            # for layout 1 column n = c*K + j of the [Co, E, K] weight reads window element j*E + c
            kw = dict(M=Co, N=K * E, K=B * P, a_m=Axis(1, 0, 0), a_k=Axis(Co, 0, 0), b_k=Axis(E, P, sent_stride),
                      b_n=(Axis(1, 0, 0) if layout == 0 else Axis(E, K, 1)), beta=beta, compute="f32")
            wcalls.append((dconvs[i], x, dW[i].view(Co, K * E), kw))
        gemm_group(wcalls)
    gemm_group(calls)
    cptrs = (C.c_void_p * n)(*[t_.data_ptr() for t_ in dcols])
    Karr = (C.c_int32 * n)(*Ks)
    _call("cgn_conv_col2im_f32", dev, n, cptrs, Karr, B, L, E, _ptr(dx), sent_stride)


if COMPUTE == _lib.COMPUTE_BF16:
    _sync_lstm_operand_mode()
