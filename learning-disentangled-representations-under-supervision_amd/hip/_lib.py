"""ctypes binding of libcgn_hip.so (the C-ABI declared in include/cgn_hip.h).

There is deliberately NO fallback: if the shared library is missing or a symbol is absent, importing the ops or
calling them raises. The product path never routes through torch ops or the CPU oracle.
"""
import ctypes as C
import os

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("CGN_HIP_LIB") or os.path.join(PKG_DIR, "lib", "libcgn_hip.so")  # env: A/B builds

CGN_OK = 0
CGN_ERR_INVALID_ARG = 1
CGN_ERR_WORKSPACE = 2
CGN_ERR_HIP_BASE = 1000

ATRANS_NONE = 0
ATRANS_EXP = 1
EPI_LINEAR = 0
EPI_SOFTMAX_BWD = 1
EPI_LINEAR_STATS = 2
COMPUTE_F32 = 0
COMPUTE_BF16 = 1
COMPUTE_F32X6 = 2
STORE_A_BF16, STORE_C_BF16, STORE_EP_BF16 = 1, 2, 4

c_float_p = C.c_void_p  # raw device pointers travel as integers
c_i64 = C.c_int64
c_i32 = C.c_int32
c_f32 = C.c_float
c_ptr = C.c_void_p


class Axis(C.Structure):
    _fields_ = [("stride", c_i64), ("rpg", c_i64), ("gs", c_i64)]


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", c_ptr), ("B", c_ptr), ("C", c_ptr),
        ("bias0", c_ptr), ("bias1", c_ptr), ("a_row_param", c_ptr),
        ("ep_logits", c_ptr), ("ep_lse", c_ptr), ("ep_rowdot", c_ptr), ("ep_target", c_ptr),
        ("workspace", c_ptr), ("workspace_bytes", c_i64),
        ("ldc", c_i64), ("ep_ld", c_i64),
        ("a_m", Axis), ("a_k", Axis), ("b_k", Axis), ("b_n", Axis),
        ("M", c_i32), ("N", c_i32), ("K", c_i32),
        ("a_transform", c_i32), ("epilogue", c_i32),
        ("alpha", c_f32), ("beta", c_f32), ("a_scale", c_f32),
        ("ep_inv_tau", c_f32), ("ep_ce_scale", c_f32),
        ("force_tile", c_i32), ("force_splitk", c_i32),
        ("dyn_scalars", c_ptr),
        ("compute", c_i32), ("storage", c_i32),
        ("c_t", c_ptr), ("ld_ct", c_i64),
        ("ep_stats", c_ptr), ("ep_stats_ld", c_i64),
    ]


class GemmBBDesc(C.Structure):
    _fields_ = [
        ("A", c_ptr), ("B", c_ptr), ("C", c_ptr), ("bias0", c_ptr), ("bias1", c_ptr),
        ("workspace", c_ptr), ("workspace_bytes", c_i64),
        ("lda", c_i64), ("ldb", c_i64), ("ldc", c_i64),
        ("M", c_i32), ("N", c_i32), ("K", c_i32),
        ("alpha", c_f32), ("beta", c_f32),
        ("c_bf16", c_i32), ("force_tile", c_i32), ("force_splitk", c_i32),
        ("a_transform", c_i32), ("epilogue", c_i32),
        ("a_scale", c_f32), ("ep_inv_tau", c_f32), ("ep_ce_scale", c_f32),
        ("a_row_param", c_ptr), ("ep_logits", c_ptr), ("ep_lse", c_ptr), ("ep_rowdot", c_ptr), ("ep_target", c_ptr),
        ("ep_ld", c_i64), ("c_t", c_ptr), ("ld_ct", c_i64), ("dyn_scalars", c_ptr),
    ]


class LstmFwdSeq(C.Structure):
    _fields_ = [("xg", c_ptr), ("xg_extra", c_ptr), ("w_hh", c_ptr), ("hbuf", c_ptr), ("cbuf", c_ptr),
                ("gates", c_ptr), ("reverse", c_i32), ("init_zero", c_i32), ("handoff", c_ptr)]


class LstmBwdSeq(C.Structure):
    _fields_ = [("w_hh_t", c_ptr), ("w_hh", c_ptr), ("cbuf", c_ptr), ("gates", c_ptr), ("dh_ext", c_ptr), ("dh_seed", c_ptr),
                ("dgates", c_ptr), ("dc_state", c_ptr), ("dh_bs", c_i64), ("dh_ts", c_i64), ("reverse", c_i32),
                ("handoff", c_ptr)]


class Lstm2Fwd(C.Structure):
    _fields_ = [(n, c_ptr) for n in ("xg0", "xg0_extra", "w_hh0", "w_ih1", "w_hh1", "b_ih1", "b_hh1", "hbuf0", "cbuf0",
                                     "gates0", "hbuf1", "cbuf1", "gates1")]


class Lstm2Bwd(C.Structure):
    _fields_ = [(n, c_ptr) for n in ("w_hh0", "w_ih1", "w_hh1", "cbuf0", "gates0", "cbuf1", "gates1", "dh_ext1",
                                     "dgates0", "dgates1")] + [("dh_bs", c_i64), ("dh_ts", c_i64), ("handoff", c_ptr)]


class CodeHeadFwd(C.Structure):
    _fields_ = [("cf", c_ptr), ("cr", c_ptr), ("ldc", c_i64), ("W2", c_ptr), ("ldw2", c_i64), ("b2", c_ptr),
                ("eps", c_ptr), ("c", c_ptr), ("Wzc", c_ptr), ("ldzc", c_i64), ("ctx", c_ptr), ("mulv", c_ptr),
                ("zc", c_ptr), ("stdv", c_ptr), ("kld", c_ptr), ("zg", c_ptr), ("scratch", c_ptr),
                ("scratch_bytes", c_i64), ("B", c_i32), ("He", c_i32), ("Z", c_i32), ("C", c_i32), ("G4", c_i32)]


class CodeHeadBwd(C.Structure):
    _fields_ = [("dg0", c_ptr), ("Wzc", c_ptr), ("ldzc", c_i64), ("mulv", c_ptr), ("eps", c_ptr), ("dyn_kld_w", c_ptr),
                ("W2", c_ptr), ("ldw2", c_i64), ("dsum", c_ptr), ("dzc", c_ptr), ("dmulv", c_ptr), ("dcf", c_ptr),
                ("dcr", c_ptr), ("kld_w", c_f32), ("B", c_i32), ("T", c_i32), ("He", c_i32), ("Z", c_i32), ("C", c_i32),
                ("G4", c_i32)]


class ConvWidths(C.Structure):
    _fields_ = [("W", c_ptr * 8), ("dW", c_ptr * 8), ("db", c_ptr * 8),
                ("K", c_i32 * 8), ("Co", c_i32 * 8), ("col0", c_i32 * 8),
                ("n", c_i32), ("F", c_i32), ("act", c_i32), ("layout", c_i32)]


ACT_RELU = 0
ACT_TANH = 1

# name -> (restype, argtypes); mirrors include/cgn_hip.h one to one (tests check the header against this table)
PROTOTYPES = {
    "cgn_abi_version": (c_i32, []),
    "cgn_gemm_f32": (c_i32, [c_ptr, C.POINTER(GemmDesc)]),
    "cgn_gemm_timed_f32": (c_i32, [c_ptr, C.POINTER(GemmDesc), c_ptr, c_ptr, C.POINTER(c_i32)]),
    "cgn_gemm_group_f32": (c_i32, [c_ptr, C.POINTER(GemmDesc), c_i32]),
    "cgn_act_maxpool_group_f32": (c_i32, [c_ptr, c_i32, C.POINTER(c_ptr), C.POINTER(c_i32), C.POINTER(c_i32),
                                          C.POINTER(c_i32), c_i32, c_i32, c_ptr, c_i64, c_ptr, c_i64]),
    "cgn_gemm_plan": (c_i32, [C.POINTER(GemmDesc), C.POINTER(c_i32), C.POINTER(c_i32), C.POINTER(c_i32),
                              C.POINTER(c_i32)]),
    "cgn_gemm_stats_chunks": (c_i32, [C.POINTER(GemmDesc), C.POINTER(c_i32)]),
    "cgn_softmax_nll_from_stats_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_i64, c_ptr, c_i32, c_i32, c_f32, c_ptr,
                                               c_ptr, c_ptr, c_ptr, c_ptr, c_i64]),
    "cgn_softmax_nll_fwd_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_i32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "cgn_softmax_nll_fwd_bf16": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_i32, c_f32, c_ptr, c_ptr, c_ptr, c_ptr]),
    "cgn_log_softmax_rows_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_i32, c_i64, c_i32, c_ptr, c_i64]),
    "cgn_lstm_set_operand_mode": (c_i32, [c_i32]),
    "cgn_lstm_seq_fwd_f32": (c_i32, [c_ptr, C.POINTER(LstmFwdSeq), c_i32, c_i32, c_i32, c_i32]),
    "cgn_lstm_seq_bwd_f32": (c_i32, [c_ptr, C.POINTER(LstmBwdSeq), c_i32, c_i32, c_i32, c_i32]),
    "cgn_lstm_persistent_query": (c_i32, [c_i32, c_i32, c_i32, c_i32, C.POINTER(c_i64)]),
    "cgn_lstm_seq_fwd_persistent_f32": (c_i32, [c_ptr, C.POINTER(LstmFwdSeq), c_i32, c_i32, c_i32, c_i32, c_ptr, c_i64]),
    "cgn_lstm_seq_fwd_persistent_noalloc_f32": (c_i32, [c_ptr, C.POINTER(LstmFwdSeq), c_i32, c_i32, c_i32, c_i32, c_ptr]),
    "cgn_lstm_seq_bwd_persistent_f32": (c_i32, [c_ptr, C.POINTER(LstmBwdSeq), c_i32, c_i32, c_i32, c_i32, c_ptr, c_i64]),
    "cgn_lstm2_persistent_query": (c_i32, [c_i32, c_i32, c_i32, C.POINTER(c_i64)]),
    "cgn_lstm2_fwd_persistent_f32": (c_i32, [c_ptr, C.POINTER(Lstm2Fwd), c_i32, c_i32, c_i32, c_ptr, c_i64]),
    "cgn_lstm2_bwd_persistent_f32": (c_i32, [c_ptr, C.POINTER(Lstm2Bwd), c_i32, c_i32, c_i32, c_ptr, c_i64]),
    "cgn_beam_topk_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr,
                                  c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "cgn_act_maxpool_f32": (c_i32, [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_i64, c_i32, c_ptr, c_i64]),
    "cgn_pool_bwd_scatter_f32": (c_i32, [c_ptr, C.POINTER(ConvWidths), c_ptr, c_ptr, c_ptr, c_i32, c_i32,
                                         C.POINTER(c_ptr)]),
    "cgn_conv_col2im_f32": (c_i32, [c_ptr, c_i32, C.POINTER(c_ptr), C.POINTER(c_i32), c_i32, c_i32, c_i32, c_ptr, c_i64]),
    "cgn_conv_bwd_dx_f32": (c_i32, [c_ptr, C.POINTER(ConvWidths), c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr,
                                    c_i64]),
    "cgn_conv_bwd_dw_f32": (c_i32, [c_ptr, C.POINTER(ConvWidths), c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32,
                                    c_i32, c_f32]),
    "cgn_embedding_bwd_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_f32, c_ptr, c_ptr, c_i64]),
    "cgn_copy2d_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_f32, c_f32]),
    "cgn_copy3d_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_i64, c_ptr, c_i64, c_i64, c_i32, c_i32, c_i32, c_f32]),
    "cgn_float_to_index_f32": (c_i32, [c_ptr, c_ptr, c_i32, c_ptr]),
    "cgn_mahalanobis_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_f32, c_ptr, c_ptr]),
    "cgn_transpose_f32": (c_i32, [c_ptr, c_ptr, c_i32, c_i32, c_ptr]),
    "cgn_gather_rows_f32": (c_i32, [c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i64]),
    "cgn_highway_gate_fwd_f32": (c_i32, [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_i64]),
    "cgn_highway_gate_bwd_f32": (c_i32, [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_i64, c_ptr]),
    "cgn_reparam_kld_fwd_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr]),
    "cgn_reparam_kld_bwd_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_ptr, c_i32, c_i32, c_ptr]),
    "cgn_colsum_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i32, c_f32, c_ptr, c_ptr, c_i64]),
    "cgn_colsum_bf16": (c_i32, [c_ptr, c_ptr, c_i64, c_i32, c_i32, c_f32, c_ptr, c_ptr, c_i64]),
    "cgn_sum_time_f32": (c_i32, [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr]),
    "cgn_rowdot_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_i64, c_i32, c_i32, c_ptr]),
    "cgn_adam_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_ptr, c_f32]),
    "cgn_incr_i32": (c_i32, [c_ptr, c_ptr]),
    "cgn_stamp_u64": (c_i32, [c_ptr, c_ptr]),
    "cgn_code_head_fwd_f32": (c_i32, [c_ptr, C.POINTER(CodeHeadFwd)]),
    "cgn_code_head_bwd_f32": (c_i32, [c_ptr, C.POINTER(CodeHeadBwd)]),
    "cgn_textcnn_head_f32": (c_i32, [c_ptr, c_i32, C.POINTER(c_ptr), C.POINTER(c_ptr), C.POINTER(c_i32),
                                     C.POINTER(c_i32), c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_f32, c_ptr, c_ptr, c_ptr,
                                     c_ptr, c_i64]),
    "cgn_conv_col2im_rowdot_f32": (c_i32, [c_ptr, c_i32, C.POINTER(c_ptr), C.POINTER(c_i32), c_i32, c_i32, c_i32,
                                           c_i32, c_f32, c_ptr, c_ptr, c_i64, c_ptr]),
    "cgn_disc_head_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_f32, c_f32, c_ptr,
                                  c_ptr, c_ptr, c_ptr]),
    "cgn_disc_head_bwd_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_f32, c_ptr, c_ptr,
                                      c_ptr]),
    "cgn_bernoulli_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_i32, c_ptr]),
    "cgn_fill_f32": (c_i32, [c_ptr, c_ptr, c_i64, c_f32]),
    "cgn_mul_f32": (c_i32, [c_ptr, c_ptr, c_ptr, c_ptr, c_i64]),
    "cgn_lincomb_f32": (c_i32, [c_ptr, c_i32, C.POINTER(c_ptr), C.POINTER(c_f32), c_ptr]),
    "cgn_gemm_bb": (c_i32, [c_ptr, C.POINTER(GemmBBDesc), c_ptr, c_ptr, C.POINTER(c_i32), C.POINTER(c_i32)]),
    "cgn_gemm_bb_wants": (c_i32, [c_i32, c_i32, c_i32]),
    "cgn_cast_bf16_f32": (c_i32, [c_ptr, c_ptr, Axis, c_i32, c_i32, c_ptr, c_i64, c_i32]),
}

_lib = None


def load():
    """Load the shared library (once) and attach prototypes. Raises if it is missing: no fallback exists."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libcgn_hip.so not found at %s. Build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). There is no CPU/PyTorch fallback for the CGN hot path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc == CGN_OK:
        return
    if rc == CGN_ERR_INVALID_ARG:
        raise RuntimeError("%s: invalid argument (shape/alignment/null pointer)" % what)
    if rc == CGN_ERR_WORKSPACE:
        raise RuntimeError("%s: workspace too small" % what)
    raise RuntimeError("%s: HIP error %d" % (what, rc - CGN_ERR_HIP_BASE))
