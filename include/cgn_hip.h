/*
 * cgn_hip.h — C-ABI of libcgn_hip.so: the MI355X (gfx950) kernels behind the CGN wake-sleep training step.
 *
 * The reference (nikhil-dce/Learning-Disentangled-Representations-under-Supervision) has NO FFI/plugin boundary:
 * all arithmetic is stock torch ops called from Python. Each entry point below therefore cites the
 * reference call site whose torch op(s) it replaces (paths relative to the reference root).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch/C++ types; device pointers unless a comment says "host";
 *   - the caller owns every buffer, workspaces included; nothing here allocates or frees device memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are asynchronous on it;
 *   - return 0 on success, CGN_ERR_INVALID_ARG for shape/alignment/argument errors (nothing was launched),
 *     CGN_ERR_HIP_BASE + hipError_t when a launch failed; never throws;
 *   - fp32 everywhere, int64 indices (the reference's LongTensor batches, model/cgn.py:1054-1056).
 */
#ifndef CGN_HIP_H
#define CGN_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CGN_OK 0
#define CGN_ERR_INVALID_ARG 1
#define CGN_ERR_WORKSPACE 2
#define CGN_ERR_HIP_BASE 1000

/* Library/ABI version; bumped when a signature changes. */
int cgn_abi_version(void);

/* ------------------------------------------------------------------------------------------------------------
 * GEMM (fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32): C = epilogue(alpha * A·B [+ beta*C] [+ bias]).
 * Replaces every nn.Linear / t.mm / LSTM input projection / Conv2d-as-GEMM call site on the path:
 *   selfModules/highway.py:39-42, model/encoder.py:17-21 (W_ih), model/generator.py:17-22,61 (W_ih, fc),
 *   model/cgn.py:91-92 (e2mu/e2logvar), model/cgn.py:1106 (softmax @ E), model/discriminator_sentiment.py:64
 *   (Conv2d(1,Co,(K,E)) as an implicit GEMM over overlapping windows), and all their backward GEMMs.
 *
 * Element addressing is two-level on every axis so that overlapping conv windows, [B,T+1,H] history buffers and
 * transposed operands need no copies:   off(i) = (i / rpg) * gs + (i % rpg) * stride      (rpg == 0: i * stride)
 *   A(m,k) = A[off_am(m) + off_ak(k)],  B(k,n) = B[off_bk(k) + off_bn(n)],  C(m,n) = C[m*ldc + n].
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cgn_axis {
    int64_t stride;
    int64_t rpg; /* rows per group, 0 = plain linear axis */
    int64_t gs;  /* group stride (elements) */
} cgn_axis;

#define CGN_ATRANS_NONE 0
#define CGN_ATRANS_EXP 1 /* a(m,k) := exp(a(m,k)*a_scale - a_row_param[m])  (softmax probabilities on the fly) */

#define CGN_EPI_LINEAR 0      /* C = alpha*acc + beta*C + bias0[n] + bias1[n] */
#define CGN_EPI_SOFTMAX_BWD 1 /* see cgn_gemm_desc.ep_* */

typedef struct cgn_gemm_desc {
    const float* A;
    const float* B;
    float* C;
    const float* bias0;       /* nullable, length N */
    const float* bias1;       /* nullable, length N */
    const float* a_row_param; /* CGN_ATRANS_EXP: per-row log-sum-exp, length M */
    /* CGN_EPI_SOFTMAX_BWD (model/cgn.py:1086-1108 backward, fused): with p = exp(logits[m,n]*inv_tau - lse[m]),
     *   C[m,n] = p * (acc - rowdot[m]) * inv_tau + ce_scale * (p - (n == target[m]))
     * where acc = (dOut · Eᵀ)[m,n] is this GEMM's product. */
    const float* ep_logits;
    const float* ep_lse;
    const float* ep_rowdot;
    const int64_t* ep_target;
    float* workspace; /* split-K slabs; may be NULL (then no split-K) */
    int64_t workspace_bytes;
    int64_t ldc;
    int64_t ep_ld;
    cgn_axis a_m, a_k, b_k, b_n;
    int32_t M, N, K;
    int32_t a_transform;
    int32_t epilogue;
    float alpha, beta;
    float a_scale;
    float ep_inv_tau, ep_ce_scale;
    int32_t force_tile;   /* 0 = heuristic, 64 or 128 = force that square tile (tests/tuning) */
    int32_t force_splitk; /* 0 = heuristic */
} cgn_gemm_desc;

int cgn_gemm_f32(void* stream, const cgn_gemm_desc* desc /* host */);

/* ------------------------------------------------------------------------------------------------------------
 * Temperature softmax + NLL  (model/cgn.py:1086-1092: logits/tau -> F.log_softmax -> F.nll_loss, mean over ALL rows)
 *   lse[m]  = logsumexp_n(logits[m,:] * inv_tau)
 *   nll[m]  = lse[m] - logits[m,target[m]] * inv_tau
 *   loss[0] = mean_m nll[m]            (deterministic two-stage sum)
 * logits is [M, V] with row stride ld.
 * ---------------------------------------------------------------------------------------------------------- */
int cgn_softmax_nll_fwd_f32(void* stream, const float* logits, int64_t ld, const int64_t* target, int32_t M, int32_t V,
                            float inv_tau, float* lse, float* nll, float* loss);

#ifdef __cplusplus
}
#endif
#endif /* CGN_HIP_H */
