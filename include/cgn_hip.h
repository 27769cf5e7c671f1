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
 * GEMM (fp32 in memory unless `storage` says otherwise, fp32 accumulate; fp32 or, optionally, bf16 MFMA operands):
 *   C = epilogue(alpha * A·B [+ beta*C] [+ bias]).
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

#define CGN_COMPUTE_F32 0
#define CGN_COMPUTE_BF16 1
#define CGN_COMPUTE_F32X6 2 /* fp32 result from the bf16 matrix pipe: see cgn_gemm_desc.compute */

/* cgn_gemm_desc.storage bits: bf16 STORAGE of the logits-sized tensors in the optional bf16 mode (compute must be
 * CGN_COMPUTE_BF16). Everything not named here stays fp32 in memory. */
#define CGN_STORE_A_BF16 1  /* A is bf16 [M][K], k-contiguous: a_k = {1,0,0}, a_m = {row stride in bf16 elements, 0, 0}
                             * with stride % 8 == 0, K % 8 == 0, A 16-byte aligned; B must allow 16-byte loads */
#define CGN_STORE_C_BF16 2  /* C is bf16 (ldc in bf16 elements); no split-K (tall-K products keep fp32 C), beta == 0 */
#define CGN_STORE_EP_BF16 4 /* CGN_EPI_SOFTMAX_BWD: ep_logits is bf16 (ep_ld in bf16 elements); may alias C */

#define CGN_EPI_LINEAR 0      /* C = alpha*acc + beta*C + bias0[n] + bias1[n] */
#define CGN_EPI_SOFTMAX_BWD 1 /* see cgn_gemm_desc.ep_* */
/* CGN_EPI_LINEAR with beta == 0 PLUS the statistics a temperature softmax over the rows of C needs: the vocabulary
 * projection fused with the first pass of the log-softmax / CE behind it (model/generator.py:61 ->
 * model/cgn.py:1086-1092), so that the [B*T, V] logits are not read back for it. For row m and 64-column chunk c
 * (columns [64c, 64c+64) below N), with s = ep_inv_tau (or dyn_scalars[0]):
 *   ep_stats[(m * ep_stats_ld + c) * 2 + 0] = max_n s*C[m,n],  [... + 1] = sum_n exp(s*C[m,n] - that max).
 * cgn_softmax_nll_from_stats_f32 combines the chunks into lse / NLL / mean NLL. Only products that run the 128-tile X6
 * kernel unsplit can do this (both operands k-contiguous with 16-byte loads, compute == CGN_COMPUTE_F32X6, M, N >= 256):
 * ask cgn_gemm_stats_chunks first; cgn_gemm_f32 returns CGN_ERR_INVALID_ARG for a descriptor it cannot serve. */
#define CGN_EPI_LINEAR_STATS 2

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
    float* workspace; /* split-K slabs, 16-byte aligned: splitk * M * ceil4(N) floats are used; may be NULL (no split-K) */
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
    /* Optional DEVICE scalars {inv_tau, ce_scale} (NULL = use the immediates above). When given, a_scale (exp
     * transform) and ep_inv_tau read [0] and ep_ce_scale reads [1] at run time, so a captured hipGraph of the step
     * can be replayed while the temperature schedule (utils/functional.py:35-42) advances. */
    const float* dyn_scalars;
    /* Operand type of the product (the optional bf16 mode): CGN_COMPUTE_F32 = exact fp32 (v_mfma_f32_32x32x2_f32);
     * CGN_COMPUTE_BF16 = A (after a_transform) and B are rounded to bf16 (nearest-even) as they are staged into LDS
     * and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation. Memory operands, bias, epilogue, split-K
     * slabs and C stay fp32 in both modes unless `storage` names a tensor.
     * CGN_COMPUTE_F32X6 = fp32 accuracy on the bf16 pipe: every operand element is split EXACTLY into three bf16 terms
     * (x = p0 + p1 + p2, round-to-nearest terms, error-free) as it is staged into LDS, and the product is formed as six
     * partial bf16 MFMA products a0b0 + a0b1 + a1b0 + a0b2 + a1b1 + a2b0 with fp32 accumulation; the three dropped
     * terms are below 2^-23 |a b| per product, the size of fp32's own rounding. 2.67x less matrix-pipe time than the
     * fp32 MFMA. Products whose layout has no such variant (scalar loads, two-level K axes other than the history
     * walk, grouped launches) run the CGN_COMPUTE_F32 kernel; operands with +-inf give NaN (inf - inf in the split). */
    int32_t compute;
    int32_t storage; /* CGN_STORE_* bits, 0 = every tensor fp32 */
    /* CGN_EPI_SOFTMAX_BWD with CGN_STORE_C_BF16 only, nullable: a transposed bf16 copy of C, c_t[n * ld_ct + m] — the
     * k-contiguous A operand of the weight-gradient product dW_fc = dlogitsᵀ · h (model/generator.py:22,61 backward). */
    void* c_t;
    int64_t ld_ct;
    /* CGN_EPI_LINEAR_STATS only: [M][ep_stats_ld][2] floats out, ep_stats_ld >= ceil(N / 64) */
    float* ep_stats;
    int64_t ep_stats_ld;
} cgn_gemm_desc;

int cgn_gemm_f32(void* stream, const cgn_gemm_desc* desc /* host */);
/* *chunks = ceil(N / 64) when cgn_gemm_f32 would serve this descriptor with epilogue CGN_EPI_LINEAR_STATS (the
 * descriptor's epilogue field may still say CGN_EPI_LINEAR; ep_stats is not looked at), 0 when it would not. */
int cgn_gemm_stats_chunks(const cgn_gemm_desc* desc /* host */, int32_t* chunks /* host, out */);
/* n <= 8 independent products in ONE launch (64x64 tiles, linear epilogue, no A transform; all members must have the
 * same operand orientations and compute type). For the chains of small same-shaped convolution products: the six
 * char-TDNN widths (selfModules/tdnn.py:41-44, a Python loop over F.conv1d) and the text-CNN's filter widths
 * (model/discriminator_sentiment.py:64, a list comprehension over nn.Conv2d). A group with few tiles is split along K
 * inside the launch (slabs in the FIRST member's workspace, one grouped reduce launch, fixed summation order);
 * without a workspace it runs unsplit. Same results as n cgn_gemm_f32 calls up to fp32 summation order. */
int cgn_gemm_group_f32(void* stream, const cgn_gemm_desc* descs /* host array */, int32_t n);
/* Which kernel variant cgn_gemm_f32 would launch for `desc` (square tile 64|128, split-K factor, operand
 * orientations); used by bench.py to attribute per-launch timings to kernel symbols. Launches nothing. */
int cgn_gemm_plan(const cgn_gemm_desc* desc /* host */, int32_t* tile, int32_t* splitk, int32_t* a_kcontig,
                  int32_t* b_kcontig);
/* cgn_gemm_f32 with two hipEvent_t recorded on `stream` directly before and directly after the product's MFMA kernel
 * (the split-K reduce launch, if any, follows ev_stop): bench.py's live per-kernel timing, so that the event time is the
 * duration of the one kernel symbol rocprofv3 reports. vec_mode: nullable, receives the kernel's vector-load template
 * arguments (bit 0: A loads 16 bytes, bit 1: B, bits 2-3: fast K addressing mode, bit 4: A stored as bf16). Measurement aid,
 * no counterpart in the reference. */
int cgn_gemm_timed_f32(void* stream, const cgn_gemm_desc* desc /* host */, void* ev_start, void* ev_stop,
                       int32_t* vec_mode);

/* ------------------------------------------------------------------------------------------------------------
 * Temperature softmax + NLL  (model/cgn.py:1086-1092: logits/tau -> F.log_softmax -> F.nll_loss, mean over ALL rows)
 *   lse[m]  = logsumexp_n(logits[m,:] * inv_tau)
 *   nll[m]  = lse[m] - logits[m,target[m]] * inv_tau
 *   loss[0] = mean_m nll[m]            (deterministic two-stage sum)
 * logits is [M, V] with row stride ld.
 * ---------------------------------------------------------------------------------------------------------- */
int cgn_softmax_nll_fwd_f32(void* stream, const float* logits, int64_t ld, const int64_t* target, int32_t M, int32_t V,
                            float inv_tau, const float* dyn_inv_tau /* device, nullable: overrides inv_tau */,
                            float* lse, float* nll, float* loss);
/* The same over logits stored as bf16 (written by cgn_gemm_f32 with CGN_STORE_C_BF16; ld in bf16 elements); lse, nll and
 * the loss stay fp32. model/cgn.py:1086-1092 in the optional bf16 mode with bf16 storage. */
int cgn_softmax_nll_fwd_bf16(void* stream, const void* logits, int64_t ld, const int64_t* target, int32_t M, int32_t V,
                             float inv_tau, const float* dyn_inv_tau /* device, nullable: overrides inv_tau */,
                             float* lse, float* nll, float* loss);

/* The second half of the fused vocabulary projection + log-softmax + CE (model/generator.py:61 -> model/cgn.py:1086-1092):
 * combines the per-chunk statistics a CGN_EPI_LINEAR_STATS product left in stats [M][ld_stats][2] (chunks per row =
 * ceil(V / 64)) into lse / nll / loss exactly as cgn_softmax_nll_fwd_f32 defines them; of the logits only the M target
 * elements are read. inv_tau (dyn_inv_tau) must be the product's ep_inv_tau. scratch: (ceil(M / 4) + 1) 4-byte words,
 * the first zero on entry and left zero (the ticket of the fixed-order mean). */
int cgn_softmax_nll_from_stats_f32(void* stream, const float* stats, int64_t ld_stats, int32_t chunks,
                                   const float* logits, int64_t ld, const int64_t* target, int32_t M, int32_t V,
                                   float inv_tau, const float* dyn_inv_tau /* device, nullable */, float* lse, float* nll,
                                   float* loss, void* scratch /* device */, int64_t scratch_bytes);

/* The sampler's learning=True output (model/cgn.py:880,901-902 with utils/beam_search.py:140-152 get_hyp_probs): the
 * log_softmax rows (cgn.py:803) of the parent beams on the best hypothesis' back-pointer chain, picked after the search:
 *   out[i, :] = logits[rows[i], :] - lse[rows[i]],  rows[i] in [0, n_src). logits [n_src, V] row stride ld, out row
 * stride ldo. */
int cgn_log_softmax_rows_f32(void* stream, const float* logits, int64_t ld, const float* lse, const int64_t* rows,
                             int32_t n_rows, int64_t n_src, int32_t V, float* out, int64_t ldo);

/* ------------------------------------------------------------------------------------------------------------
 * LSTM recurrence (nn.LSTM at model/encoder.py:17-21,40 and model/generator.py:17-20,58). The input projections
 * x·W_ihᵀ + b_ih + b_hh are cgn_gemm_f32 calls; these entry points run the T dependent steps (one launch per step,
 * up to 4 independent sequences — e.g. the encoder's two directions — per launch). Layouts (batch-major fp32):
 *   xg [B,T,4H] gate order i,f,g,o · hbuf/cbuf [B,T+1,H]: forward sequences keep the initial state in slot 0 and
 *   the state after time t in slot t+1; reverse sequences keep the initial state in slot T and time t in slot t ·
 *   gates [B,T,4H] activated gates (saved for backward). The caller initialises the initial-state slots unless
 *   init_zero is set. H % 4 == 0.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cgn_lstm_fwd_seq {
    const float* xg;
    const float* xg_extra; /* nullable [B,4H]: added at every timestep (the tiled [z,c] input, generator.py:54-56) */
    const float* w_hh;     /* [4H,H] */
    float* hbuf;
    float* cbuf;
    float* gates;
    int32_t reverse;
    int32_t init_zero; /* 1: zero initial state (nn.LSTM called without (h0,c0)); the library zeroes the initial slots */
    /* nullable, persistent entry points only: scratch of 2*B16*H*4 bytes (B16 = B rounded up to 16; 16-byte aligned)
     * through which the workgroups hand h_t to each other, blocked by k group so that a wave's load covers whole cache
     * lines: fp32 in every fp32-accurate kernel (same values and summation order as without it: results do not change
     * by a bit; the recurrence's loads take half the L2 requests), bf16 in operand mode 2 at H > 512
     * (cgn_lstm_set_operand_mode). Contents need not survive the call; launches that run at the same time need
     * scratches of their own. */
    void* handoff;
} cgn_lstm_fwd_seq;

int cgn_lstm_seq_fwd_f32(void* stream, const cgn_lstm_fwd_seq* seqs /* host */, int32_t nseq, int32_t B, int32_t T,
                         int32_t H);

/* Operand mode of the recurrences, process-wide (read at launch time; a captured hipGraph keeps the kernels it was
 * captured with): 0 (default) = every recurrent product at fp32 accuracy (fp32 MFMA, or X6 where the three-term weight
 * slice fits the registers: H <= 512). 1, 2 = the optional bf16 mode (north_star: <= 1e-2) for the row-tiled
 * persistent kernels at H > 512:
 *   1: h.W_hh^T / dgates.W_hh from TWO-term bf16 splits of both operands (a0b0 + a0b1 + a1b0 on
 *      v_mfma_f32_16x16x32_bf16, fp32 accumulation, ~2^-16 relative per product; the two-term weight slice takes the
 *      same 128 registers as the fp32 one);
 *   2: plain bf16 operands (what every other product of the mode uses), and the workgroups hand h_t / dgates_t to each
 *      other as bf16 through the sequences' `handoff` scratch (a quarter of the bytes, an eighth of the L2 requests of
 *      the fp32 history buffers: the recurrence at H = 1024 is paced by those requests). Needs H % 256 == 0 and a
 *      handoff scratch on every sequence of the launch; launches without run as mode 1.
 * Cell state, gates, accumulation and everything written to the history buffers stay fp32 in all modes.
 * No counterpart in the reference (nn.LSTM, model/encoder.py:17-21, generator.py:17-20). */
int cgn_lstm_set_operand_mode(int32_t mode);

/* Persistent form of the two recurrences: ONE launch runs all T timesteps with the workgroup's W_hh slice, c_t / dc_t
 * held in registers; the workgroups sharing a batch tile hand h_t (dgates_t) over through write-through stores and a
 * per-group arrival counter (csrc/lstm.hip). Replaces the same reference call sites as cgn_lstm_seq_fwd/bwd_f32
 * (nn.LSTM at model/encoder.py:17-21,40 and model/generator.py:17-20,58), same buffers and results.
 * cgn_lstm_persistent_query: *sync_bytes = size of the caller-owned, 16-byte aligned device scratch `sync_ws` the
 * launch needs, or 0 when the shape is not eligible (H % 16, H <= 1024, nseq * ceil(B/(16*RT)) * H/16 workgroups <= CUs
 * of the current device for RT = 1, 2 or 4 row tiles per workgroup; H > 512 always takes the row-tiled kernels): use
 * the step kernels then, or one launch per sequence when a single sequence is eligible. The CALLER zeroes sync_ws ONCE
 * (hipMemset after allocation); the library never does: every launch leaves the arrival counters at zero again on its
 * way out (the last workgroup of a hand-off group to leave re-arms them), so no memset sits in front of a launch and
 * the entry points are plain kernel launches under hipGraph capture. The LAST whole 128-byte line of the buffer
 * actually passed (sync_ws_bytes) is a STICKY timeout flag: set by any launch whose bounded wait ran out (results
 * invalid from then on), never cleared by the library. While it is set every wait of every later launch on this
 * sync_ws gives up after at most 64 polls (launches drain at once, their results are invalid, their counters still end
 * at zero): fail-stop. The caller checks it at its host sync points and zeroes it to resume. (ABI 4 also had a
 * per-launch timeout word, zeroed by a memset node before each launch: gone in ABI 5.)
 * Launches sharing one sync_ws must not overlap. */
int cgn_lstm_persistent_query(int32_t nseq, int32_t B, int32_t T, int32_t H, int64_t* sync_bytes);
int cgn_lstm_seq_fwd_persistent_f32(void* stream, const cgn_lstm_fwd_seq* seqs /* host */, int32_t nseq, int32_t B,
                                    int32_t T, int32_t H, void* sync_ws, int64_t sync_ws_bytes);

/* cgn_lstm_seq_fwd_persistent_f32 for callers that cannot allocate. NO COUNTERPART IN THE REFERENCE: it exists for
 * callers that may not call hipMalloc - code under hipGraph capture, tight serving loops, harnesses that forbid it.
 * Same kernels, buffers and results as cgn_lstm_seq_fwd_persistent_f32, but the arrival counters / timeout word live in
 * a file-scope device array of the library (33,024 bytes per GPU, held until the process ends) instead of a caller-owned
 * sync_ws; the call allocates nothing (no hipMalloc / hipFree, no allocating library routine): it is ONE kernel launch
 * on `stream` (the array starts zeroed, every launch leaves its counters at zero). The FIRST call on a device resolves
 * the array's device address (hipGetSymbolAddress) and must therefore be made outside stream capture. Limit: nseq * ceil(B/16) <= 256 counter groups (and the shape must be eligible per
 * cgn_lstm_persistent_query), else CGN_ERR_INVALID_ARG. Every call on a GPU shares that GPU's array: calls that overlap
 * on one GPU (different streams) race - serialise them. status: nullable device word, set to 1 by a launch whose bounded
 * wait ran out (results invalid), never cleared by the library; NULL keeps the flag in the library's own array. */
int cgn_lstm_seq_fwd_persistent_noalloc_f32(void* stream, const cgn_lstm_fwd_seq* seqs /* host */, int32_t nseq,
                                            int32_t B, int32_t T, int32_t H, uint32_t* status);

/* BPTT. dgates [B,T,4H] receives d(pre-activation gates); dW_ih/dW_hh/db/dX follow as GEMMs / column sums.
 * dh_ext: nullable gradient wrt the emitted h, element (b,t,j) at dh_ext[b*dh_bs + t*dh_ts + j];
 * dh_seed: nullable [B,H] gradient wrt the final hidden state; dc_state [B,H]: in = gradient wrt the final cell
 * state (the encoder's context, encoder.py:43-46), out = gradient wrt the initial cell state. */
typedef struct cgn_lstm_bwd_seq {
    const float* w_hh_t; /* [H,4H] = W_hhᵀ (cgn_transpose_f32); the persistent form takes w_hh instead when given */
    const float* w_hh;   /* nullable [4H,H]: untransposed weight (persistent form only) */
    const float* cbuf;
    const float* gates;
    const float* dh_ext;
    const float* dh_seed;
    float* dgates;
    float* dc_state;
    int64_t dh_bs, dh_ts;
    int32_t reverse;
    void* handoff; /* nullable, persistent entry point only: scratch of 2*B16*4H*4 bytes for dgates_t, as cgn_lstm_fwd_seq.handoff */
} cgn_lstm_bwd_seq;

int cgn_lstm_seq_bwd_f32(void* stream, const cgn_lstm_bwd_seq* seqs /* host */, int32_t nseq, int32_t B, int32_t T,
                         int32_t H);
int cgn_lstm_seq_bwd_persistent_f32(void* stream, const cgn_lstm_bwd_seq* seqs /* host */, int32_t nseq, int32_t B,
                                    int32_t T, int32_t H, void* sync_ws, int64_t sync_ws_bytes);

/* Two stacked LSTM layers in ONE persistent launch each way (the generator's nn.LSTM(num_layers=2, batch_first),
 * model/generator.py:17-20,58, zero initial state as model/cgn.py:1082 calls it). The layers form a wavefront, layer 1
 * one timestep behind layer 0; layer 1's input projection h0·W_ih1ᵀ + b_ih1 + b_hh1 is folded into its recurrence
 * (K = 2H over [h0_t ; h1_{t-1}]), and in the backward pass the gradient wrt h0 (dg1·W_ih1) into the bottom layer's
 * (K = 8H over [dg1_t ; dg0_{t+1}]). Same buffers and layouts as cgn_lstm_seq_fwd/bwd_f32; layer 1 has no xg.
 * Gradients wrt the final states are zero, gradients wrt the initial states are not produced.
 * cgn_lstm2_persistent_query: *sync_bytes = scratch size (as cgn_lstm_persistent_query), 0 = shape not eligible
 * (H % 16, H <= 512, 2 * ceil(B/16) * H/16 workgroups <= CUs): run the layers one after the other then. */
typedef struct cgn_lstm2_fwd {
    const float* xg0;       /* [B,T,4H] layer-0 input projections incl. both biases */
    const float* xg0_extra; /* nullable [B,4H], added at every timestep */
    const float* w_hh0;     /* [4H,H] */
    const float* w_ih1;     /* [4H,H] */
    const float* w_hh1;     /* [4H,H] */
    const float* b_ih1;     /* [4H] */
    const float* b_hh1;     /* [4H] */
    float* hbuf0;
    float* cbuf0;
    float* gates0;
    float* hbuf1;
    float* cbuf1;
    float* gates1;
} cgn_lstm2_fwd;
typedef struct cgn_lstm2_bwd {
    const float* w_hh0; /* untransposed [4H,H] weights */
    const float* w_ih1;
    const float* w_hh1;
    const float* cbuf0;
    const float* gates0;
    const float* cbuf1;
    const float* gates1;
    const float* dh_ext1; /* nullable: gradient wrt the emitted h1, element (b,t,j) at dh_ext1[b*dh_bs + t*dh_ts + j] */
    float* dgates0;       /* [B,T,4H] out */
    float* dgates1;       /* [B,T,4H] out */
    int64_t dh_bs, dh_ts;
    /* nullable: scratch of 2*B*T*4H*4 bytes (16-byte aligned) for blocked copies of both layers' gate gradients, read by
     * the workgroups instead of dgates0/1 (whole cache lines per load; results do not change by a bit), as
     * cgn_lstm_fwd_seq.handoff */
    void* handoff;
} cgn_lstm2_bwd;
int cgn_lstm2_persistent_query(int32_t B, int32_t T, int32_t H, int64_t* sync_bytes);
int cgn_lstm2_fwd_persistent_f32(void* stream, const cgn_lstm2_fwd* d /* host */, int32_t B, int32_t T, int32_t H,
                                 void* sync_ws, int64_t sync_ws_bytes);
int cgn_lstm2_bwd_persistent_f32(void* stream, const cgn_lstm2_bwd* d /* host */, int32_t B, int32_t T, int32_t H,
                                 void* sync_ws, int64_t sync_ws_bytes);

/* ------------------------------------------------------------------------------------------------------------
 * One beam-search step for S sentences x K beams (rows r = s*K + k), replacing Beam.advance
 * (utils/beam_search.py:68-99) and the per-sentence state reordering of model/cgn.py:813-838 in the sleep-phase
 * sampler (sample_from_generator, cgn.py:747-921). Candidates (logits[r,v] - lse[r]) + scores[r]; first != 0: beam 0
 * only (beam_search.py:79). Per live sentence: scores[r] <- the K best, descending (ties: smaller flat index);
 * tokens[r] / origin[r] <- word and back-pointer (:88-92); src_rows[r] <- 2*(s*K + origin) + 1 (row of the parent's
 * new state in a [R, 2, H] history buffer viewed as [2R, H]); hist_tok/hist_org receive the same (this step's slice
 * of the [steps, R] history); length[s] += 1; done[s] <- 1 when the top beam emits eos (:96-97). Sentences with
 * done[s] != 0 are frozen: scores kept, tokens = pad, origin = identity. K <= 64, K*V < 2^31.
 * ---------------------------------------------------------------------------------------------------------- */
int cgn_beam_topk_f32(void* stream, const float* logits, int64_t ld, const float* lse, int32_t S, int32_t K, int32_t V,
                      int32_t first, int32_t eos, int32_t pad, float* scores, int32_t* done, int32_t* length,
                      int64_t* tokens, int32_t* origin, int64_t* src_rows, int64_t* hist_tok, int32_t* hist_org,
                      float* cand_val /* scratch [S*K*K] */, int32_t* cand_idx /* scratch [S*K*K] */);

/* ------------------------------------------------------------------------------------------------------------
 * Convolution + max-over-time pooling helpers, shared by the text-CNN discriminator
 * (model/discriminator_sentiment.py:56-78; act = 0 ReLU, weight layout 0 = [Co][K][E]) and the char-TDNN
 * (selfModules/tdnn.py:40-43; act = 1 tanh, weight layout 1 = [od][Cc][kW]). The convolution product itself is
 * cgn_gemm_f32 over overlapping windows.
 * ---------------------------------------------------------------------------------------------------------- */
#define CGN_ACT_RELU 0
#define CGN_ACT_TANH 1
typedef struct cgn_conv_widths {
    const float* W[8]; /* per width: weights */
    float* dW[8];      /* per width: weight gradient (cgn_conv_bwd_dw_f32) */
    float* db[8];      /* per width: bias gradient, nullable (TDNN has no bias) */
    int32_t K[8];      /* window length */
    int32_t Co[8];     /* maps of this width */
    int32_t col0[8];   /* first feature column of this width in the pooled feature row */
    int32_t n;         /* number of widths (<= 8) */
    int32_t F;         /* pooled feature row length (sum of Co) */
    int32_t act, layout;
} cgn_conv_widths;

/* conv [B*P, Co] (row (b,p), bias included) -> feat[b, col0+co] = act(max_p conv), arg = argmax position
 * (-1 where the ReLU gate is closed). */
int cgn_act_maxpool_f32(void* stream, const float* conv, int32_t B, int32_t P, int32_t Co, int32_t act, float* feat,
                        int64_t ldf, int32_t col0, int32_t* arg, int64_t lda);
/* The same for n <= 8 widths in one launch: width i has its own conv buffer conv[i] [B*P[i], Co[i]] and writes columns
 * col0[i] .. col0[i]+Co[i] of feat / arg (the torch.cat over widths of tdnn.py:46, discriminator_sentiment.py:70). */
int cgn_act_maxpool_group_f32(void* stream, int32_t n, const float* const* conv /* host array of device pointers */,
                              const int32_t* P /* host */, const int32_t* Co /* host */, const int32_t* col0 /* host */,
                              int32_t B, int32_t act, float* feat, int64_t ldf, int32_t* arg, int64_t lda);
/* dX[b, 0:L, 0:E] (sentence stride sent_stride) from dfeat [B,F] through the kept windows; feat needed for tanh. */
/* Dense route for the same dX (faster where every map's window would otherwise be walked on the VALU): (1) spread the
 * pooled gradient back over the convolution outputs, dconv[i] [B*(L-K_i+1), Co_i] = g at the kept position, 0 elsewhere
 * (dconv: host array of n 16-byte aligned device pointers); (2) dcols_i = dconv_i · W_i [Co_i, K_i*E] in window order (k, e) — a
 * cgn_gemm_group_f32 launch over the widths; (3) cgn_conv_col2im_f32 sums the overlapping windows:
 * dx[b,t,e] = sum_i sum_k dcols_i[(b, t-k), k*E + e]. Replaces the autograd backward of F.conv1d / nn.Conv2d + max
 * (selfModules/tdnn.py:42-43, model/discriminator_sentiment.py:64-66) wrt the input. */
int cgn_pool_bwd_scatter_f32(void* stream, const cgn_conv_widths* widths, const float* dfeat, const float* feat,
                             const int32_t* arg, int32_t B, int32_t L, float* const* dconv /* host array */);
int cgn_conv_col2im_f32(void* stream, int32_t n, const float* const* dcols /* host array */, const int32_t* K /* host */,
                        int32_t B, int32_t L, int32_t E, float* dx, int64_t sent_stride);
int cgn_conv_bwd_dx_f32(void* stream, const cgn_conv_widths* widths /* host */, const float* dfeat, const float* feat,
                        const int32_t* arg, int32_t B, int32_t L, int32_t E, float* dx, int64_t sent_stride);
/* dW/db (+= when beta = 1) from dfeat through the kept windows of x [B, L, E]. */
int cgn_conv_bwd_dw_f32(void* stream, const cgn_conv_widths* widths /* host */, const float* dfeat, const float* feat,
                        const int32_t* arg, const float* x, int64_t sent_stride, int32_t B, int32_t L, int32_t E,
                        float beta);
/* Embedding-table gradient (char_embed, selfModules/embedding.py:45 backward): dtable[v,:] = beta*dtable[v,:] +
 * sum_{pos: idx[pos]==v} dx[pos,:]; atomics-free, fixed order. */
int cgn_embedding_bwd_f32(void* stream, const int64_t* idx, const float* dx, int32_t n, int32_t V, int32_t E, float beta,
                          float* dtable, float* workspace /* >= 64*V*E floats */, int64_t workspace_bytes);

/* ------------------------------------------------------------------------------------------------------------
 * Streaming glue kernels.
 * ---------------------------------------------------------------------------------------------------------- */
/* dst[r,c] = scale*src[r,c] + beta*dst[r,c] (strided rows). */
int cgn_copy2d_f32(void* stream, const float* src, int64_t lds, float* dst, int64_t ldd, int32_t rows, int32_t cols,
                   float scale, float beta);
/* dst[b,t,0:H] = src[b,t,0:H] + beta*dst[b,t,0:H] with independent batch/time strides (LSTM history slots <->
 * batch-first tensors; s_ts = 0 broadcasts over t, the zc tiling of model/generator.py:54). */
int cgn_copy3d_f32(void* stream, const float* src, int64_t s_bs, int64_t s_ts, float* dst, int64_t d_bs, int64_t d_ts,
                  int32_t B, int32_t T, int32_t H, float beta);
/* dst[cols, rows] = src[rows, cols]ᵀ (dense). */
int cgn_transpose_f32(void* stream, const float* src, int32_t rows, int32_t cols, float* dst);
/* out[i, 0:E] = table[idx[i], :] * mask[i, :] (nn.Embedding lookups of the frozen word table,
 * selfModules/embedding.py:42,57; mask = explicit word-dropout mask, model/generator.py:52; nullable). */
int cgn_gather_rows_f32(void* stream, const float* table, int32_t V, int32_t E, const int64_t* idx, int32_t n,
                        const float* mask, float* out, int64_t ldo);
/* Highway gate (selfModules/highway.py:38-44). pre = [N, 3*Sp]: [nonlinear | linear | gate] pre-activation blocks of
 * S columns at block stride Sp >= S (Sp a multiple of 4 keeps every block 16-byte aligned for the GEMMs). */
int cgn_highway_gate_fwd_f32(void* stream, const float* pre, int32_t N, int32_t S, int32_t Sp, float* out, int64_t ldo);
int cgn_highway_gate_bwd_f32(void* stream, const float* pre, int32_t N, int32_t S, int32_t Sp, const float* dy,
                             int64_t ldy, float* dpre);
/* Reparameterisation + KLD (model/cgn.py:94-101,1076-1080). mulv [B,2Z] = [mu|logvar]; zc [B,Z+C] = [z|c]. */
int cgn_reparam_kld_fwd_f32(void* stream, const float* mulv, const float* eps, const float* c, int32_t B, int32_t Z,
                            int32_t C, float* zc, float* stdv, float* kld);
/* dmulv = d/d[mu|logvar] of (g·z + kld_weight*B*kld), g = dL/dz (nullable, row stride ldg), kld_weight = kld_coef/B. */
int cgn_reparam_kld_bwd_f32(void* stream, const float* mulv, const float* eps, const float* g, int64_t ldg,
                            float kld_weight, const float* dyn_kld_weight /* device, nullable: overrides kld_weight */,
                            int32_t B, int32_t Z, float* dmulv);
/* out[c] = beta*out[c] + sum_r x[r,c] (bias gradients); workspace >= 64*C floats. */
int cgn_colsum_f32(void* stream, const float* x, int64_t ld, int32_t R, int32_t C, float beta, float* out,
                   float* workspace, int64_t workspace_bytes);
/* The same over a bf16 matrix (the bf16 mode's dlogits, cgn_gemm_desc.storage): generator.fc bias gradient. */
int cgn_colsum_bf16(void* stream, const void* x, int64_t ld, int32_t R, int32_t C, float beta, float* out,
                    float* workspace, int64_t workspace_bytes);
/* out[b,c] = sum_t x[b,t,c]. */
int cgn_sum_time_f32(void* stream, const float* x, int32_t B, int32_t T, int32_t C, float* out);
/* out[r] = sum_c a[r,c]*b[r,c]. */
int cgn_rowdot_f32(void* stream, const float* a, int64_t lda, const float* b, int64_t ldb, int32_t R, int32_t C,
                   float* out);
/* torch.optim.Adam defaults (model/cgn.py:267,1046-1047) over a flat parameter arena; g is read as g*grad_scale
 * (1/world_size after a sum all-reduce). step >= 1 is the step count AFTER this update. */
int cgn_adam_f32(void* stream, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                 float beta2, float eps, int32_t step, const int32_t* dyn_step /* device, nullable: overrides step */,
                 float grad_scale);
/* counter[0] += 1 on the stream: advances a device-resident Adam step count inside a captured hipGraph. */
int cgn_incr_i32(void* stream, int32_t* counter);
/* ------------------------------------------------------------------------------------------------------------
 * Fused launch-bound chains of the encoder/generator step (csrc/fused.hip). Each replaces a chain of tiny dependent
 * kernels between two large products by ONE launch with one workgroup per sentence. `scratch`: caller-owned device
 * words, (B + 1) * 4 bytes, zeroed ONCE by the caller (a ticket the last workgroup re-arms + one float per sentence for
 * the fixed-order batch sums); launches sharing a scratch must not overlap.
 * ---------------------------------------------------------------------------------------------------------- */
/* The code head: context = cat(c_n fwd, c_n bwd) (model/encoder.py:40-46) -> mu | logvar (model/cgn.py:91-92, weights
 * stacked [2Z, 2He]) -> z = eps * exp(0.5 logvar) + mu, std (cgn.py:94-101) -> KLD (cgn.py:1076) -> [z, c] (cgn.py:1078-
 * 1080) -> zg = [z,c] . W_ih_l0[:, E:]^T, the tiled input's share of the generator's layer-0 gates (model/generator.py:
 * 54-56, once per sentence instead of per token). Limits: 2He <= 2048, Z <= 128, Z + C <= 140, He % 2 == 0, ldw2 % 4 ==
 * 0, ldzc % 4 == 0, W2 / Wzc 16-byte aligned and rows readable up to the next multiple of 4 columns. */
typedef struct cgn_code_head_fwd {
    const float* cf;   /* final cell state, forward sequence: element (b, j) at cf[b * ldc + j] */
    const float* cr;   /* final cell state, reverse sequence, same addressing */
    int64_t ldc;
    const float* W2;   /* [2Z, 2He], row stride ldw2 */
    int64_t ldw2;
    const float* b2;   /* [2Z] */
    const float* eps;  /* [B, Z] */
    const float* c;    /* [B, C] (nullable when C == 0) */
    const float* Wzc;  /* nullable: 4Hg rows of Z + C weights, row stride ldzc */
    int64_t ldzc;
    float* ctx;        /* [B, 2He] out */
    float* mulv;       /* [B, 2Z] out */
    float* zc;         /* [B, Z + C] out */
    float* stdv;       /* [B, Z] out */
    float* kld;        /* [1] out */
    float* zg;         /* [B, G4] out (with Wzc) */
    void* scratch;
    int64_t scratch_bytes;
    int32_t B, He, Z, C, G4;
} cgn_code_head_fwd;
int cgn_code_head_fwd_f32(void* stream, const cgn_code_head_fwd* d /* host */);

/* Its backward (cgn.py:1099, :1142-1148): dsum = sum_t dgates0[:, t, :]; d[z,c] = dsum . W_ih_l0[:, E:] (its first Z
 * columns are g = d(79 CE + recon)/dz); d[mu|logvar] through the reparametrisation plus the KLD term's own gradient
 * (weight kld_w = kld_coef / B, or *dyn_kld_w); d(context) = d[mu|logvar] . W2, written as the two final cell states'
 * gradients. Limits: G4 <= 4096, Z <= 128, Z + C <= 144. */
typedef struct cgn_code_head_bwd {
    const float* dg0;       /* [B, T, G4] */
    const float* Wzc;
    int64_t ldzc;
    const float* mulv;      /* [B, 2Z] */
    const float* eps;       /* [B, Z] */
    const float* dyn_kld_w; /* device, nullable: overrides kld_w */
    const float* W2;
    int64_t ldw2;
    float* dsum;            /* [B, G4] out */
    float* dzc;             /* [B, Z + C] out */
    float* dmulv;           /* [B, 2Z] out */
    float* dcf;             /* [B, He] out */
    float* dcr;             /* [B, He] out */
    float kld_w;
    int32_t B, T, He, Z, C, G4;
} cgn_code_head_bwd;
int cgn_code_head_bwd_f32(void* stream, const cgn_code_head_bwd* d /* host */);

/* The c-reconstruction head behind the frozen text-CNN's convolution products (cgn.py:1114-1118 with
 * discriminator_sentiment.py:66-77 in eval mode): ReLU + max over time with argmax, fc1, CE against c.long(), and the
 * way back: dlogits = w_ce / B * (softmax - onehot), through fc1 and the pooling into dconv[i] (same shape as conv[i]:
 * the gradient at each map's kept position, zeros elsewhere). conv[i] / dconv[i]: [B * P[i], Co[i]]; c: label b at
 * c[b * ldc]; feat / arg [B, F = sum Co] out; stats[0] = mean CE. F <= 1024. */
int cgn_textcnn_head_f32(void* stream, int32_t n, const float* const* conv /* host array */,
                         float* const* dconv /* host array */, const int32_t* P /* host */, const int32_t* Co /* host */,
                         int32_t B, const float* W, const float* bias, const float* c, int64_t ldc, float w_ce,
                         float* feat, int32_t* arg, float* stats, void* scratch, int64_t scratch_bytes);

/* cgn_conv_col2im_f32 for the soft embedding's gradient [B, T, E] (contiguous rows (b, t)): rows t < L receive the
 * summed windows (+ beta * dx), rows L <= t < T zero (+ beta * dx): the dropped last timestep of cgn.py:1108; and
 * rowdot[b*T + t] = sum_e dx[.., e] * other[(b*T + t) * ldo + e], the row dot product of the softmax Jacobian that the
 * CGN_EPI_SOFTMAX_BWD epilogue reads (cgn.py:1103-1106 backwards). */
int cgn_conv_col2im_rowdot_f32(void* stream, int32_t n, const float* const* dcols /* host array */,
                               const int32_t* K /* host */, int32_t B, int32_t L, int32_t T, int32_t E, float beta,
                               float* dx, const float* other, int64_t ldo, float* rowdot);

/* Measurement aid (no counterpart in the reference): *slot = the device's constant-rate wall clock (wall_clock64, 100 MHz)
 * at the moment `stream` reaches this point. Unlike HIP events it can be a node of a captured hipGraph, so it times the
 * phases of an UN-profiled replay (tools/step_stamps.py; hip/ops.py STAMPS). */
int cgn_stamp_u64(void* stream, uint64_t* slot);
/* 2-class head (discriminator_sentiment.py:76-77 + losses of cgn.py:167-170,306,318-334,1118):
 * logits = (feat*mask)·Wᵀ + b; stats = {mean CE(y), mean entropy}; p1 = softmax[:,1];
 * dlogits = d(w_ce*CE + w_ent*entropy)/dlogits. mask, y, p1, dlogits, stats nullable. */
int cgn_disc_head_f32(void* stream, const float* feat, int64_t ldf, const float* mask, const float* W, const float* bias,
                      const int64_t* y, int32_t B, int32_t F, float w_ce, float w_ent, float* logits, float* p1,
                      float* dlogits, float* stats);
int cgn_disc_head_bwd_f32(void* stream, const float* feat, int64_t ldf, const float* mask, const float* W,
                          const float* dlogits, int32_t B, int32_t F, float beta, float* dW, float* db, float* dfeat);
/* c = (u < p1) — t.bernoulli(softmax[:,1]) with the uniform draw given (model/cgn.py:170). */
int cgn_bernoulli_f32(void* stream, const float* p1, const float* u, int32_t n, float* c);
int cgn_fill_f32(void* stream, float* p, int64_t n, float v);
/* out[i] = (int64) x[i] — `c.long()` (model/cgn.py:1117). */
int cgn_float_to_index_f32(void* stream, const float* x, int32_t n, int64_t* out);
/* z-reconstruction loss (model/cgn.py:47-57,1134-1137): mulv1 = [mu|logvar] of the first encoder pass (constants),
 * mulv2 of the second pass; z_loss[0] = sum_b dist_b / B; dmulv2 = weight * d z_loss / d mulv2. */
int cgn_mahalanobis_f32(void* stream, const float* mulv1, const float* mulv2, int32_t B, int32_t Z, float weight,
                        float* z_loss, float* dmulv2);
/* out[i] = a[i]*b[i] (explicit dropout masks, model/generator.py:52). */
int cgn_mul_f32(void* stream, const float* a, const float* b, float* out, int64_t n);
/* out[0] = sum_i coef[i]*src[i][0], n <= 4 (loss totals of model/cgn.py:329,334,1151-1152); src, coef: host arrays. */
int cgn_lincomb_f32(void* stream, int32_t n, const float* const* src, const float* coef, float* out);

/* ------------------------------------------------------------------------------------------------------------
 * bf16 x bf16 products of the optional bf16 mode (BASELINE configs[3] is stated in bf16): the nn.Linear / nn.LSTM
 * input products of model/generator.py:17-22,58-61 and model/encoder.py:17-21,40 and their backward products, with
 * BOTH operands held as bf16 in memory. cgn_gemm_f32 with CGN_COMPUTE_BF16 rounds fp32 operands while it stages them
 * (every tile of an operand is re-read and re-rounded by every block that needs it, 4 bytes per element); here an
 * operand is rounded ONCE by cgn_cast_bf16_f32 (round to nearest even, the same rounding) and the product reads
 * 2 bytes per element with 16-byte loads: 256x256 block tiles, 8 waves of 128x64, v_mfma_f32_32x32x16_bf16, fp32
 * accumulation. Same result as the CGN_COMPUTE_BF16 product of the fp32 operands up to fp32 summation order.
 *   C[m,n] = alpha * sum_k A[m*lda + k] * B[n*ldb + k] + beta * C[m,n] + bias0[n] + bias1[n]
 * A [M][lda], B [N][ldb]: bf16, k-contiguous, 16-byte aligned, lda % 8 == ldb % 8 == 0, K % 8 == 0 (a caller with
 * another K pads with zeros: cgn_cast_bf16_f32 does), each operand below 2 GB. C fp32, or bf16 when c_bf16 != 0
 * (then beta == 0, ldc % 8 == 0, C 16-byte aligned, no split along K).
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct cgn_gemm_bb_desc {
    const void* A;
    const void* B;
    void* C;
    const float* bias0; /* nullable, length N */
    const float* bias1; /* nullable, length N */
    float* workspace;   /* split-K slabs, 16-byte aligned: splitk * M * ceil4(N) floats; may be NULL (no split) */
    int64_t workspace_bytes;
    int64_t lda, ldb, ldc; /* elements */
    int32_t M, N, K;
    float alpha, beta;
    int32_t c_bf16;
    int32_t force_tile;   /* 0 = heuristic; 256 = 256x256 block tile, 192 = 256x128, 128 = 128x128, 384 = 128x384 (fp32 C,
                           * N <= 384: one column tile; the heuristic takes it for CGN_ATRANS_EXP products that fit) */
    int32_t force_splitk; /* 0 = heuristic, else 1..16 */
    /* The two logits-sized products of model/cgn.py:1103-1108 with the logits HELD in bf16, as cgn_gemm_desc has them:
     * a_transform CGN_ATRANS_EXP: A (bf16 logits) enters the product as exp(A * a_scale - a_row_param[m]) - the soft
     *   embedding softmax(logits / tau) @ E with the probabilities formed as the tiles are staged;
     * epilogue CGN_EPI_SOFTMAX_BWD (c_bf16 only, no split along K): with p = exp(ep_logits[m,n] * ep_inv_tau - ep_lse[m]),
     *   C[m,n] = p * (acc - ep_rowdot[m]) * ep_inv_tau + ep_ce_scale * (p - (n == ep_target[m])); ep_logits is bf16
     *   [M][ep_ld] and may alias C; c_t: nullable bf16 [N][ld_ct >= M], the transposed copy of C.
     * dyn_scalars: nullable DEVICE {inv_tau, ce_scale}, overriding a_scale / ep_inv_tau and ep_ce_scale at run time. */
    int32_t a_transform;
    int32_t epilogue;
    float a_scale, ep_inv_tau, ep_ce_scale;
    const float* a_row_param;
    const void* ep_logits;
    const float* ep_lse;
    const float* ep_rowdot;
    const int64_t* ep_target;
    int64_t ep_ld;
    void* c_t;
    int64_t ld_ct;
    const float* dyn_scalars;
} cgn_gemm_bb_desc;
/* ev_start / ev_stop: nullable hipEvent_t recorded on `stream` directly around the product kernel (bench.py's per-kernel
 * timing, as cgn_gemm_timed_f32); tile / splitk: nullable, receive the plan that ran. */
int cgn_gemm_bb(void* stream, const cgn_gemm_bb_desc* desc /* host */, void* ev_start, void* ev_stop, int32_t* tile,
                int32_t* splitk);
/* 1 when cgn_gemm_bb takes the shape (M, N >= 128, K >= 64: below that the casts cost more than they save) */
int cgn_gemm_bb_wants(int32_t M, int32_t N, int32_t K);
/* dst = bf16(src), round to nearest even. src: fp32, n_rows rows addressed by the two-level axis `rows` (as cgn_gemm_desc
 * axes: plain matrices, [B,T+1,H] history buffers), n_cols columns of unit stride.
 *   transpose == 0: dst[r * ld_dst + c], columns n_cols .. ld_dst-1 of every row written as zeros
 *   transpose != 0: dst[c * ld_dst + r], columns n_rows .. ld_dst-1 of every row written as zeros
 * ld_dst % 8 == 0 and >= the row length, dst 16-byte aligned. */
int cgn_cast_bf16_f32(void* stream, const float* src, cgn_axis rows, int32_t n_rows, int32_t n_cols, void* dst,
                      int64_t ld_dst, int32_t transpose);

#ifdef __cplusplus
}
#endif
#endif /* CGN_HIP_H */
