// One beam-search step for S sentences at once: Beam.advance of the reference (utils/beam_search.py:68-99) plus the
// back-pointer bookkeeping of model/cgn.py:813-838, on the device.
//
// Rows are sentence-major: row r = s*K + k holds beam k of sentence s. Per sentence the candidates are
//   beam_lk[k, v] = (logits[r, v] - lse[r]) + scores[r]      (log_softmax + accumulated score, :77)
// or, on the first step, beam 0 only (:79). The K best of the flattened [K*V] list are taken in descending order
// (:83, torch.topk sorted) under a strict order: (value descending, flat index ascending), so ties resolve to the
// smaller flat index. Two phases: every ROW first yields its own K best words (they contain the row's share of the
// sentence's K best), S*K workgroups; then one workgroup per sentence merges the K x K candidates. Each arg-max round
// admits only candidates strictly after the previous pick, so no "taken" list is needed. (One workgroup per sentence
// over all K*V candidates took 620 us per step at S = 64: 72 % of the sampler.)
#include "common.h"

namespace {

constexpr int BEAM_NT = 256;
constexpr int BEAM_MAXK = 64;

// strict order used everywhere: value descending, then index ascending
__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

__device__ __forceinline__ void block_argmax(float& bv, int& bi, float* red_v, int* red_i) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(bv, o, 64);
        const int i2 = __shfl_xor(bi, o, 64);
        if (better(v2, i2, bv, bi)) {
            bv = v2;
            bi = i2;
        }
    }
    __syncthreads();  // the previous round's readers are done with red_*
    if (lane == 0) {
        red_v[wave] = bv;
        red_i[wave] = bi;
    }
    __syncthreads();
    bv = red_v[0];
    bi = red_i[0];
#pragma unroll
    for (int w = 1; w < BEAM_NT / 64; ++w) {
        if (better(red_v[w], red_i[w], bv, bi)) {
            bv = red_v[w];
            bi = red_i[w];
        }
    }
}

// Phase 1, one workgroup per row r = s*K + k: the K largest logits of the row (value descending, word index ascending),
// by K rounds of a block-wide arg-max that admits only candidates strictly after the previous pick. Within a row the
// log-softmax offset and the beam score are constants, so raw logits order the row. The row (80 KB at V = 20k) stays
// in L2 between rounds.
__global__ __launch_bounds__(BEAM_NT) void beam_row_topk_kernel(const float* __restrict__ logits, long ld, int K, int V,
                                                                const int32_t* __restrict__ done,
                                                                float* __restrict__ cand_val,
                                                                int32_t* __restrict__ cand_idx) {
    __shared__ float red_v[BEAM_NT / 64];
    __shared__ int red_i[BEAM_NT / 64];
    const int r = blockIdx.x;
    if (done[r / K]) return;  // frozen sentence
    const float* x = logits + (long)r * ld;
    float prev_v = INFINITY;
    int prev_i = -1;
    for (int round = 0; round < K; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int v0 = threadIdx.x; v0 < V; v0 += BEAM_NT * 8) {  // 8 loads in flight
            float xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[u] = x[min(v0 + u * BEAM_NT, V - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int v = v0 + u * BEAM_NT;
                const float val = xv[u];
                const bool after = (val < prev_v) || (val == prev_v && v > prev_i);
                if (v < V && after && better(val, v, bv, bi)) {
                    bv = val;
                    bi = v;
                }
            }
        }
        block_argmax(bv, bi, red_v, red_i);
        prev_v = bv;
        prev_i = bi;
        if (threadIdx.x == 0) {
            cand_val[(long)r * K + round] = bv;       // -inf / 0x7fffffff when the row has fewer than K words
            cand_idx[(long)r * K + round] = bi;
        }
    }
}

// Phase 2, one workgroup per sentence: the K best of the (K rows) x (K candidates) list under
// (logit - lse[row]) + score[row], flat index k*V + v breaking ties, then the bookkeeping of Beam.advance.
__global__ __launch_bounds__(BEAM_NT) void beam_merge_kernel(const float* __restrict__ cand_val,
                                                             const int32_t* __restrict__ cand_idx,
                                                             const float* __restrict__ lse, int K, int V, int first,
                                                             int eos, int pad, float* __restrict__ scores,
                                                             int32_t* __restrict__ done, int32_t* __restrict__ length,
                                                             int64_t* __restrict__ tokens, int32_t* __restrict__ origin,
                                                             int64_t* __restrict__ src_rows, int64_t* __restrict__ hist_tok,
                                                             int32_t* __restrict__ hist_org) {
    __shared__ float red_v[BEAM_NT / 64];
    __shared__ int red_i[BEAM_NT / 64];
    __shared__ float pick_v[BEAM_MAXK];
    __shared__ int pick_i[BEAM_MAXK];
    const int s = blockIdx.x;
    const int r0 = s * K;
    if (done[s]) {  // a finished beam is frozen (cgn.py:815-816): identity back-pointers, pad tokens, same scores
        for (int k = threadIdx.x; k < K; k += BEAM_NT) {
            tokens[r0 + k] = pad;
            origin[r0 + k] = k;
            src_rows[r0 + k] = 2 * (long)(r0 + k) + 1;
            hist_tok[r0 + k] = pad;
            hist_org[r0 + k] = k;
        }
        return;
    }
    const int nk = first ? 1 : K;       // first step: beam 0 only (beam_search.py:79)
    const int ncand = nk * K;
    float prev_v = INFINITY;
    int prev_i = -1;
    for (int round = 0; round < K; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int c = threadIdx.x; c < ncand; c += BEAM_NT) {
            const int k = c / K;
            const int v = cand_idx[(long)r0 * K + c];
            if (v == 0x7fffffff) continue;
            const float val = (cand_val[(long)r0 * K + c] - lse[r0 + k]) + scores[r0 + k];
            const int e = k * V + v;
            const bool after = (val < prev_v) || (val == prev_v && e > prev_i);
            if (after && better(val, e, bv, bi)) {
                bv = val;
                bi = e;
            }
        }
        block_argmax(bv, bi, red_v, red_i);
        prev_v = bv;
        prev_i = bi;
        if (threadIdx.x == 0) {
            pick_v[round] = bv;
            pick_i[round] = bi;
        }
    }
    __syncthreads();  // every thread has read the old scores; picks are visible
    for (int k = threadIdx.x; k < K; k += BEAM_NT) {
        int fi = pick_i[k];
        float fv = pick_v[k];
        if (fi == 0x7fffffff) {  // fewer than K candidates (K > V on the first step): repeat the best pick at -inf
            fi = pick_i[0];
            fv = -INFINITY;
        }
        const int pk = fi / V;                 // beam_search.py:88 (integer division)
        const int tok = fi - pk * V;           // :92
        scores[r0 + k] = fv;                   // :84
        tokens[r0 + k] = tok;
        origin[r0 + k] = pk;
        src_rows[r0 + k] = 2 * (long)(r0 + pk) + 1;
        hist_tok[r0 + k] = tok;
        hist_org[r0 + k] = pk;
        if (k == 0) {
            length[s] += 1;
            if (tok == eos) done[s] = 1;       // :96-97: top-of-beam is EOS
        }
    }
}

}  // namespace

extern "C" int cgn_beam_topk_f32(void* stream, const float* logits, int64_t ld, const float* lse, int32_t S, int32_t K,
                                 int32_t V, int32_t first, int32_t eos, int32_t pad, float* scores, int32_t* done,
                                 int32_t* length, int64_t* tokens, int32_t* origin, int64_t* src_rows, int64_t* hist_tok,
                                 int32_t* hist_org, float* cand_val, int32_t* cand_idx) {
    CGN_CHECK_ARG(S >= 0 && K >= 1 && K <= BEAM_MAXK && V >= 1 && ld >= V);
    CGN_CHECK_ARG((long)K * V < (1L << 31) && eos >= 0 && eos < V && pad >= 0 && pad < V);
    if (S == 0) return CGN_OK;
    CGN_CHECK_ARG(logits && lse && scores && done && length && tokens && origin && src_rows && hist_tok && hist_org);
    CGN_CHECK_ARG(cand_val && cand_idx);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(beam_row_topk_kernel, dim3(S * K), dim3(BEAM_NT), 0, st, logits, (long)ld, K, V, done, cand_val,
                       cand_idx);
    CGN_LAUNCH_CHECK();
    hipLaunchKernelGGL(beam_merge_kernel, dim3(S), dim3(BEAM_NT), 0, st, cand_val, cand_idx, lse, K, V, first, eos, pad,
                       scores, done, length, tokens, origin, src_rows, hist_tok, hist_org);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}
