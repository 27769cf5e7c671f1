// One beam-search step for S sentences at once: Beam.advance of the reference (utils/beam_search.py:68-99) plus the
// back-pointer bookkeeping of model/cgn.py:813-838, on the device.
//
// Rows are sentence-major: row r = s*K + k holds beam k of sentence s. Per sentence the candidates are
//   beam_lk[k, v] = (logits[r, v] - lse[r]) + scores[r]      (log_softmax + accumulated score, :77)
// or, on the first step, beam 0 only (:79). The K best of the flattened [K*V] list are taken in descending order
// (:83, torch.topk sorted) by K rounds of a block-wide arg-max under a strict order: (value descending, flat index
// ascending), each round admitting only candidates strictly after the previous pick, so no "taken" list is needed and
// ties resolve to the smaller flat index. HBM/L2-bound: K passes over the sentence's K*V logits (800 KB at K=10,
// V=20k), all L2-resident after the first.
#include "common.h"

namespace {

constexpr int BEAM_NT = 1024;
constexpr int BEAM_MAXK = 64;

__global__ __launch_bounds__(BEAM_NT) void beam_topk_kernel(const float* __restrict__ logits, long ld,
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
    __shared__ float s_score[BEAM_MAXK], s_lse[BEAM_MAXK];
    const int s = blockIdx.x;
    const int r0 = s * K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
    for (int k = threadIdx.x; k < K; k += BEAM_NT) {
        s_score[k] = scores[r0 + k];
        s_lse[k] = lse[r0 + k];
    }
    __syncthreads();
    const int nk = first ? 1 : K;
    const long total = (long)nk * V;
    float prev_v = INFINITY;
    int prev_i = -1;
    for (int round = 0; round < K; ++round) {
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (long e = threadIdx.x; e < total; e += BEAM_NT) {
            const int k = (int)(e / V), v = (int)(e - (long)k * V);
            const float val = (logits[(long)(r0 + k) * ld + v] - s_lse[k]) + s_score[k];
            const bool after = (val < prev_v) || (val == prev_v && (int)e > prev_i);
            if (after && (val > bv || (val == bv && (int)e < bi))) {
                bv = val;
                bi = (int)e;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float v2 = __shfl_xor(bv, o, 64);
            const int i2 = __shfl_xor(bi, o, 64);
            if (v2 > bv || (v2 == bv && i2 < bi)) {
                bv = v2;
                bi = i2;
            }
        }
        if (lane == 0) {
            red_v[wave] = bv;
            red_i[wave] = bi;
        }
        __syncthreads();
        float fv = red_v[0];
        int fi = red_i[0];
#pragma unroll
        for (int w = 1; w < BEAM_NT / 64; ++w) {
            const float v2 = red_v[w];
            const int i2 = red_i[w];
            if (v2 > fv || (v2 == fv && i2 < fi)) {
                fv = v2;
                fi = i2;
            }
        }
        prev_v = fv;
        prev_i = fi;
        if (threadIdx.x == 0) {
            pick_v[round] = fv;
            pick_i[round] = fi;
        }
        __syncthreads();
    }
    for (int k = threadIdx.x; k < K; k += BEAM_NT) {
        int fi = pick_i[k];
        float fv = pick_v[k];
        if (fi == 0x7fffffff) {  // fewer than K candidates (K > V on the first step): repeat the last valid pick
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
                                 int32_t* hist_org) {
    CGN_CHECK_ARG(S >= 0 && K >= 1 && K <= BEAM_MAXK && V >= 1 && ld >= V);
    CGN_CHECK_ARG((long)K * V < (1L << 31) && eos >= 0 && eos < V && pad >= 0 && pad < V);
    if (S == 0) return CGN_OK;
    CGN_CHECK_ARG(logits && lse && scores && done && length && tokens && origin && src_rows && hist_tok && hist_org);
    hipLaunchKernelGGL(beam_topk_kernel, dim3(S), dim3(BEAM_NT), 0, (hipStream_t)stream, logits, (long)ld, lse, K, V,
                       first, eos, pad, scores, done, length, tokens, origin, src_rows, hist_tok, hist_org);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}
