// Temperature softmax + NLL over [M, V] logits (model/cgn.py:1086-1092 of the reference).
// HBM-bound: one streaming read of the logits (16 B/lane), online max/sum per thread, block reduce per row.
#include "common.h"

namespace {

constexpr int SM_NT = 256;

__device__ __forceinline__ void online_update(float& m, float& s, float x) {
    if (x > m) {
        s = s * expf(m - x) + 1.f;
        m = x;
    } else {
        s += expf(x - m);
    }
}

typedef unsigned u32x4_sm __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void quad_update(float& m, float& s, float q0, float q1, float q2, float q3) {
    // one max per quad keeps the expf count at 4 (+1 on a new maximum)
    const float qm = fmaxf(fmaxf(q0, q1), fmaxf(q2, q3));
    if (qm > m) {
        s *= expf(m - qm);
        m = qm;
    }
    s += expf(q0 - m) + expf(q1 - m) + expf(q2 - m) + expf(q3 - m);
}

// the bf16-stored logits: v_exp_f32 (2^-21 relative) is far below the storage rounding; with expf's ~12 instructions
// per element the one-read kernel was VALU-bound there (1 GB of bf16 logits at configs[3] shapes in 316 us = 3.3 TB/s)
__device__ __forceinline__ void quad_update_fast(float& m, float& s, float q0, float q1, float q2, float q3) {
    const float qm = fmaxf(fmaxf(q0, q1), fmaxf(q2, q3));
    if (qm > m) {
        s *= __expf(m - qm);
        m = qm;
    }
    s += (__expf(q0 - m) + __expf(q1 - m)) + (__expf(q2 - m) + __expf(q3 - m));
}

__device__ __forceinline__ float bf16_at(const void* base, long i) {
    return __builtin_bit_cast(float, (unsigned)reinterpret_cast<const unsigned short*>(base)[i] << 16);
}

// IN16: the logits are stored as bf16 (the optional bf16 mode's storage format, cgn_gemm_desc.storage); ld in elements
template <bool IN16>
__global__ __launch_bounds__(SM_NT) void softmax_nll_fwd_kernel(const void* __restrict__ logits, long ld,
                                                                const int64_t* __restrict__ target, int V,
                                                                float inv_tau_imm, const float* __restrict__ dyn,
                                                                float* __restrict__ lse, float* __restrict__ nll) {
    const float inv_tau = dyn ? dyn[0] : inv_tau_imm;
    __shared__ float red_m[SM_NT / 64];
    __shared__ float red_s[SM_NT / 64];
    const int row = blockIdx.x;
    float m = -INFINITY, s = 0.f;
    const float* x = IN16 ? nullptr : reinterpret_cast<const float*>(logits) + (long)row * ld;
    const unsigned short* xh = IN16 ? reinterpret_cast<const unsigned short*>(logits) + (long)row * ld : nullptr;
    if constexpr (IN16) {
        if ((uintptr_t)xh % 16 == 0) {
            const int nv = V / 8;
            const u32x4_sm* xv = reinterpret_cast<const u32x4_sm*>(xh);
            auto take = [&](const u32x4_sm v) {
                float q[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    q[2 * j] = __builtin_bit_cast(float, v[j] << 16) * inv_tau;
                    q[2 * j + 1] = __builtin_bit_cast(float, v[j] & 0xffff0000u) * inv_tau;
                }
                quad_update_fast(m, s, q[0], q[1], q[2], q[3]);
                quad_update_fast(m, s, q[4], q[5], q[6], q[7]);
            };
            int i = threadIdx.x;
            for (; i + SM_NT < nv; i += 2 * SM_NT) {  // two loads in flight per thread
                const u32x4_sm v0 = xv[i], v1 = xv[i + SM_NT];
                take(v0);
                take(v1);
            }
            if (i < nv) take(xv[i]);
            for (int i = nv * 8 + threadIdx.x; i < V; i += SM_NT) online_update(m, s, bf16_at(xh, i) * inv_tau);
        } else {
            for (int i = threadIdx.x; i < V; i += SM_NT) online_update(m, s, bf16_at(xh, i) * inv_tau);
        }
    } else {
        const bool vec = ((uintptr_t)x % 16 == 0);
        if (vec) {
            const int nv = V / 4;
            const f32x4* xv = reinterpret_cast<const f32x4*>(x);
            for (int i = threadIdx.x; i < nv; i += SM_NT) {
                const f32x4 v = xv[i];
                quad_update(m, s, v[0] * inv_tau, v[1] * inv_tau, v[2] * inv_tau, v[3] * inv_tau);
            }
            for (int i = nv * 4 + threadIdx.x; i < V; i += SM_NT) online_update(m, s, x[i] * inv_tau);
        } else {
            for (int i = threadIdx.x; i < V; i += SM_NT) online_update(m, s, x[i] * inv_tau);
        }
    }
    // wave reduce of (m, s) pairs
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64);
        const float s2 = __shfl_xor(s, o, 64);
        const float mn = fmaxf(m, m2);
        const float e1 = (m == -INFINITY) ? 0.f : expf(m - mn);
        const float e2 = (m2 == -INFINITY) ? 0.f : expf(m2 - mn);
        s = s * e1 + s2 * e2;
        m = mn;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red_m[w] = m;
        red_s[w] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = red_m[0];
#pragma unroll
        for (int i = 1; i < SM_NT / 64; ++i) mm = fmaxf(mm, red_m[i]);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < SM_NT / 64; ++i) ss += (red_m[i] == -INFINITY) ? 0.f : red_s[i] * expf(red_m[i] - mm);
        const float l = mm + logf(ss);
        lse[row] = l;
        const int64_t tg = target[row];
        // an out-of-range target yields NaN instead of an out-of-bounds read
        nll[row] = (tg >= 0 && tg < V) ? l - (IN16 ? bf16_at(xh, tg) : x[tg]) * inv_tau : NAN;
    }
}

// Deterministic mean of n floats: single block, fixed order.
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float scale,
                                                   float* __restrict__ out) {
    __shared__ float red[8];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += v[i];
    s = block_sum<256>(s, red);
    if (threadIdx.x == 0) out[0] = s * scale;
}

}  // namespace

// out[i, :] = logits[rows[i], :] - lse[rows[i]]: the log_softmax rows a caller picks AFTER the fact (the sampler's
// learning=True walk-back, utils/beam_search.py:140-152); one block per picked row
__global__ void __launch_bounds__(SM_NT) log_softmax_rows_kernel(const float* __restrict__ logits, long ld,
                                                                  const float* __restrict__ lse,
                                                                  const long* __restrict__ rows, long n_src, int V,
                                                                  float* __restrict__ out, long ldo) {
    const long r = rows[blockIdx.x];
    if (r < 0 || r >= n_src) return;  // host validated; never index outside the source
    const float l = lse[r];
    const float* src = logits + r * ld;
    float* dst = out + (long)blockIdx.x * ldo;
    for (int v = threadIdx.x; v < V; v += SM_NT) dst[v] = src[v] - l;
}

static int softmax_nll_fwd(void* stream, const void* logits, bool in16, int64_t ld, const int64_t* target, int32_t M,
                           int32_t V, float inv_tau, const float* dyn_inv_tau, float* lse, float* nll, float* loss) {
    CGN_CHECK_ARG(M >= 0 && V > 0 && ld >= V);
    if (M == 0) return CGN_OK;
    CGN_CHECK_ARG(logits && target && lse && nll && loss);
    hipStream_t s = (hipStream_t)stream;
    if (in16)
        hipLaunchKernelGGL(softmax_nll_fwd_kernel<true>, dim3(M), dim3(SM_NT), 0, s, logits, (long)ld, target, V, inv_tau,
                           dyn_inv_tau, lse, nll);
    else
        hipLaunchKernelGGL(softmax_nll_fwd_kernel<false>, dim3(M), dim3(SM_NT), 0, s, logits, (long)ld, target, V, inv_tau,
                           dyn_inv_tau, lse, nll);
    CGN_LAUNCH_CHECK();
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, nll, M, 1.0f / (float)M, loss);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}

extern "C" int cgn_softmax_nll_fwd_f32(void* stream, const float* logits, int64_t ld, const int64_t* target, int32_t M,
                                       int32_t V, float inv_tau, const float* dyn_inv_tau, float* lse, float* nll,
                                       float* loss) {
    return softmax_nll_fwd(stream, logits, false, ld, target, M, V, inv_tau, dyn_inv_tau, lse, nll, loss);
}

extern "C" int cgn_softmax_nll_fwd_bf16(void* stream, const void* logits, int64_t ld, const int64_t* target, int32_t M,
                                        int32_t V, float inv_tau, const float* dyn_inv_tau, float* lse, float* nll,
                                        float* loss) {
    CGN_CHECK_ARG((uintptr_t)logits % 2 == 0);
    return softmax_nll_fwd(stream, logits, true, ld, target, M, V, inv_tau, dyn_inv_tau, lse, nll, loss);
}

extern "C" int cgn_log_softmax_rows_f32(void* stream, const float* logits, int64_t ld, const float* lse,
                                        const int64_t* rows, int32_t n_rows, int64_t n_src, int32_t V, float* out,
                                        int64_t ldo) {
    CGN_CHECK_ARG(n_rows >= 0 && V > 0 && ld >= V && ldo >= V && n_src >= 0);
    if (n_rows == 0) return CGN_OK;
    CGN_CHECK_ARG(logits && lse && rows && out);
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3(n_rows), dim3(SM_NT), 0, (hipStream_t)stream, logits, (long)ld, lse,
                       (const long*)rows, (long)n_src, V, out, (long)ldo);
    CGN_LAUNCH_CHECK();
    return CGN_OK;
}
