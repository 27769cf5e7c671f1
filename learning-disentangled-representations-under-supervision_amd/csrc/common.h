// Shared helpers for the CGN HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/cgn_hip.h"

#define CGN_WAVE 64

// Return-code helpers: the C-ABI never throws; every entry point returns an int.
#define CGN_CHECK_ARG(cond)                       \
    do {                                          \
        if (!(cond)) return CGN_ERR_INVALID_ARG;  \
    } while (0)

#define CGN_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) return CGN_ERR_HIP_BASE + (int)e__; \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// ceil(a / b) as an int, SATURATING at INT_MAX (block counts derived from it are clamped by their callers; tile counts
// cannot reach it: every GEMM dimension is checked < 2^30)
static inline int cgn_cdiv(long a, long b) {
    const long q = (a + b - 1) / b;
    return q > 2147483647L ? 2147483647 : (int)q;
}

__device__ __forceinline__ float cgn_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// 64-lane butterfly reductions.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x == NT (multiple of 64); result valid in every thread.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red /* NT/64 + 1 floats of LDS */) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NT / 64; ++i) t += red[i];
    return t;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NT / 64; ++i) t = fmaxf(t, red[i]);
    return t;
}
