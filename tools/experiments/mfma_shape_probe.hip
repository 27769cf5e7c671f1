// Probe: sustained FLOP/s of a bare fp32 MFMA loop on random data, 32x32x2 against 16x16x4 (MI355X_MICROARCH.md, DVFS
// item 7 reports a clock difference between the bf16 shapes). Same FLOPs per wave in both loops, operands in registers,
// waves per SIMD = argv[1] (1 or 2). Build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k32(const float* in, float* out, int iters) {
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[(e + i) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k16(const float* in, float* out, int iters) {
    float a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[threadIdx.x * 8 + i]; b[i] = in[threadIdx.x * 8 + 4 + i]; }
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
        // 32 MFMAs of 16x16x4 (2048 FLOP each... per wave 16*16*4*2) = the 16 MFMAs of 32x32x2 above (32*32*2*2 each)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(e + i) & 3], b[(e * 2 + (i >> 2)) & 3], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
    const int wps = argc > 1 ? atoi(argv[1]) : 1;
    const int iters = 20000;
    const int blocks = 256 * wps;  // 4 waves per block = 1 per SIMD
    float *in, *out;
    hipMalloc(&in, 256 * 8 * 4);
    hipMalloc(&out, blocks * 256 * 4);
    std::vector<float> h(256 * 8);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = (double)blocks * 4 * iters * 16 * (32.0 * 32 * 2 * 2);
    for (int rep = 0; rep < 3; ++rep) {
        for (int which = 0; which < 2; ++which) {
            for (int w = 0; w < 3; ++w) { if (which) k16<<<blocks, 256>>>(in, out, iters); else k32<<<blocks, 256>>>(in, out, iters); }
            hipEventRecord(e0);
            for (int w = 0; w < 5; ++w) { if (which) k16<<<blocks, 256>>>(in, out, iters); else k32<<<blocks, 256>>>(in, out, iters); }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s waves/SIMD %d: %.3f ms per launch, %.1f TFLOP/s\n", which ? "16x16x4" : "32x32x2", wps, ms / 5, flop / (ms / 5 * 1e-3) / 1e12);
        }
    }
    return 0;
}
