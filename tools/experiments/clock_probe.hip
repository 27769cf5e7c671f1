// Does the shader clock drop for a full-chip fp32 MFMA kernel that follows a low-power stretch? (NOT part of libcgn_hip.so.)
// Every round: a "quiet" kernel (256 workgroups that mostly sleep, like the latency-bound LSTM recurrences) for
// quiet_us, then a full-chip MFMA kernel with a fixed instruction count. The MFMA kernel reads the shader-clock counter
// (clock64: ticks at sclk) and the constant-rate counter (wall_clock64: 100 MHz) around its loop: their ratio is the clock
// it ran at. The dW_fc product of the step runs at 265 us alone but at 335 us (x 1.264 = 2400 / 1900) in a varying share
// of its in-step launches.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/clock_probe.hip -o tools/experiments/probe_clock
//   tools/experiments/probe_clock [quiet_us [mfma_iters [rounds]]]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void quiet_kernel(long wall_ticks) {
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < wall_ticks) __builtin_amdgcn_s_sleep(32);
}

__global__ __launch_bounds__(256) void mfma_kernel(int iters, long* sclk, long* wall, float* sink) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
    const long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    const long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) sink[0] = s;
    if (threadIdx.x == 0) {
        sclk[blockIdx.x] = c1 - c0;
        wall[blockIdx.x] = w1 - w0;
    }
}

#define CK(x)                                                       \
    do {                                                            \
        hipError_t e_ = (x);                                        \
        if (e_ != hipSuccess) {                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
            exit(1);                                                \
        }                                                           \
    } while (0)

int main(int argc, char** argv) {
    const int quiet_us = argc > 1 ? atoi(argv[1]) : 500;
    const int iters = argc > 2 ? atoi(argv[2]) : 180;
    const int rounds = argc > 3 ? atoi(argv[3]) : 60;
    const int blocks = 512;  // 2 per CU, 4 waves each: like the 128-tile GEMM
    long *sclk, *wall;
    float* sink;
    CK(hipMalloc(&sclk, blocks * sizeof(long)));
    CK(hipMalloc(&wall, blocks * sizeof(long)));
    CK(hipMalloc(&sink, 4));
    std::vector<long> hs(blocks), hw(blocks);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("quiet %d us, then %d x 32 MFMA per wave on %d workgroups; per round: kernel us, median shader MHz of its blocks\n",
           quiet_us, iters, blocks);
    std::vector<double> mhz_all, us_all;
    for (int r = 0; r < rounds; ++r) {
        if (quiet_us > 0) hipLaunchKernelGGL(quiet_kernel, dim3(256), dim3(512), 0, 0, (long)quiet_us * 100);
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(mfma_kernel, dim3(blocks), dim3(256), 0, 0, iters, sclk, wall, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(hs.data(), sclk, blocks * sizeof(long), hipMemcpyDeviceToHost));
        CK(hipMemcpy(hw.data(), wall, blocks * sizeof(long), hipMemcpyDeviceToHost));
        std::vector<double> mhz(blocks);
        for (int i = 0; i < blocks; ++i) mhz[i] = hw[i] > 0 ? 100.0 * (double)hs[i] / (double)hw[i] : 0.0;
        std::sort(mhz.begin(), mhz.end());
        mhz_all.push_back(mhz[blocks / 2]);
        us_all.push_back(ms * 1e3);
    }
    for (int r = 0; r < rounds; ++r) printf("%s%.0f/%.0f", r ? " " : "", us_all[r], mhz_all[r]);
    printf("\n");
    std::vector<double> u = us_all;
    std::sort(u.begin(), u.end());
    printf("kernel us: min %.1f median %.1f max %.1f (max/min %.3f)\n", u.front(), u[u.size() / 2], u.back(), u.back() / u.front());
    return 0;
}
