// One-wave kernel that measures the shader clock over a short busy loop: slot[0] = clock64 ticks, slot[1] = wall_clock64
// ticks (100 MHz). Built as a tiny shared library for tools/dwfc_clock_probe.py (NOT part of libcgn_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/experiments/clock_sample.hip -o tools/experiments/probe_clock_sample.so
#include <hip/hip_runtime.h>

__global__ void clock_sample_kernel(long* slot, int spins) {
    const long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (int i = 0; i < spins; ++i) x = x * 1.0001f + 0.5f;
    const long c1 = clock64(), w1 = wall_clock64();
    if (x == 123.f) slot[2] = 1;
    if (threadIdx.x == 0) {
        slot[0] = c1 - c0;
        slot[1] = w1 - w0;
    }
}

extern "C" int clock_sample(void* stream, long* slot) {
    hipLaunchKernelGGL(clock_sample_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, slot, 1500);
    return (int)hipGetLastError();
}
