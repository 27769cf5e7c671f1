// Ablation probe of the GEMM main loop (a COPY of csrc/gemm_core.h with pieces of the K step compiled out):
//   ABLATE 0 full kernel; 1 no global loads in the loop; 2 + no LDS stores; 3 + no barrier; 4 + no LDS fragment reads
// (results are wrong for ABLATE > 0; only the time matters). NT 4096^3 (or argv M N K), tile = PTILE (64|128).
#ifndef ABLATE
#define ABLATE 0
#endif
#ifndef PTILE
#define PTILE 128
#endif
#ifndef PLB
#define PLB 2
#endif
#ifndef PBK64
#define PBK64 32
#endif
#ifndef PSTAGES
#define PSTAGES 2
#endif
#ifndef PBK128
#define PBK128 16
#endif
#include "gemm_core_probe.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace cgn_gemm;
int main(int argc, char** argv) {
    const int M = argc > 3 ? atoi(argv[1]) : 4096, N = argc > 3 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 4096;
    float *A, *B, *C;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&C, (size_t)M * N * 4);
    std::vector<float> h((size_t)(M > N ? M : N) * K);
    srand(1);
    for (auto& v : h) v = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    GemmParams p = {};
    p.A = A; p.B = B; p.C = C;
    p.ldc = N; p.ldw = N;
    cgn_axis am = {K, 0, 0}, ak = {1, 0, 0}, bk = {1, 0, 0}, bn = {K, 0, 0};
    p.am = make_axis(am); p.ak = make_axis(ak); p.bk = make_axis(bk); p.bn = make_axis(bn);
    p.M = M; p.N = N; p.K = K; p.k_per_split = (K + KROUND - 1) / KROUND * KROUND; p.splitk = 1;
    p.a_vec = 1; p.b_vec = 1; p.a_transform = CGN_ATRANS_NONE; p.epilogue = CGN_EPI_LINEAR;
    p.fastk = 1; p.a_bytes = (unsigned)((size_t)M * K * 4); p.b_bytes = (unsigned)((size_t)N * K * 4);
    p.tiles_m = (M + PTILE - 1) / PTILE; p.tiles_n = (N + PTILE - 1) / PTILE; p.order = 2; p.alpha = 1.f;
    dim3 grid(p.tiles_m * p.tiles_n, 1);
    {
        int nb = 0;
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)gemm_kernel<PTILE, true, true, 3, false, PSTAGES, (PTILE == 128 ? PBK128 : PBK64), false, false, 1>, NT, 0);
        hipFuncAttributes fa;
        hipFuncGetAttributes(&fa, (const void*)gemm_kernel<PTILE, true, true, 3, false, PSTAGES, (PTILE == 128 ? PBK128 : PBK64), false, false, 1>);
        printf("occupancy query: %d blocks/CU; regs %d, static LDS %zu, max threads %d\n", nb, fa.numRegs, fa.sharedSizeBytes, fa.maxThreadsPerBlock);
    }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((gemm_kernel<PTILE, true, true, 3, false, PSTAGES, (PTILE == 128 ? PBK128 : PBK64), false, false, 1>), grid, dim3(NT), 0, 0, p);
        hipEventRecord(e0);
        for (int w = 0; w < 10; ++w) hipLaunchKernelGGL((gemm_kernel<PTILE, true, true, 3, false, PSTAGES, (PTILE == 128 ? PBK128 : PBK64), false, false, 1>), grid, dim3(NT), 0, 0, p);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 10 < best) best = ms / 10;
    }
    printf("tile %d ablate %d  %dx%dx%d: %.1f us  %.1f TFLOP/s\n", PTILE, ABLATE, M, N, K, best * 1e3, 2.0 * M * N * K / (best * 1e-3) / 1e12);
    return hipGetLastError() != hipSuccess;
}
