// X6 variants of the MFMA GEMM (cgn_gemm_desc.compute == CGN_COMPUTE_F32X6): fp32 operands in HBM, split exactly into
// three bf16 terms as they are staged into LDS (gemm_core.h split3_pk), six partial products per block on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation = an fp32-accurate product at 2.67x less matrix-pipe time than
// v_mfma_f32_32x32x2_f32. Bias, epilogues, split-K slabs and C are the fp32 kernels'. Entry points live in gemm.hip.
#include "gemm_core.h"

namespace cgn_gemm {
bool launch_gemm_x6(const GemmParams& p, int bt, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    return bt == 128 ? launch_x6_t<128>(p, a_kc, b_kc, grid, s) : launch_x6_t<64>(p, a_kc, b_kc, grid, s);
}
bool launch_group_x6(const GemmGroup& g, bool a_kc, bool b_kc, int fk, unsigned blocks, hipStream_t s) {
    return launch_group_x6_t(g, a_kc, b_kc, fk, blocks, s);
}
}  // namespace cgn_gemm
