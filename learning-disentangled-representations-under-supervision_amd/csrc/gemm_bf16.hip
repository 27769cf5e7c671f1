// bf16-operand variants of the MFMA GEMM (cgn_gemm_desc.compute == CGN_COMPUTE_BF16): operands are rounded to bf16 when
// they are staged into LDS and multiplied on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; everything outside the
// product (operands in HBM, bias, epilogues, split-K slabs, C) stays fp32. Entry points live in gemm.hip.
#include "gemm_core.h"

namespace cgn_gemm {
void launch_gemm_bf16ops(const GemmParams& p, int bt, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    launch_gemm<true>(p, bt, a_kc, b_kc, grid, s);
}
void launch_group_bf16ops(const GemmGroup& g, int vec, bool a_kc, bool b_kc, unsigned blocks, hipStream_t s) {
    launch_group<true>(g, vec, a_kc, b_kc, blocks, s);
}
}  // namespace cgn_gemm
