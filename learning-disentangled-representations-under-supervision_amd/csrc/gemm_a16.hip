// bf16 STORAGE variants of the MFMA GEMM (cgn_gemm_desc.storage & CGN_STORE_A_BF16): the A operand is a k-contiguous bf16
// matrix in HBM (the logits-sized tensors of the generator's output layer: logits, dlogits and its transposed copy), read
// with 16-byte loads straight into the bf16 LDS image; B is fp32 in memory and rounded on the way into LDS as in
// gemm_bf16.hip. Entry points live in gemm.hip.
#include "gemm_core.h"

namespace cgn_gemm {
void launch_gemm_a16(const GemmParams& p, int bt, bool b_kc, dim3 grid, hipStream_t s) {
    if (bt == 128) launch_a16_t<128>(p, b_kc, grid, s);
    else launch_a16_t<64>(p, b_kc, grid, s);
}
}  // namespace cgn_gemm
