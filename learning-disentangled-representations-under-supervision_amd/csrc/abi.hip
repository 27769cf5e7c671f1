// Library / ABI version (include/cgn_hip.h). A translation unit of its own: bumping it rebuilds nothing else.
//   7: cgn_gemm_bb, cgn_gemm_bb_wants, cgn_cast_bf16_f32 (bf16 x bf16 products of the bf16 mode)
//   5: persistent LSTM scratch is zeroed once by the caller (no per-launch memset, no per-launch timeout word)
#include "common.h"

extern "C" int cgn_abi_version(void) { return 7; }
