// gemm_glds_f16.hip — the IEEE-half (v_mfma_f32_16x16x32_f16) instantiations of the direct-to-LDS GEMM kernels: gemm_glds.hip compiled a second
// time with the 16-bit format switched (a translation unit of its own so that the two flavours build in parallel).
#define XVA_GLDS_F16 1
#include "gemm_glds.hip"
