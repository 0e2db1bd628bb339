// gemm_bf16.hip — MODE 1 instantiations (bf16 storage, bf16-input MFMA with fp32 accumulation).
#include "gemm_core.h"
void xva_gemm_launch_bf16(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st) {
    xva_gemm_impl::launch_mode<1>(p, bn, nblocks, st);
}
