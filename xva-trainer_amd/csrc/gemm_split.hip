// gemm_split.hip — MODE 3 instantiations (fp32 storage split into bf16 hi + lo while staged, three bf16-input MFMAs per product).
#include "gemm_core.h"
void xva_gemm_launch_split(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st) {
    xva_gemm_impl::launch_mode<3>(p, bn, nblocks, st);
}
