// gemm_mixed.hip — MODE 2 instantiations (fp32 storage rounded to bf16 while staged, bf16-input MFMA).
#include "gemm_core.h"
void xva_gemm_launch_mixed(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st) {
    xva_gemm_impl::launch_mode<2>(p, bn, nblocks, st);
}
