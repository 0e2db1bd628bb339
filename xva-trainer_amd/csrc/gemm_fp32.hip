// gemm_fp32.hip — MODE 0 instantiations (fp32 storage, exact-fp32 MFMA): the parity path.
#include "gemm_core.h"
void xva_gemm_launch_fp32(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st) {
    xva_gemm_impl::launch_mode<0>(p, bn, nblocks, st);
}
