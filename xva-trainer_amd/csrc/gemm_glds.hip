// gemm_glds.hip — instantiations of the direct-to-LDS GEMM main loop (gemm_glds.h) and its eligibility test.
#include "gemm_glds.h"

// Can this problem take the direct-to-LDS path?  (bf16-stored operands, bf16 MFMA, segment / K-block lengths that keep a
// 64-deep K tile inside one segment, 8-element granularity of every index-contiguous dimension.)
bool xva_gemm_glds_eligible(const xva_gemm_params& p) {
    if (p.compute != 1 || p.a_dtype != XVA_BF16 || p.b_dtype != XVA_BF16) return false;
    if (p.a_lrelu || p.b_lrelu) return false;
    if (p.K % 8 != 0 || p.K < 8) return false;
    if (p.layout == XVA_GEMM_TN) {
        if (p.M % 8 != 0 || p.N % 8 != 0 || p.M < 8 || p.N < 8) return false;
        if (p.kb_len > 0 && p.kb_len % xva_glds::GK != 0) return false;
    } else {
        if (p.a_seglen > 0 && p.a_segadj != 0 && p.a_seglen % xva_glds::GK != 0) return false;
        if (p.layout == XVA_GEMM_NN) {
            if (p.N % 8 != 0 || p.N < 8) return false;
            if (p.seglen > 0 && p.seglen % xva_glds::GK != 0) return false;
        }
    }
    return true;
}

static int launch_tiles(const xva_gemm_params& p, int tile, int vec, hipStream_t st);

// tile: 0 = 128 x 128 (4 waves), 1 = 256 x 256 (8 waves)
int xva_gemm_launch_glds(const xva_gemm_params& pin, int tile, hipStream_t st) {
    using namespace xva_glds;
    xva_gemm_params p = pin;
    // split-K through slabs needs N % 4 == 0 and enough scratch; otherwise fall back to fp32 atomics
    const int64_t need = (int64_t)p.splitk * p.batch * p.batch2 * (int64_t)p.M * p.N * 4;
    if (p.splitk <= 1 || !p.sk_ws || p.sk_ws_bytes < need || p.N % 4 != 0 || ((uintptr_t)p.sk_ws % 16) != 0) p.sk_ws = nullptr;
    auto al = [](const void* q, int b) { return ((uintptr_t)q % b) == 0; };
    int vec = (p.N % 4 == 0) && (p.ldc % 4 == 0) && (p.sC % 4 == 0) && (p.sC2 % 4 == 0) && al(p.C, p.c_dtype == XVA_BF16 ? 8 : 16);
    if (p.R) vec = vec && (p.ldr % 4 == 0) && (p.sR % 4 == 0) && (p.sR2 % 4 == 0) && al(p.R, p.r_dtype == XVA_BF16 ? 8 : 16);
    if (p.G) vec = vec && (p.ldg % 4 == 0) && (p.sG % 4 == 0) && (p.sG2 % 4 == 0) && al(p.G, p.g_dtype == XVA_BF16 ? 8 : 16);
    int rc = launch_tiles(p, tile, vec, st);
    if (rc == 0 && p.sk_ws) {
        const int64_t quads = (int64_t)p.M * p.N / 4;
        int gx = (int)((quads + 255) / 256); if (gx > 2048) gx = 2048; if (gx < 1) gx = 1;
        hipLaunchKernelGGL(xva_gemm_splitk_reduce_kernel, dim3(gx, p.batch * p.batch2), dim3(256), 0, st, p);
    }
    return rc;
}

static int launch_tiles(const xva_gemm_params& p, int tile, int vec, hipStream_t st) {
    using namespace xva_glds;
    if (tile == 1) {
        switch (p.layout) {
            case XVA_GEMM_NT: return launch_tile<XVA_GEMM_NT, 256, 256, 128, 64>(p, vec, st);
            case XVA_GEMM_NN: return launch_tile<XVA_GEMM_NN, 256, 256, 128, 64>(p, vec, st);
            default: return launch_tile<XVA_GEMM_TN, 256, 256, 128, 64>(p, vec, st);
        }
    }
    switch (p.layout) {
        case XVA_GEMM_NT: return launch_tile<XVA_GEMM_NT, 128, 128, 64, 64>(p, vec, st);
        case XVA_GEMM_NN: return launch_tile<XVA_GEMM_NN, 128, 128, 64, 64>(p, vec, st);
        default: return launch_tile<XVA_GEMM_TN, 128, 128, 64, 64>(p, vec, st);
    }
}
