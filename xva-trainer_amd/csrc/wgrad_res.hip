// wgrad_res.hip — host side of the resident-operand weight-gradient kernel (wgrad_res.h): recognising the problem, the plan, the launch.
#include "wgrad_res.h"

namespace xva_glds { __global__ void xva_gemm_splitk_reduce_kernel(xva_gemm_params p); }
int xva_gemm_launch_splitk_reduce(const xva_gemm_params& p, hipStream_t st);

// xva_gemm_set_wgrad(0): weight gradients stay on the general TN kernel (A/B switch, tests)
static int g_wgrad_mode = 1;
extern "C" int xva_gemm_set_wgrad(int mode) { int old = g_wgrad_mode; g_wgrad_mode = mode; return old; }
// tuning overrides (tools/wgrad_bench.py): rows per chunk, DMA instructions per wave per chunk (2 / 4 / 6), workgroups in flight (0 = the plan's own)
static int g_tune_r = 0, g_tune_niw = 0, g_tune_wgs = 0, g_tune_ablate = 0;
extern "C" void xva_gemm_wgrad_tune(int r, int niw, int wgs) { g_tune_r = r; g_tune_niw = niw; g_tune_wgs = wgs & 0xffff; g_tune_ablate = wgs >> 16; }

// The TN problem
//      C[M = CO][N = k * CI] += alpha * A[K][CO]^T * B[K][n -> column segments of CI, stepping d * rowpitch]
// with K blocks (items) or one merged block is a convolution weight gradient (hg_conv.h: hg_conv_bwd_weight, non-swapped form;
// a_rowpitch = channel count of X, ldb = stride * a_rowpitch).  Returns 1 and fills the plan when the resident kernel takes it.
int xva_gemm_wgrad_res_plan(const xva_gemm_params& p, xva_wgrad::Plan* w, int* mi_out, int* nbw_out, int* niw_out) {
    using namespace xva_wgrad;
    if (g_wgrad_mode == 0) return 0;
    if (p.layout != XVA_GEMM_TN || p.compute != 1 || p.a_dtype != XVA_BF16 || p.b_dtype != XVA_BF16 || p.c_dtype != XVA_F32) return 0;
    if (!p.accumulate || p.accumulate == 2 || p.act != XVA_ACT_NONE || p.G || p.R || p.bias || p.mask_mode != XVA_MASK_NONE || p.c_trans || p.C2) return 0;
    if (p.a_lrelu || p.b_lrelu || p.a_seglen != 0 || p.drop_p > 0.f) return 0;
    if (p.batch > 1 || !p.sk_ws || ((uintptr_t)p.sk_ws % 16) != 0) return 0;
    const int CI = p.seglen, CO = p.M, N = p.N;
    if (!(CI == 8 || CI == 16 || CI == 32 || CI == 64 || CI == 128) || N % CI != 0 || CO % 16 != 0 || CO < 16) return 0;
    const int k = N / CI;
    const int64_t rp = p.a_rowpitch > 0 ? p.a_rowpitch : p.ldb;
    if (rp < CI || rp % 8 != 0 || p.ldb % rp != 0) return 0;
    const int s = (int)(p.ldb / rp);
    if (!(s == 1 || s == 2 || s == 4)) return 0;
    int d = 1;
    if (k > 1) {
        const int64_t step = p.segstride + CI;
        if (step <= 0 || step % rp != 0) return 0;
        d = (int)(step / rp);
        if (d > 16) return 0;
    }
    const int groups = p.batch2 > 1 ? p.batch2 : 1;
    if (p.lda % 8 != 0 || p.sA2 % 8 != 0 || p.sB2 % 8 != 0 || p.seg0 % 8 != 0 || p.kb_sA % 8 != 0 || p.kb_sB % 8 != 0) return 0;
    const int rows_blk = p.kb_len > 0 ? p.kb_len : p.K;
    const int nblk = p.kb_len > 0 ? p.K / p.kb_len : 1;
    if (p.kb_len > 0 && p.K % p.kb_len != 0) return 0;
    if ((int64_t)p.K < 2048 || rows_blk < 32) return 0;            // short reductions: the general kernel's tiles are fine
    const int MI = CO % 64 == 0 ? 4 : (CO % 32 == 0 ? 2 : 1);
    const int CO_W = MI * 16;
    const int nt_all = (N + 15) / 16;
    const int nbw_max = MI == 4 ? 6 : 10;                          // accumulators + double-buffered fragments inside 256 VGPRs
    const int col_groups = (nt_all + 8 * nbw_max - 1) / (8 * nbw_max);
    const int tiles_per_cg = (nt_all + col_groups - 1) / col_groups;
    const int per_wave = (tiles_per_cg + 7) / 8;
    const int NBW = per_wave <= 3 ? 3 : (per_wave <= 6 ? 6 : 10);
    const int tj = (tiles_per_cg * 16 + CI - 1) / CI + 1;          // taps a column group can span (upper bound)
    const int halo = (tj > k ? k - 1 : tj - 1) * d;
    // rows per chunk: the largest that fits a stage of 16 / 32 / 48 KiB,
    // not padding short blocks by more than ~15 %; ties go to the smaller stage (more workgroups per CU)
    const int rcap = (rows_blk + 63) & ~63;
    int bestR = 0, best_a = 0, best_b = 0, bestRGN = 0, bestNIW = 0;
    for (int NIW : {2, 4, 6}) {
        if (g_tune_niw && NIW != g_tune_niw) continue;
        for (int R : {256, 192, 128, 64}) {
            if (g_tune_r && R != g_tune_r) continue;
            if (R > rcap && R != 64) continue;
            const double waste = (double)((rows_blk + R - 1) / R) * R / rows_blk;
            if (waste > 1.15 && R != 64) continue;
            const int RGN = (R + halo / s + 1 + 15) & ~15;
            const int a_bytes = (R * CO_W * 2 + 1023) & ~1023;
            const int b_bytes = (s * RGN * CI * 2 + 1023) & ~1023;
            if (a_bytes + b_bytes > NIW * 8 * 1024) continue;
            if (R > bestR) { bestR = R; best_a = a_bytes; best_b = b_bytes; bestRGN = RGN; bestNIW = NIW; }
            break;
        }
    }
    if (bestR == 0) return 0;
    const int cpi = (rows_blk + bestR - 1) / bestR;
    const int64_t chunks_total = (int64_t)nblk * cpi;
    const int ntile = (CO / CO_W) * col_groups;
    // row-range splits: one workgroup (8 waves) per CU — two when the stages are 16 KiB and the accumulators few —, at least 2 chunks each,
    // inside the slab scratch
    const int per_cu = (bestNIW == 2 && MI * NBW <= 12 && NBW <= 6) ? 2 : 1;
    int64_t nsplit = (g_tune_wgs ? g_tune_wgs : 256 * per_cu) / ((int64_t)ntile * groups);
    if (nsplit < 1) nsplit = 1;
    if (nsplit > chunks_total / 2) nsplit = chunks_total / 2;
    const int64_t fit = p.sk_ws_bytes / ((int64_t)CO * N * 4 * groups);
    if (nsplit > fit) nsplit = fit;
    if (nsplit < 1) return 0;
    if (w) {
        w->CO = CO; w->CI = CI; w->k = k; w->s = s; w->d = d;
        w->co_tiles = CO / CO_W; w->col_groups = col_groups; w->tiles_per_cg = tiles_per_cg;
        w->R = bestR; w->RGN = bestRGN; w->cpi = cpi; w->nsplit = (int)nsplit;
        w->a_bytes = best_a; w->b_bytes = best_b; w->x_pitch = rp; w->ablate = g_tune_ablate;
    }
    if (mi_out) *mi_out = MI;
    if (nbw_out) *nbw_out = NBW;
    if (niw_out) *niw_out = bestNIW;
    return 1;
}

bool xva_gemm_wgrad_res_ok(const xva_gemm_params& p) { return xva_gemm_wgrad_res_plan(p, nullptr, nullptr, nullptr, nullptr) != 0; }

template <int MI, int NBW, int NIW, int D>
static int launch(const xva_gemm_params& p, const xva_wgrad::Plan& w, hipStream_t st) {
    auto kern = xva_wgrad::xva_wgrad_res_kernel<MI, NBW, NIW, D>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, NIW * 8 * 1024 * D) != hipSuccess) return -1;
        attr_set = true;
    }
    const int groups = p.batch2 > 1 ? p.batch2 : 1;
    const unsigned nblocks = (unsigned)(w.co_tiles * w.col_groups * w.nsplit * groups);
    hipLaunchKernelGGL(kern, dim3(nblocks), dim3(512), NIW * 8 * 1024 * D, st, p, w);
    return 0;
}
template <int MI, int NBW>
static int launch_stage(const xva_gemm_params& p, const xva_wgrad::Plan& w, int niw, hipStream_t st) {
    if (niw == 2) return launch<MI, NBW, 2, 4>(p, w, st);
    if (niw == 4) return launch<MI, NBW, 4, 4>(p, w, st);
    return launch<MI, NBW, 6, 3>(p, w, st);
}

// p: as handed to xva_gemm (splitk ignored).  Launches the kernel and the slab reduction.
int xva_gemm_launch_wgrad_res(const xva_gemm_params& pin, hipStream_t st, int* splits_out) {
    xva_wgrad::Plan w;
    int mi = 0, nbw = 0, niw = 0;
    if (!xva_gemm_wgrad_res_plan(pin, &w, &mi, &nbw, &niw)) return 1;
    xva_gemm_params p = pin;
    p.splitk = w.nsplit;
    p.batch = 1;
    if (p.batch2 < 1) p.batch2 = 1;
    if (splits_out) *splits_out = p.splitk;
    int rc;
    if (nbw == 3) rc = mi == 4 ? launch_stage<4, 3>(p, w, niw, st) : (mi == 2 ? launch_stage<2, 3>(p, w, niw, st) : launch_stage<1, 3>(p, w, niw, st));
    else if (nbw == 6) rc = mi == 4 ? launch_stage<4, 6>(p, w, niw, st) : (mi == 2 ? launch_stage<2, 6>(p, w, niw, st) : launch_stage<1, 6>(p, w, niw, st));
    else rc = mi == 2 ? launch_stage<2, 10>(p, w, niw, st) : launch_stage<1, 10>(p, w, niw, st);
    if (rc != 0) return -1;
    if (g_wgrad_mode == 2) return 0;                               // timing of the kernel alone (tools/wgrad_bench.py): the slabs are left unreduced
    return xva_gemm_launch_splitk_reduce(p, st);
}
