// gemm.hip — host side of the MFMA implicit-convolution GEMM: argument validation, tile selection, dispatch.
// Kernels: gemm_core.h (instantiated in gemm_fp32.hip / gemm_bf16.hip / gemm_mixed.hip).
#include "xva_common.h"
#include "../../include/xva_gemm.h"

void xva_gemm_launch_fp32(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st);
void xva_gemm_launch_bf16(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st);
void xva_gemm_launch_mixed(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st);
void xva_gemm_launch_split(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st);
bool xva_gemm_glds_eligible(const xva_gemm_params& p);
int xva_gemm_launch_glds(const xva_gemm_params& p, int tile, hipStream_t st);
int xva_gemm_launch_glds_f16(const xva_gemm_params& p, int tile, hipStream_t st);
void xva_gemm_glds_tile_dims(int tile, int* bm, int* bn);
int xva_gemm_conv_res_plan(const xva_gemm_params& p, int* stride_out = nullptr, int64_t* rowpitch_out = nullptr);
int xva_gemm_launch_conv_res(const xva_gemm_params& p, int dstep, hipStream_t st);
int xva_gemm_launch_wgrad_res(const xva_gemm_params& p, hipStream_t st, int* splits_out);
bool xva_gemm_wgrad_res_ok(const xva_gemm_params& p);
bool xva_prof_is_on();
void xva_prof_begin(hipStream_t st, double flops, int variant);
void xva_prof_end(hipStream_t st);
void xva_prof_cancel();
void xva_prof_shape(int M, int N, int K, int batch, int splitk, int bn, double bytes);

// Main-loop selection: -1 automatic (default), 0 general kernel only, 1..4 force the direct-to-LDS tile
// 128x128 / 256x256 / 128x64 / 64x64 wherever eligible.  A diagnostics / test knob, not part of the numerical contract.
static int g_glds_mode = -1;
extern "C" int xva_gemm_set_mainloop(int mode) { int old = g_glds_mode; g_glds_mode = mode; return old; }
// Products of compute == 0 (fp32-stored operands): 0 (default) = the exact fp32 MFMA, 1 = each operand split into two bf16 (hi + lo) while staged and
// three bf16 MFMAs per product (gemm_core.h MODE 3; ~1e-5 relative per product instead of 6e-8).  env XVA_GEMM_FP32_PRODUCTS
static int g_fp32_products = [] { const char* e = getenv("XVA_GEMM_FP32_PRODUCTS"); return e ? atoi(e) : 0; }();
extern "C" int xva_gemm_set_fp32_products(int mode) { int old = g_fp32_products; g_fp32_products = mode; return old; }
extern "C" int xva_gemm_get_fp32_products(void) { return g_fp32_products; }

extern "C" int xva_gemm_takes_colsum(const xva_gemm_params* pp) {
    if (!pp) return 0;
    xva_gemm_params p = *pp;
    if (p.batch < 1) p.batch = 1;
    if (p.batch2 < 1) p.batch2 = 1;
    return (p.splitk == 0 && p.layout == XVA_GEMM_TN && p.seglen > 0 && p.sk_ws && p.M > 0 && p.N > 0 && p.K > 0 && xva_gemm_wgrad_res_ok(p)) ? 1 : 0;
}
extern "C" int xva_gemm(const xva_gemm_params* pp, void* stream) {
    XVA_CHECK_ARG(pp != nullptr, "xva_gemm: null params");
    xva_gemm_params p = *pp;
    XVA_CHECK_ARG(p.A && p.B && p.C, "xva_gemm: null operand");
    XVA_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0, "xva_gemm: negative dim");
    if (p.M == 0 || p.N == 0) return XVA_OK;
    if (p.batch < 1) p.batch = 1;
    if (p.batch2 < 1) p.batch2 = 1;
    const bool auto_sk = (p.splitk == 0);     // 0: choose the split count here (needs accumulate into fp32 C, linear epilogue)
    if (p.splitk < 1) p.splitk = 1;
    XVA_CHECK_ARG(p.layout >= 0 && p.layout <= 2, "xva_gemm: bad layout");
    XVA_CHECK_ARG(p.a_dtype == p.b_dtype, "xva_gemm: A and B must share a storage dtype");
    XVA_CHECK_ARG(p.a_dtype == XVA_F32 || p.a_dtype == XVA_BF16 || p.a_dtype == XVA_F16, "xva_gemm: bad operand dtype");
    {   // one 16-bit format per problem
        const int h = p.a_dtype != XVA_F32 ? p.a_dtype : XVA_BF16;
        auto ok16 = [&](int dt) { return dt == XVA_F32 || dt == h; };
        XVA_CHECK_ARG(ok16(p.c_dtype) && (!p.R || ok16(p.r_dtype)) && (!p.G || ok16(p.g_dtype)) && (p.a_dtype != XVA_F32 || p.c_dtype != XVA_F16),
                      "xva_gemm: the 16-bit tensors of one problem (A, B, C, R, G) must share a format (bf16 or fp16)");
        XVA_CHECK_ARG(p.a_dtype != XVA_F16 || (!p.planes && !p.c_plane && !p.colsum_out), "xva_gemm: fp16 operands take no split planes / colsum_out");
    }
    XVA_CHECK_ARG(p.compute >= 0 && p.compute <= 2, "xva_gemm: compute must be 0 (fp32), 1 (bf16) or 2 (split-bf16 products of fp32 operands)");
    XVA_CHECK_ARG(!((p.compute == 0 || p.compute == 2) && p.a_dtype != XVA_F32), "xva_gemm: the fp32 pipes (exact, split products) need fp32-stored operands");
    const int ve = p.a_dtype != XVA_F32 ? 8 : 4;
    XVA_CHECK_ARG(p.lda % ve == 0 && p.ldb % ve == 0, "xva_gemm: lda/ldb must be multiples of %d (lda=%ld ldb=%ld)", ve,
                  (long)p.lda, (long)p.ldb);
    XVA_CHECK_ARG(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, "xva_gemm: operands must be 16-byte aligned");
    XVA_CHECK_ARG(p.sA % ve == 0 && p.sB % ve == 0 && p.sA2 % ve == 0 && p.sB2 % ve == 0, "xva_gemm: batch strides must be multiples of %d", ve);
    XVA_CHECK_ARG(p.seg0 % ve == 0 && p.segstride % ve == 0 && p.a_segadj % ve == 0, "xva_gemm: segment offsets must be multiples of %d", ve);
    XVA_CHECK_ARG(p.a_seglen % ve == 0, "xva_gemm: a_seglen must be a multiple of %d", ve);
    XVA_CHECK_ARG(p.layout != XVA_GEMM_NN || p.seglen % 4 == 0, "xva_gemm: NN seglen must be a multiple of 4");
    XVA_CHECK_ARG(p.layout != XVA_GEMM_TN || p.seglen % ve == 0, "xva_gemm: TN seglen must be a multiple of %d", ve);
    XVA_CHECK_ARG(p.splitk == 1 || (p.accumulate && p.c_dtype == XVA_F32), "xva_gemm: splitk > 1 requires accumulate into fp32 C");
    XVA_CHECK_ARG(p.splitk == 1 || (p.act == XVA_ACT_NONE && !p.G && p.mask_mode == XVA_MASK_NONE),
                  "xva_gemm: splitk > 1 cannot carry a non-linear epilogue");
    if (p.mask_mul == 0) p.mask_mul = 1;
    if (p.mask_len == 0) p.mask_len = p.Tp - 2 * p.mask_pad;
    XVA_CHECK_ARG(p.mask_mode == XVA_MASK_NONE ||
                      (p.batch == 1 && p.mask_pad >= 0 && p.mask_len > 0 && p.mask_pad + p.mask_len <= p.Tp &&
                       p.mask_add >= 0 && (p.mask_mode == XVA_MASK_PAD || p.lens)),
                  "xva_gemm: bad row-mask arguments");
    XVA_CHECK_ARG(p.accumulate != 2 || p.c_dtype == XVA_F32, "xva_gemm: atomic accumulation needs an fp32 C");
    XVA_CHECK_ARG(!p.C2 || (p.splitk == 1 && !p.accumulate && !p.c_trans && ((uintptr_t)p.C2 % 16) == ((uintptr_t)p.C % 16)),
                  "xva_gemm: a second output needs splitk == 1, plain (non-accumulating, non-transposed) stores and C's alignment");
    XVA_CHECK_ARG(!p.F || (p.G && p.splitk == 1 && !auto_sk), "xva_gemm: the feature-matching term needs the gate tensor G and an unsplit product");
    XVA_CHECK_ARG(p.kb_len == 0 || (p.layout == XVA_GEMM_TN && p.kb_len > 0 && p.kb_sA % ve == 0 && p.kb_sB % ve == 0), "xva_gemm: bad K-block arguments");
    XVA_CHECK_ARG(!p.planes || (p.compute == 1 && p.a_dtype == XVA_BF16 && p.a_seglen == 0 && p.a_segadj == 0 && p.seglen == 0 && p.kb_len == 0 &&
                                p.a_plane % 8 == 0 && p.b_plane % 8 == 0 && !p.a_lrelu && !p.b_lrelu),
                  "xva_gemm: split-bf16 planes need bf16 storage, compute 1, plain operands (no tap segments / K blocks / operand activation) and 8-element plane offsets");
    XVA_CHECK_ARG(!p.c_plane || (p.c_dtype == XVA_BF16 && !p.accumulate && !p.C2 && !p.c_trans && p.c_plane % 8 == 0 && p.compute == 1 && p.a_dtype == XVA_BF16),
                  "xva_gemm: a split-bf16 output pair needs a bf16 C without accumulation / second output / transposed store (direct-to-LDS kernels)");
    if (p.K == 0) p.splitk = 1;
    XVA_CHECK_ARG(!p.colsum_out || (auto_sk && p.layout == XVA_GEMM_TN && p.seglen > 0 && p.sk_ws && xva_gemm_wgrad_res_ok(p)),
                  "xva_gemm: colsum_out is honoured by the resident-operand weight-gradient kernel only (xva_gemm_takes_colsum)");
    if (auto_sk && p.layout == XVA_GEMM_TN && p.seglen > 0 && p.sk_ws && p.a_dtype != XVA_F16) {   // convolution weight gradient: resident-operand kernel (wgrad_res.h)
        const bool prof = xva_prof_is_on();
        if (prof) xva_prof_begin((hipStream_t)stream, 2.0 * p.M * (double)p.N * p.K * p.batch * p.batch2, p.layout * 3 + 1);
        int splits = 1;
        const int rc = xva_gemm_launch_wgrad_res(p, (hipStream_t)stream, &splits);
        if (rc < 0) { xva_set_error("xva_gemm: cannot raise the dynamic LDS limit"); return XVA_ERR_HIP; }
        if (rc == 0) {
            if (prof) {
                const double nbz = (double)p.batch * p.batch2;
                xva_prof_shape(p.M, p.N, p.K, p.batch * p.batch2, splits, 800000 + p.seglen, ((double)p.K * (p.M + p.seglen) * 2.0 + (double)p.M * p.N * 8.0) * nbz);
                xva_prof_end((hipStream_t)stream);
            }
            XVA_LAUNCH_CHECK();
            return XVA_OK;
        }
        if (prof) xva_prof_cancel();
    }
    int nkt = xva_cdiv(p.K, 32);
    if (p.splitk > nkt && nkt > 0) p.splitk = nkt;
    int bn = p.N <= 32 ? 32 : (p.N <= 64 ? 64 : 128);
    if (!auto_sk) {   // general kernel: narrower column tiles when the grid would leave most of the 256 CUs idle (split-K fills it otherwise)
        const long nb = (long)p.batch * p.batch2 * p.splitk;
        while (bn > 32 && (long)xva_cdiv(p.N, bn) * xva_cdiv(p.M, 128) * nb < 256) bn >>= 1;
    }
    // direct-to-LDS main loop (gemm_glds.h) for bf16-stored operands; xva_gemm_set_mainloop(0) keeps everything on the general kernel,
    // =1 forces the 128x128 tile, =2 forces 256x256 (A/B switches for profiling)
    const int glds_env = g_glds_mode;
    int glds_tile = -1;
    const bool can_split = auto_sk && p.accumulate && p.c_dtype == XVA_F32 && p.act == XVA_ACT_NONE && !p.G && p.mask_mode == XVA_MASK_NONE;
    if (glds_env != 0 && p.K >= 16 && xva_gemm_glds_eligible(p)) {
        const long nb = (long)p.batch * p.batch2;
        nkt = p.kb_len > 0 ? (p.K / p.kb_len) * xva_cdiv(p.kb_len, 64) : xva_cdiv(p.K, 64);   // K blocks are tiled one by one
        if (p.planes) nkt *= 3;                                                               // split-bf16 pairs: three passes over the K tiles
        auto ntiles = [&](int t) { int bm, bnn; xva_gemm_glds_tile_dims(t, &bm, &bnn); return (long)xva_cdiv(p.N, bnn) * xva_cdiv(p.M, bm) * nb; };
        const long t256 = ntiles(1);
        const double eff256 = (double)p.M * p.N * nb / ((double)t256 * 65536.0);
        // 256x256 tiles (one workgroup per CU) when they fill the chip — by themselves or through split-K — without much padding;
        // narrow outputs take 64-column tiles; grids that would leave most CUs idle take 64x64 tiles
        const int min_kt = p.planes ? 24 : 8;          // K tiles per split at least (planes: 8 per pass — the slabs of a 144-way split were 84 MB of traffic for a 0.3 MB result)
        const long maxsk = can_split ? (nkt / min_kt > 1 ? nkt / min_kt : 1) : p.splitk;
        if (p.N <= 32) glds_tile = 4;
        else if (p.N <= 64) glds_tile = 2;
        else if (nkt < 4) glds_tile = ntiles(0) >= 1024 ? 0 : 3;     // short reductions are prologue / epilogue bound: many small workgroups
        else if (p.layout != XVA_GEMM_TN && p.N > 256 && p.N <= 384 && ntiles(5) >= 160 && !can_split) glds_tile = 5;   // 384 x 128 tiles: no padded columns
        else if (t256 * maxsk >= 192 && eff256 >= 0.7) {
            glds_tile = 1;
            // Round quantisation (round 6): a grid a few tiles over a whole number of rounds of the 256 CUs pays a whole extra round — HiFi-GAN's period-11
            // discriminator (16 896 rows x 1 024 columns = 264 tiles of 256 x 256) ran its 1 024-channel layers at 0.52 round efficiency, 0.265 / 0.320 ms per
            // launch against 0.14 / 0.17 for the other periods.  Cost model: rounds x tile area, the 384 x 128 tile's K loop priced 15 % dearer per flop
            // (more LDS bytes per MFMA); NT / NN without split-K only (the 384-wide image has no TN form).
            if (p.layout != XVA_GEMM_TN && !can_split && p.N % 128 == 0) {
                const long t384 = ntiles(5);
                const double c256 = (double)((t256 + 255) / 256) * 65536.0, c384 = (double)((t384 + 255) / 256) * 49152.0 * 1.15;
                if (c384 < c256) glds_tile = 5;
            }
        }
        else if (p.layout != XVA_GEMM_TN && p.N % 128 == 0 && !can_split && ntiles(5) >= 176 && ntiles(5) <= 256) glds_tile = 5;   // one round of 384x128 tiles beats 1.x rounds of 128x128
        else if (ntiles(0) * maxsk >= 256) glds_tile = 0;
        else glds_tile = 3;
        if (glds_env >= 1 && glds_env <= 5) glds_tile = glds_env - 1;
        if (glds_env == 8) glds_tile = 6;                                // forced 256x128 (K tile 32, two workgroups per CU)
        if (glds_env == 7 && p.layout != XVA_GEMM_TN) glds_tile = 5;
        if (p.planes && glds_tile == 6) glds_tile = 0;                   // the 256x128 test tile has no plane passes     // forced 384x128 (NT / NN)   // forced: 1 -> 128x128, 2 -> 256x256, 3 -> 128x64, 4 -> 64x64, 5 -> 128x32
        int bm; xva_gemm_glds_tile_dims(glds_tile, &bm, &bn);
        bn = bn * 1000 + bm;   // profile tag
        if (can_split) {   // 256x256 tiles: at most one round of 256 workgroups; smaller tiles (2-3 per CU): ~1.7 rounds; >= 8 K tiles per split
            const long tiles = ntiles(glds_tile);
            // small tiles are latency-bound per K tile (one DMA stage in flight): fill the chip with ~6 workgroups per CU
            long sk = glds_tile == 1 ? 256 / tiles : ((glds_tile == 0 ? 864 : 1536) + tiles / 2) / tiles;
            if (sk > nkt / min_kt) sk = nkt / min_kt;
            if (sk > 1024) sk = 1024;
            if (p.sk_ws && p.N % 4 == 0) {   // stay inside the caller's slab scratch (atomics are much slower)
                const long fit = (long)(p.sk_ws_bytes / ((int64_t)p.M * p.N * 4 * nb));
                if (fit >= 2 && sk > fit) sk = fit;
            }
            p.splitk = sk < 1 ? 1 : (int)sk;
        }
        // Products with a NON-linear epilogue (forward / backward-data: bias, activation, gate, masks, dropout, bf16 output) whose grid leaves most
        // of the chip idle while the reduction is long — FastPitch's encoder feed-forward products: 4 864 rows x 384 columns over K = 4 608 are 114
        // tiles of 128 x 128 with 72 K tiles each — split along K as well when the caller offers slab scratch and asks for it (splitk = 0): the
        // partial sums go through the slabs and xva_gemm_splitk_reduce applies the whole epilogue (the same epilogue4 the tiles use).
        if (auto_sk && !can_split && p.sk_ws && p.layout != XVA_GEMM_TN && p.N % 4 == 0 && !p.C2 && glds_env < 0 && nkt >= 32 &&
            ((uintptr_t)p.sk_ws % 16) == 0) {
            const long t0 = ntiles(0);
            static const long fs_num = 520L;
            long sk = t0 > 0 ? fs_num / t0 : 1;                    // ~one round of 128 x 128 tiles at two workgroups per CU (512 slots); swept 288 / 400 / 520 / 700 on the encoder products: 520 (4 splits) best
            if (sk > nkt / 12) sk = nkt / 12;
            const long fit = (long)(p.sk_ws_bytes / ((int64_t)p.M * p.N * 4 * nb));
            if (sk > fit) sk = fit;
            if (sk >= 2 && t0 <= 144) { glds_tile = 0; p.splitk = (int)sk; int bm0; xva_gemm_glds_tile_dims(0, &bm0, &bn); bn = bn * 1000 + bm0; }
        }
        if (p.splitk > nkt) p.splitk = nkt;
    } else if (can_split) {
        const long tiles = (long)xva_cdiv(p.N, bn) * xva_cdiv(p.M, 128) * p.batch * p.batch2;
        long sk = (768 + tiles - 1) / tiles;
        if (sk > nkt / 8) sk = nkt / 8;
        p.splitk = sk < 1 ? 1 : (int)sk;
    }
    XVA_CHECK_ARG(p.a_dtype != XVA_F16 || glds_tile >= 0, "xva_gemm: fp16 operands run on the direct-to-LDS kernels only (K >= 16, 8-element granularity; set XVA_GEMM_GLDS != 0)");
    const int res_dstep = (glds_tile >= 0 && glds_env != 6) ? xva_gemm_conv_res_plan(p) : 0;   // mode 6: resident-input kernel off
    if (res_dstep != 0) bn = 900000 + p.a_seglen;
    long nblocks = glds_tile >= 0 ? 1 : (long)xva_cdiv(p.N, bn) * xva_cdiv(p.M, 128) * p.batch * p.batch2 * p.splitk;
    XVA_CHECK_ARG(nblocks < (1L << 31), "xva_gemm: grid too large");
    hipStream_t st = (hipStream_t)stream;
    const int mode = p.compute == 0 ? (g_fp32_products == 1 ? 3 : 0) : (p.compute == 2 ? 3 : (p.a_dtype != XVA_F32 ? 1 : 2));
    const bool prof = xva_prof_is_on();
    if (prof) { xva_prof_begin(st, 2.0 * p.M * (double)p.N * p.K * p.batch * p.batch2, p.layout * 3 + (mode == 3 ? 0 : mode)); 
        // algorithmic bytes: each distinct element of A, B once (tap segments re-address the SAME rows/columns), C written (read too when
        // accumulating), residual and gate read
        const double es_ab = p.a_dtype != XVA_F32 ? 2.0 : 4.0, es_c = p.c_dtype != XVA_F32 ? 2.0 : 4.0, nbz = (double)p.batch * p.batch2;
        double ua, ub;
        if (p.layout == XVA_GEMM_TN) { ua = (double)p.K * (p.a_seglen > 0 ? p.a_seglen : p.M); ub = (double)p.K * (p.seglen > 0 ? p.seglen : p.N); }
        else { ua = (double)p.M * (p.a_seglen > 0 ? p.a_seglen : p.K); ub = (double)p.N * p.K; }
        double by = (ua + ub) * es_ab + (double)p.M * p.N * es_c * (p.accumulate ? 2.0 : 1.0);
        if (p.R) by += (double)p.M * p.N * (p.r_dtype != XVA_F32 ? 2.0 : 4.0);
        if (p.G) by += (double)p.M * p.N * (p.g_dtype != XVA_F32 ? 2.0 : 4.0);
        if (p.F) by += (double)p.M * p.N * (p.g_dtype != XVA_F32 ? 2.0 : 4.0);
        xva_prof_shape(p.M, p.N, p.K, p.batch * p.batch2, p.splitk, bn, by * nbz); }
    XVA_CHECK_ARG(!p.C2 || glds_tile >= 0, "xva_gemm: the second output is written by the direct-to-LDS kernels only (bf16 operands, K >= 64)");
    XVA_CHECK_ARG(!(p.planes || p.c_plane) || glds_tile >= 0, "xva_gemm: split-bf16 planes run on the direct-to-LDS kernels only (K >= 16, 8-element granularity)");
    if (res_dstep != 0) {   // conv over 32 / 64 / 128 channels (per group), stride 1 / 2 / 4: resident input tile
        if (xva_gemm_launch_conv_res(p, res_dstep, st) != 0) { xva_set_error("xva_gemm: cannot raise the dynamic LDS limit"); return XVA_ERR_HIP; }
    } else if (glds_tile >= 0) { if ((p.a_dtype == XVA_F16 ? xva_gemm_launch_glds_f16(p, glds_tile, st) : xva_gemm_launch_glds(p, glds_tile, st)) != 0) { xva_set_error("xva_gemm: cannot raise the dynamic LDS limit"); return XVA_ERR_HIP; } }
    else if (mode == 0) xva_gemm_launch_fp32(p, bn, (unsigned)nblocks, st);
    else if (mode == 3) xva_gemm_launch_split(p, bn, (unsigned)nblocks, st);
    else if (mode == 1) xva_gemm_launch_bf16(p, bn, (unsigned)nblocks, st);
    else xva_gemm_launch_mixed(p, bn, (unsigned)nblocks, st);
    if (prof) xva_prof_end(st);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
