// gemm_glds.hip — instantiations of the direct-to-LDS GEMM main loop (gemm_glds.h) and its eligibility test.
// Compiled twice: as is (bf16 kernels + everything flavour-independent) and through gemm_glds_f16.hip with XVA_GLDS_F16 = 1 (the IEEE-half kernels only).
#include "gemm_glds.h"
#ifndef XVA_GLDS_F16
#define XVA_GLDS_F16 0
#endif
static constexpr bool F16V = XVA_GLDS_F16 != 0;
int xva_gemm_vec_epilogue_ok(const xva_gemm_params& p);
int xva_gemm_launch_splitk_reduce(const xva_gemm_params& p, hipStream_t st);
int xva_gemm_conv_res_plan(const xva_gemm_params& p, int* stride_out, int64_t* rowpitch_out);
int xva_gemm_glds_kloop8();
int xva_gemm_glds_kloop384();

#if !XVA_GLDS_F16
namespace xva_glds {
// C (+)= epilogue(sum_s slab[s])  for one batch item per blockIdx.y.
// gridDim.z == 1: one pass with the full epilogue of the tile kernels (epilogue4).  gridDim.z > 1 (many slabs of a small output; host-checked: pure fp32
// accumulation, no bias / residual): each z sums its share of the slabs and adds alpha * partial atomically.
__global__ __launch_bounds__(256) void xva_gemm_splitk_reduce_kernel(xva_gemm_params p) {
    const int b2n = p.batch2 > 1 ? p.batch2 : 1;
    const int bz = blockIdx.y, z1 = bz / b2n, z2 = bz - z1 * b2n;
    const int64_t MN = (int64_t)p.M * p.N;
    const float* slab = reinterpret_cast<const float*>(p.sk_ws) + (int64_t)bz * p.splitk * MN;
    const int64_t coff = (int64_t)z1 * p.sC + (int64_t)z2 * p.sC2;
    const int64_t roff = (int64_t)z1 * p.sR + (int64_t)z2 * p.sR2;
    const int per = (p.splitk + gridDim.z - 1) / gridDim.z;
    const int k0 = blockIdx.z * per, k1 = min(p.splitk, k0 + per);
    if (k0 >= k1) return;
    xva_gemm_params q1 = p; q1.splitk = 1;          // the epilogue of an unsplit product (plain stores / read-modify-write accumulation)
    auto al = [](const void* ptr, int by) { return ((uintptr_t)ptr % by) == 0; };
    const bool misaligned = !al(p.C, p.c_dtype != XVA_F32 ? 8 : 16) || (p.R && !al(p.R, p.r_dtype != XVA_F32 ? 8 : 16)) ||
                            (p.G && !al(p.G, p.g_dtype != XVA_F32 ? 8 : 16)) || (p.C2 && !al(p.C2, p.c_dtype != XVA_F32 ? 8 : 16));
    for (int64_t e = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; e < MN; e += (int64_t)gridDim.x * 1024) {
        float4 s = *reinterpret_cast<const float4*>(slab + k0 * MN + e);
        for (int k = k0 + 1; k < k1; ++k) {
            float4 t = *reinterpret_cast<const float4*>(slab + k * MN + e);
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        float v[4] = {s.x, s.y, s.z, s.w};
        if (gridDim.z > 1) {
            const int row = (int)(e / p.N), col = (int)(e - (int64_t)row * p.N);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t ci = coff + (p.c_trans ? (int64_t)(col + q) * p.ldc + row : (int64_t)row * p.ldc + col + q);
                atomicAdd(reinterpret_cast<float*>(p.C) + ci, p.alpha * v[q]);
            }
            continue;
        }
        const int row = (int)(e / p.N), col = (int)(e - (int64_t)row * p.N);   // N % 4 == 0: the 4 values share a row
        // the FULL epilogue of the tile kernels (bias, alpha, dropout, gate, residual, activation, row mask, dtype, transposed store, accumulate), so
        // that products with a non-linear epilogue can be split along K as well (xva_gemm: forward / backward-data products with a long reduction
        // and a grid far below 256 workgroups)
        bool live = true;
        if (p.mask_mode != XVA_MASK_NONE) {
            const int64_t rr = (int64_t)row * p.mask_mul + p.mask_add;
            const int t = (int)(rr % p.Tp);
            live = t >= p.mask_pad && t < p.mask_pad + p.mask_len;
            if (live && p.mask_mode == XVA_MASK_LEN) live = (t - p.mask_pad) < p.lens[rr / p.Tp];
        }
        const int64_t goff = (int64_t)z1 * p.sG + (int64_t)z2 * p.sG2;
        const bool vec = p.c_trans == 0 && (p.ldc % 4 == 0) && (p.sC % 4 == 0) && (p.sC2 % 4 == 0) &&
                         (!p.R || (p.ldr % 4 == 0 && p.sR % 4 == 0 && p.sR2 % 4 == 0)) && (!p.G || (p.ldg % 4 == 0 && p.sG % 4 == 0 && p.sG2 % 4 == 0)) && !misaligned;
        const f32x4 a4 = {v[0], v[1], v[2], v[3]};
        const bool f16 = p.c_dtype == XVA_F16 || (p.R && p.r_dtype == XVA_F16) || (p.G && p.g_dtype == XVA_F16);      // the flavour of the problem's 16-bit tensors
        if (f16) {
            if (vec) epilogue4<true, true>(q1, a4, row, col, live, true, z2, coff, roff, goff);
            else epilogue4<false, true>(q1, a4, row, col, live, true, z2, coff, roff, goff);
        } else if (vec) epilogue4<true>(q1, a4, row, col, live, true, z2, coff, roff, goff);
        else epilogue4<false>(q1, a4, row, col, live, true, z2, coff, roff, goff);
    }
}

}  // namespace xva_glds

// Can this problem take the direct-to-LDS path?  (bf16-stored operands, bf16 MFMA, tap-segment lengths that divide or are divided
// by the 64-deep K tile, K-block lengths that are multiples of it, 8-element granularity of every index-contiguous dimension.)
bool xva_gemm_glds_eligible(const xva_gemm_params& p) {
    if (p.compute != 1 || p.a_dtype == XVA_F32 || p.b_dtype != p.a_dtype) return false;
    // (TN: k indexes ROWS of both operands — rows past K come from the zero page lane by lane; taken for the split-bf16 pairs, whose products have no other kernel)
    const bool only_here = p.planes || p.a_dtype == XVA_F16;      // products with no other kernel
    if ((p.K % 8 != 0 && !(only_here && p.layout == XVA_GEMM_TN)) || p.K < 8) return false;
    if (p.layout == XVA_GEMM_TN) {
        // split-bf16 pairs: M need not be a multiple of 8 when every k-row of A is readable up to the next multiple (lda covers it): the extra columns only
        // reach C rows >= M, which are never stored (FastPitch's attention gradients: M = T + 2 keys, rows of Ts = round-up-8 elements)
        const bool m_ok = p.M % 8 == 0 || (only_here && p.a_seglen == 0 && p.lda >= ((p.M + 7) & ~7));
        if (!m_ok || p.N % 8 != 0 || p.M < 8 || p.N < 8) return false;
        if (p.kb_len > 0 && p.K % p.kb_len != 0) return false;
    } else {
        auto seg_ok = [](int len) { return len <= 0 || len % xva_glds::GK == 0 || xva_glds::GK % len == 0; };   // a K tile maps onto whole segments
        if (p.a_segadj != 0 && !seg_ok(p.a_seglen)) return false;
        if (p.layout == XVA_GEMM_NN) {
            if (p.N % 8 != 0 || p.N < 8) return false;
            if (!seg_ok(p.seglen)) return false;
        }
    }
    return true;
}

// C (+)= alpha * sum of the p.splitk slabs in p.sk_ws (written by a split-K launch of any of the kernels)
int xva_gemm_launch_splitk_reduce(const xva_gemm_params& p, hipStream_t st) {
    using namespace xva_glds;
    const int64_t quads = (int64_t)p.M * p.N / 4;
    int gx = (int)((quads + 255) / 256); if (gx > 2048) gx = 2048; if (gx < 1) gx = 1;
    // few output elements, many slabs: spread the slabs over gridDim.z (needs a purely additive fp32 epilogue)
    int gz = 1;
    if (p.splitk > 32 && gx * p.batch * p.batch2 < 256 && p.accumulate && p.c_dtype == XVA_F32 && !p.bias && !p.R) {
        gz = 256 / (gx * p.batch * p.batch2);
        if (gz > p.splitk / 8) gz = p.splitk / 8;
        if (gz < 1) gz = 1;
    }
    hipLaunchKernelGGL(xva_gemm_splitk_reduce_kernel, dim3(gx, p.batch * p.batch2, gz), dim3(256), 0, st, p);
    return 0;
}
// K loop of the 256 x 256 tile: 0 = all waves in one phase (two barriers per 64-deep K tile), 1 (default) = two wave groups one barrier
// apart (xva_gemm_glds8_kernel), 2 = the latter for NT only.  Measured (tools/gemm_tile_ab.py): NT +5 ... +18 %, NN +2 ... +6 %, TN +-1 % on
// warm operands; inside the training steps (operands from HBM) FastPitch -0.8 %, HiFi-GAN -1.0 % step time.
static int g_kloop8_v = 1;
extern "C" int xva_gemm_set_kloop(int mode) { int old = g_kloop8_v; g_kloop8_v = mode; return old; }
int xva_gemm_glds_kloop8() { return g_kloop8_v; }
// K loop of the 384 x 128 tile: 1 (default) = the staggered loop, 0 = the lock-step loop of xva_gemm_glds_kernel.
// NT products on the 256 x 256 tile: 1 (default) = whole-line DMA pieces over a ring of five operand units (xva_gemm_glds8w_kernel), 0 = the 32-deep tiles of
// xva_gemm_glds8_kernel for every layout
static int g_wholeline_v = XVA_GLDS8_WHOLE;
extern "C" int xva_gemm_set_wholeline(int mode) { int old = g_wholeline_v; g_wholeline_v = mode; return old; }
namespace xva_glds { int xva_gemm_glds_wholeline() { return g_wholeline_v; } }
static int g_kloop384_v = 1;
extern "C" int xva_gemm_set_kloop384(int mode) { int old = g_kloop384_v; g_kloop384_v = mode; return old; }
int xva_gemm_glds_kloop384() { return g_kloop384_v; }
#endif   // !XVA_GLDS_F16
#define vec_epilogue_ok xva_gemm_vec_epilogue_ok
#define g_kloop8 xva_gemm_glds_kloop8()
#define g_kloop384 xva_gemm_glds_kloop384()
static int launch_tiles(const xva_gemm_params& p, int tile, int vec, hipStream_t st);

// tile: see launch_tiles
#if XVA_GLDS_F16
int xva_gemm_launch_glds_f16(const xva_gemm_params& pin, int tile, hipStream_t st) {
#else
int xva_gemm_launch_glds(const xva_gemm_params& pin, int tile, hipStream_t st) {
#endif
    using namespace xva_glds;
    xva_gemm_params p = pin;
    // split-K through slabs needs N % 4 == 0 and enough scratch; otherwise fall back to fp32 atomics
    const int64_t need = (int64_t)p.splitk * p.batch * p.batch2 * (int64_t)p.M * p.N * 4;
    if (p.splitk <= 1 || !p.sk_ws || p.sk_ws_bytes < need || p.N % 4 != 0 || ((uintptr_t)p.sk_ws % 16) != 0) p.sk_ws = nullptr;
    int vec = vec_epilogue_ok(p);
    if (p.sk_ws && p.N % 8 == 0) vec = 2;        // slab stores are [M][N] fp32 rows whatever C looks like
    // Non-temporal stores for outputs that cannot stay in the 8 x 4 MB of L2 anyway (FastPitch's 27 584 x 1 536 feed-forward intermediate: 85 MB):
    // written the default way they evict the weight / activation panels the tile's next rounds (and the other lane's kernels) re-read.
    static const long nt_mb = 0;      // (A/B at 12 / 30 / 60 MB thresholds, round 3: nothing outside the run-to-run band; off)
    if (nt_mb > 0 && vec == 2 && !p.sk_ws && !p.accumulate && !p.C2 && !p.c_trans && p.c_dtype != XVA_F32 &&
        (int64_t)p.M * p.N * 2 * p.batch * p.batch2 >= nt_mb * (1L << 20)) vec |= 16;
    int rc = launch_tiles(p, tile, vec, st);
    if (rc == 0 && p.sk_ws) rc = xva_gemm_launch_splitk_reduce(p, st);
    return rc;
}

template <int BM, int BN, int WM, int WN>
static int launch_layout(const xva_gemm_params& p, int vec, hipStream_t st) {
    using namespace xva_glds;
    switch (p.layout) {
        case XVA_GEMM_NT: return launch_tile<XVA_GEMM_NT, BM, BN, WM, WN, F16V>(p, vec, st);
        case XVA_GEMM_NN: return launch_tile<XVA_GEMM_NN, BM, BN, WM, WN, F16V>(p, vec, st);
        default: return launch_tile<XVA_GEMM_TN, BM, BN, WM, WN, F16V>(p, vec, st);
    }
}
// tile: 0 = 128x128 (4 waves of 64x64), 1 = 256x256 (8 waves of 128x64), 2 = 128x64 (4 waves of 32x64), 3 = 64x64 (4 waves of 32x32),
//       4 = 128x32 (4 waves of 32x32), 5 = 384x128 (8 waves of 96x64; NT / NN only: a 384-wide index-contiguous operand image is not laid out)
static int launch_tiles(const xva_gemm_params& p, int tile, int vec, hipStream_t st) {
    switch (tile) {
        case 1:
            if (g_kloop8 == 1 || (g_kloop8 == 2 && p.layout == XVA_GEMM_NT)) {
                switch (p.layout) {
                    case XVA_GEMM_NT: return xva_glds::launch_tile8<XVA_GEMM_NT, 256, 256, 128, 64, F16V>(p, vec, st);
                    case XVA_GEMM_NN: return xva_glds::launch_tile8<XVA_GEMM_NN, 256, 256, 128, 64, F16V>(p, vec, st);
                    default: return xva_glds::launch_tile8<XVA_GEMM_TN, 256, 256, 128, 64, F16V>(p, vec, st);
                }
            }
            return launch_layout<256, 256, 128, 64>(p, vec, st);
        case 2: return launch_layout<128, 64, 32, 64>(p, vec, st);
        case 3: return launch_layout<64, 64, 32, 32>(p, vec, st);
        case 4: return launch_layout<128, 32, 32, 32>(p, vec, st);       // N <= 32: no padded columns through the matrix pipe
        case 5:                                                          // 256 < N <= 384 (FastPitch d_model): no padded columns, one workgroup per CU
            if (g_kloop384) {                                            // the staggered K loop (xva_gemm_glds8_kernel<.., 384, 128, 96, 64>)
                if (p.layout == XVA_GEMM_NT) return xva_glds::launch_tile8<XVA_GEMM_NT, 384, 128, 96, 64, F16V>(p, vec, st);
                if (p.layout == XVA_GEMM_NN) return xva_glds::launch_tile8<XVA_GEMM_NN, 384, 128, 96, 64, F16V>(p, vec, st);
            }
            return launch_layout<384, 128, 96, 64>(p, vec, st);
        case 6:                                                          // 256x128, K tile 32, two workgroups per CU
            switch (p.layout) {
                case XVA_GEMM_NT: return xva_glds::launch_tile3<XVA_GEMM_NT, 256, 128, F16V>(p, vec, st);
                case XVA_GEMM_NN: return xva_glds::launch_tile3<XVA_GEMM_NN, 256, 128, F16V>(p, vec, st);
                default: return xva_glds::launch_tile3<XVA_GEMM_TN, 256, 128, F16V>(p, vec, st);
            }
        default: return launch_layout<128, 128, 64, 64>(p, vec, st);
    }
}
#if !XVA_GLDS_F16
void xva_gemm_glds_tile_dims(int tile, int* bm, int* bn) {
    static const int d[7][2] = {{128, 128}, {256, 256}, {128, 64}, {64, 64}, {128, 32}, {384, 128}, {256, 128}};
    *bm = d[tile % 7][0]; *bn = d[tile % 7][1];
}

#undef vec_epilogue_ok
int xva_gemm_vec_epilogue_ok(const xva_gemm_params& p) {
    auto al = [](const void* q, int b) { return ((uintptr_t)q % b) == 0; };
    auto level = [&](int n) {   // n-element granularity of every row the epilogue touches
        const int ab = n == 4 ? 8 : 16;                                     // bf16 rows: 8 / 16 bytes; fp32 rows: 16 bytes either way
        int ok = (p.N % n == 0) && (p.ldc % n == 0) && (p.sC % n == 0) && (p.sC2 % n == 0) && al(p.C, p.c_dtype != XVA_F32 ? ab : 16);
        if (p.R) ok = ok && (p.ldr % n == 0) && (p.sR % n == 0) && (p.sR2 % n == 0) && al(p.R, p.r_dtype != XVA_F32 ? ab : 16);
        if (p.G) ok = ok && (p.ldg % n == 0) && (p.sG % n == 0) && (p.sG2 % n == 0) && al(p.G, p.g_dtype != XVA_F32 ? ab : 16);
        if (p.F) ok = ok && al(p.F, p.g_dtype != XVA_F32 ? ab : 16);
        return ok;
    };
    if (!level(4)) return 0;
    if (level(8) && (!p.bias || (al(p.bias, 16) && p.sbias2 % 4 == 0))) return 2;
    return 1;
}

#define vec_epilogue_ok xva_gemm_vec_epilogue_ok
#endif   // !XVA_GLDS_F16
// ---- convolutions over 8 / 16 / 32 / 64 / 128 input channels (per group), stride 1 / 2 / 4: resident-input kernel (gemm_glds.h) ----------
// (bf16 only: no caller runs these layers on fp16 operands — xva_gemm_conv_res_plan declines them and the general tiles take the product)
#if !XVA_GLDS_F16
template <int LAYOUT, int CIN>
static int conv_res_bn(const xva_gemm_params& p, int vec, int dstep, int stride, int64_t rp, hipStream_t st) {
    using namespace xva_glds;
    if (p.N > 64) return launch_conv_res<LAYOUT, CIN, 128, 64, 64, F16V>(p, vec, dstep, stride, rp, st);
    if (p.N > 32) return launch_conv_res<LAYOUT, CIN, 64, 32, 64, F16V>(p, vec, dstep, stride, rp, st);
    return launch_conv_res<LAYOUT, CIN, 32, 32, 32, F16V>(p, vec, dstep, stride, rp, st);
}
// The resident-input plan of a problem: signed rows between consecutive taps (0 = does not qualify), input rows per output row, row pitch.
//   forward (NT): A(r, tap j, c) = X[(stride * r + j * d) * rowpitch + c]  ->  lda = stride * rowpitch, a_seglen + a_segadj = d * rowpitch
//   backward-data (NN): the same over dY, taps walking backwards; one polyphase component of a strided conv is a stride-1 problem
//   grouped: a_seglen = channels per group < rowpitch, the group index is the second batch level
int xva_gemm_conv_res_plan(const xva_gemm_params& p, int* stride_out, int64_t* rowpitch_out) {
    if (!xva_gemm_glds_eligible(p) || p.layout == XVA_GEMM_TN || p.splitk != 1 || p.a_dtype != XVA_BF16) return 0;
    const int cin = p.a_seglen;
    const int64_t rp = p.a_rowpitch > 0 ? p.a_rowpitch : p.lda;
    if (!(cin == 8 || cin == 16 || cin == 32 || cin == 64 || cin == 128) || rp < cin || rp % 8 != 0 || p.lda % rp != 0 || p.K % cin != 0 || p.K / cin < 2) return 0;
    const int stride = (int)(p.lda / rp);
    if (!(stride == 1 || stride == 2 || stride == 4)) return 0;
    if (p.batch2 > 1 && p.sA2 % 8 != 0) return 0;                    // a group's channels start on a 16-byte boundary
    {   // resident tile + two weight stages inside the 160 KiB of a CU
        const int bn = p.N > 64 ? 128 : (p.N > 32 ? 64 : 32);
        if (xva_glds::res_a_bytes(cin, stride) + 2 * bn * xva_glds::GK * 2 > 160 * 1024) return 0;
    }
    const int64_t step = (int64_t)p.a_seglen + p.a_segadj;          // elements between consecutive taps of one output row
    if (step == 0 || step % rp != 0) return 0;
    const int dstep = (int)(step / rp);                             // > 0: forward convolution, < 0: backward-data (taps walk backwards)
    if ((int64_t)(p.K / cin - 1) * (dstep < 0 ? -dstep : dstep) > xva_glds::RES_HALO) return 0;
    if (p.layout == XVA_GEMM_NN && (p.N % 8 != 0)) return 0;
    if (stride_out) *stride_out = stride;
    if (rowpitch_out) *rowpitch_out = rp;
    return dstep;
}
template <int LAYOUT>
static int conv_res_cin(const xva_gemm_params& p, int cin, int vec, int dstep, int stride, int64_t rp, hipStream_t st) {
    switch (cin) {
        case 8: return conv_res_bn<LAYOUT, 8>(p, vec, dstep, stride, rp, st);
        case 16: return conv_res_bn<LAYOUT, 16>(p, vec, dstep, stride, rp, st);
        case 32: return conv_res_bn<LAYOUT, 32>(p, vec, dstep, stride, rp, st);
        case 64: return conv_res_bn<LAYOUT, 64>(p, vec, dstep, stride, rp, st);
        default: return conv_res_bn<LAYOUT, 128>(p, vec, dstep, stride, rp, st);
    }
}
int xva_gemm_launch_conv_res(const xva_gemm_params& p, int dstep, hipStream_t st) {
    const int vec = vec_epilogue_ok(p);
    int stride = 1; int64_t rp = p.lda;
    if (xva_gemm_conv_res_plan(p, &stride, &rp) == 0) return -1;
    if (p.layout == XVA_GEMM_NT) return conv_res_cin<XVA_GEMM_NT>(p, p.a_seglen, vec, dstep, stride, rp, st);
    return conv_res_cin<XVA_GEMM_NN>(p, p.a_seglen, vec, dstep, stride, rp, st);
}
#endif   // !XVA_GLDS_F16
