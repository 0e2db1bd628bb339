// conv_pair.hip — forward of one ResBlock1 pair (python/hifigan/models.py:41-48) of the 32 / 64-channel generator stages as ONE launch.
//
// Unfused, a pair is two resident-input convolutions (gemm_glds.h: xva_conv_res_kernel) and six tensor passes over HBM: read lrelu(x), write the activated
// intermediate t; read t, read x, write the block output and its activated copy.  Here a workgroup owns R = 128 - (k2 - 1) output rows of one item:
//   1. its 128 + (k1 - 1) d1 input rows travel HBM -> LDS once (global_load_lds, the resident tile of xva_conv_res_kernel);
//   2. phase 1 runs the dilated convolution over 128 rows (the R output rows + the (k2 - 1) / 2 halo rows the second convolution needs either side),
//      adds the bias, applies the LeakyReLU, zeroes rows outside the item (the second convolution's zero padding) and writes the bf16 tile to LDS;
//   3. phase 2 runs the second convolution from that tile (same K loop, weights through the same two-stage ring) and ends in the row-contiguous
//      epilogue of the tile kernels (bias, residual, 1/3 scaling and accumulation into the stage sum, activated copy);
//   4. the R own rows of the intermediate are stored from LDS with 16-byte row pieces (the backward pass reads them: gate and weight-gradient operand).
// Tensor passes: 5 (x_raw = 0: operand = the stored activated copy, residual from HBM; bit-identical to the two launches) or 4 (the raw block input is
// staged; x_raw = 1: LeakyReLU applied to the operand fragments, the residual comes from the resident tile; x_raw = 2: the tile is activated in place once
// and the residual rows are re-read through the epilogue — they were fetched for the tile a few microseconds before).  The halo rows are recomputed by the neighbouring
// workgroups: (k2 - 1) / 128 = 1.6 ... 7.8 % more MFMA work in both phases.
#include "gemm_glds.h"
#include "conv_pair.h"

namespace xva_glds {

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int PAIR_BM = 128, PAIR_T_ROWS = 144;          // rows of the intermediate tile in LDS: 128 computed + the last taps' reach (never-stored outputs)

template <int C>
__global__ __launch_bounds__(256, C <= 32 ? 4 : 2) void xva_conv_pair_kernel(xva_gemm_params p, xva_conv_pair e) {
    constexpr int BM = PAIR_BM, WM = 32, WN = C, NW = 4, MI = WM / 16, NJ = WN / 16;
    constexpr int B_BYTES = C * GK * 2, T_BYTES = PAIR_T_ROWS * C * 2;
    constexpr int CPR = C / 8, RPI = 64 / CPR;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;

    const int k1 = e.k1, d1 = e.d1, k2 = p.K / C;
    const int halo1 = (k1 - 1) * d1, h2 = (k2 - 1) / 2;
    const int rout = BM - 2 * h2;
    const int nby = (p.M + rout - 1) / rout;
    int Lg;
    {
        const unsigned total = gridDim.x, id = blockIdx.x;
        const unsigned xcd = id & 7u, slot = id >> 3, q = total >> 3, r = total & 7u;
        Lg = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot);
    }
    const int tm = Lg % nby, z1 = Lg / nby;
    const int m0 = tm * rout;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave;
    const uint16_t* X = reinterpret_cast<const uint16_t*>(e.X) + (int64_t)z1 * e.sX;
    const int nrows = BM + halo1, ninstr = (nrows + RPI - 1) / RPI;
    const int X_BYTES = ninstr * 1024;
    XVA_LDS uint8_t* Tt = smem + X_BYTES;
    XVA_LDS uint8_t* ring = Tt + T_BYTES;
    XVA_LDS uint8_t* scr = ring + 2 * B_BYTES;

    {   // resident input rows m0 .. m0 + 127 + halo1 (row 0 = valid row -(h1 + h2)); rows past the item's pad rows clamp onto the last readable one
        const int rmax = p.M - 1 + halo1 + 2 * h2;
        for (int q = wave; q < ninstr; q += NW) {
            const int r = q * RPI + lane / CPR, pch = lane % CPR;
            const int c = pch ^ res_swz<C>(r);
            const uint16_t* src = X + (int64_t)min(m0 + r, rmax) * e.ldx + c * 8;
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)src, (XVA_LDS void*)(smem + q * 1024), 16, 0, 0);
        }
    }
    Loader<KC, C, NW> lb;
    lb.init(lane, wave, 0, C, p.ldb, 0, 0, 0);             // both weight matrices are [C][k * C]: ldb = k2 * C serves the second, the first passes its own below
    Loader<KC, C, NW> lb1;
    lb1.init(lane, wave, 0, C, (int64_t)k1 * C, 0, 0, 0);
    KcReader krb;
    krb.init(lane);
    const int g = lane >> 4;

    f32x4 acc[MI][NJ];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();
    // bias of the first convolution for this lane's columns (j * 16 + g * 4 ..): loaded before the K loop
    float b1[NJ][4];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const float4 b = e.bias1 ? *reinterpret_cast<const float4*>(e.bias1 + j * 16 + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        b1[j][0] = b.x; b1[j][1] = b.y; b1[j][2] = b.z; b1[j][3] = b.w;
    }

    constexpr int LOADS = Loader<KC, C, NW>::NI;
    constexpr int WAIT_YOUNGEST = 0x0F70 | (LOADS & 15) | ((LOADS >> 4) << 14);       // s_waitcnt vmcnt(LOADS)
    // One convolution over a resident LDS tile `At` (rows of C channels, chunks swizzled by res_swz): the K loop of xva_conv_res_kernel.
    // arow: this lane's tile row for output row block 0; taps step by dstep rows.  The first two weight tiles are already in flight (prime()).
    auto conv_loop = [&](const XVA_LDS uint8_t* At, int arow, int dstep, int ntaps, const uint16_t* Bw, const Loader<KC, C, NW>& ld, int K, bool a_lrelu, float a_slope) {
        const int nkt = (K + GK - 1) / GK;
        auto read_frags = [&](const XVA_LDS uint8_t* Bt, int kt, int kh, bf16x8 (&af)[MI], Frag<KC> (&bfr)[NJ]) {
            const int kl = kt * GK + kh * 32 + g * 8;
            const int tap = min(kl / C, ntaps - 1);
            const int ch = (kl % C) / 8;
            const int shift = tap * dstep;
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = krb.read(Bt, j * 16, kh);
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = arow + i * 16 + shift;
                af[i] = *reinterpret_cast<const XVA_LDS bf16x8*>(At + r * (C * 2) + ((ch ^ res_swz<C>(r)) << 4));
            }
            if (a_lrelu) {
#pragma unroll
                for (int i = 0; i < MI; ++i) af[i] = lrelu_frag(af[i], a_slope);
            }
        };
        auto mfma_all = [&](const bf16x8 (&af)[MI], Frag<KC> (&bfrr)[NJ]) {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_value(bfrr[j]), af[i], acc[i][j], 0, 0, 0);
        };
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const XVA_LDS uint8_t* Bt = ring + cur * B_BYTES;
            bf16x8 af[MI];
            Frag<KC> bfr[NJ];
            read_frags(Bt, kt, 0, af, bfr);
            mfma_all(af, bfr);
            read_frags(Bt, kt, 1, af, bfr);
            __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
            __builtin_amdgcn_s_barrier();                            // A: the weight buffer of tile kt is free
            const bool more = kt + 2 < nkt;
            if (more) ld.issue(Bw + (kt + 2) * GK, (kt + 2) * GK, K, ring + cur * B_BYTES, wave);
            mfma_all(af, bfr);
            if (more) __builtin_amdgcn_s_waitcnt(WAIT_YOUNGEST); else __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_s_barrier();                            // B: tile kt + 1 has landed
        }
    };
    auto prime = [&](const uint16_t* Bw, const Loader<KC, C, NW>& ld, int K) {
        ld.issue(Bw, 0, K, ring, wave);
        if (K > GK) ld.issue(Bw + GK, GK, K, ring + B_BYTES, wave);
    };

    // ---- phase 1: t = lrelu(conv1(x) + b1) over tile rows 0 .. 127 = valid rows m0 - h2 .. m0 - h2 + 127
    const uint16_t* W1 = reinterpret_cast<const uint16_t*>(e.W1);
    const int K1 = k1 * C;
    prime(W1, lb1, K1);
    if (K1 > GK) __builtin_amdgcn_s_waitcnt(WAIT_YOUNGEST); else __builtin_amdgcn_s_waitcnt(0x0F70);     // the input tile and weight tile 0 have landed
    __builtin_amdgcn_s_barrier();
    if (e.x_raw == 2) {      // the raw input is activated ONCE, in place (the residual then comes through the epilogue: the same rows, from the L2 that just served them)
        for (int idx = threadIdx.x; idx < ninstr * 64; idx += 256) {
            XVA_LDS bf16x8* q = reinterpret_cast<XVA_LDS bf16x8*>(smem + idx * 16);
            *q = lrelu_frag(*q, e.x_slope);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __builtin_amdgcn_s_barrier();
    }
    conv_loop(smem, wm * WM + (lane & 15), d1, k1, W1, lb1, K1, e.x_raw == 1, e.x_slope);

    // the second convolution's first weight tiles travel while the intermediate is written
    const uint16_t* W2 = reinterpret_cast<const uint16_t*>(p.B);
    prime(W2, lb, p.K);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int q = wm * WM + i * 16 + (lane & 15);
        const int s = m0 - h2 + q;
        const bool live = s >= 0 && s < p.M;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = j * 16 + g * 4;
            float v[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { v[t] = lrelu(acc[i][j][t] + b1[j][t], e.slope1); if (!live) v[t] = 0.f; }
            *reinterpret_cast<XVA_LDS u32x2*>(Tt + q * (C * 2) + (((c >> 3) ^ res_swz<C>(q)) << 4) + (c & 7) * 2) = (u32x2){pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
        }
    }
    // ---- phase 2: y = conv2(t) over output rows m0 .. m0 + rout - 1 (tile rows q .. q + k2 - 1 of t)
    if (e.x_raw == 1) {     // residual from the resident tile: the accumulators start from x (alpha == beta: alpha * (conv + b + x))
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int xr = wm * WM + i * 16 + (lane & 15) + halo1 / 2 + h2;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int c = j * 16 + g * 4;
                const u32x2 r = *reinterpret_cast<const XVA_LDS u32x2*>(smem + xr * (C * 2) + (((c >> 3) ^ res_swz<C>(xr)) << 4) + (c & 7) * 2);
                acc[i][j] = (f32x4){__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xffff0000u), __uint_as_float(r.y << 16), __uint_as_float(r.y & 0xffff0000u)};
            }
        }
    } else zero_acc();
    if (p.K > GK) __builtin_amdgcn_s_waitcnt(WAIT_YOUNGEST); else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();                                    // the intermediate tile is complete, weight tile 0 of conv2 has landed
    conv_loop(Tt, wm * WM + (lane & 15), 1, k2, W2, lb, p.K, false, 0.f);

    xva_gemm_params pp = p;
    pp.M = min(p.M, m0 + rout);                                      // rows past this workgroup's range belong to the next one
    tile_epilogue_rows<MI, NJ>(pp, acc, reinterpret_cast<XVA_LDS float*>(scr + wave * epi_scratch_bytes(WN)), m0 + wm * WM, 0, lane, z1, 0, z1, 0);

    // ---- the own rows of the intermediate: LDS -> HBM in 16-byte row pieces
    uint16_t* T1 = reinterpret_cast<uint16_t*>(e.T1) + (int64_t)z1 * e.sT1;
    for (int idx = threadIdx.x; idx < rout * CPR; idx += 256) {
        const int rr = idx / CPR, pc = idx % CPR;
        const int q = h2 + rr, s = m0 + rr;
        if (s < p.M) {
            const u32x4 v = *reinterpret_cast<const XVA_LDS u32x4*>(Tt + q * (C * 2) + ((pc ^ res_swz<C>(q)) << 4));
            *reinterpret_cast<u32x4*>(T1 + (int64_t)s * e.ldt + pc * 8) = v;
        }
    }
}

template <int C>
static int launch_pair(const xva_gemm_params& p, const xva_conv_pair& e, hipStream_t st) {
    constexpr int CPR = C / 8, RPI = 64 / CPR;
    constexpr int LDS_MAX = ((PAIR_BM + 64 + RPI - 1) / RPI) * 1024 + PAIR_T_ROWS * C * 2 + 2 * C * GK * 2 + 4 * epi_scratch_bytes(C);
    const int halo1 = (e.k1 - 1) * e.d1;
    const int LDS = ((PAIR_BM + halo1 + RPI - 1) / RPI) * 1024 + PAIR_T_ROWS * C * 2 + 2 * C * GK * 2 + 4 * epi_scratch_bytes(C);
    auto kern = xva_conv_pair_kernel<C>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess) return -1;
        attr_set = true;
    }
    const int k2 = p.K / C, rout = PAIR_BM - (k2 - 1);
    const long nblocks = (long)xva_cdiv(p.M, rout) * p.batch;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), LDS, st, p, e);
    return 0;
}

}  // namespace xva_glds

// per-launch profiling records of the GEMM entry point (core.hip): the pair registers itself with the algorithmic work of BOTH convolutions
bool xva_prof_is_on();
void xva_prof_begin(hipStream_t st, double flops, int variant);
void xva_prof_end(hipStream_t st);
void xva_prof_shape(int M, int N, int K, int batch, int splitk, int bn, double bytes);

int xva_conv_pair_fwd(const xva_gemm_params* conv2, const xva_conv_pair* conv1, void* stream) {
    const xva_gemm_params& p = *conv2;
    const xva_conv_pair& e = *conv1;
    auto al16 = [](const void* q) { return ((uintptr_t)q % 16) == 0; };
    const int C = p.N;
    if (!(C == 32 || C == 64)) return -1;
    if (p.layout != XVA_GEMM_NT || p.compute == 0 || p.a_dtype != XVA_BF16 || p.b_dtype != XVA_BF16 || p.c_dtype != XVA_BF16) return -1;
    if (p.batch2 > 1 || p.splitk != 1 || p.c_trans || p.G || p.mask_mode != XVA_MASK_NONE || p.drop_p > 0.f || p.c_plane || p.planes) return -1;
    if (p.K % C != 0 || p.ldb != p.K) return -1;
    const int k2 = p.K / C;
    if (k2 < 1 || k2 > 11 || (k2 & 1) == 0 || e.k1 < 1 || (e.k1 & 1) == 0 || e.d1 < 1 || (e.k1 - 1) * e.d1 > 64 || e.k1 * C < 2 * xva_glds::GK / 2) return -1;
    // the row-contiguous epilogue's alignment rules (gemm_glds.hip: vec_epilogue_ok level 8) + 16-byte rows of X / T1
    if (p.ldc % 8 || p.sC % 8 || !al16(p.C) || (p.C2 && !al16(p.C2)) || (p.bias && !al16(p.bias)) || (e.bias1 && !al16(e.bias1))) return -1;
    if (p.R && (p.r_dtype != XVA_BF16 || p.ldr % 8 || p.sR % 8 || !al16(p.R))) return -1;
    if (e.ldx % 8 || e.sX % 8 || !al16(e.X) || e.ldt % 8 || e.sT1 % 8 || !al16(e.T1) || !al16(e.W1) || !al16(p.B)) return -1;
    if (e.x_raw == 1 && (p.R || p.alpha != p.beta)) return -1;
    if (p.M < 1 || p.batch < 1) return -1;
    hipStream_t st = (hipStream_t)stream;
    const bool prof = xva_prof_is_on();
    if (prof) {
        // algorithmic work: both products (no halo rows); bytes: the input once, both weights, the intermediate and the output written once (read too when
        // accumulating), the residual when it comes from HBM — the intermediate is NOT re-read (tag 700000 + C; variant NT / bf16)
        const double rows = (double)p.M * p.batch, Kt = (double)(e.k1 + k2) * C;
        xva_prof_begin(st, 2.0 * rows * C * Kt, XVA_GEMM_NT * 3 + 1);
        double by = rows * C * 2.0 * (3.0 + (p.accumulate ? 1.0 : 0.0) + (p.R ? 1.0 : 0.0)) + Kt * C * 2.0;
        xva_prof_shape(p.M, C, (int)Kt, p.batch, 1, 700000 + C, by);
    }
    const int rc = C == 32 ? xva_glds::launch_pair<32>(p, e, st) : xva_glds::launch_pair<64>(p, e, st);
    if (prof) xva_prof_end(st);
    return rc;
}
