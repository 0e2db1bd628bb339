// gemm_core.h — LDS-tiled MFMA implicit-convolution GEMM for gfx950 (MI355X).  See include/xva_gemm.h for the
// operand model.  One template, instantiated per (layout, mode, N-tile) in gemm_fp32.hip / gemm_bf16.hip /
// gemm_mixed.hip so the three translation units build in parallel.
//
// Tile: 128 x BN x 32 per 256-thread workgroup = 4 wave64s in a 2x2 grid; each wave owns 64 x BN/2 = 4 x (BN/32)
// MFMA 16x16 accumulators.  BN in {128, 64, 32} so narrow-channel layers (HiFi-GAN C = 32 / 64, grouped convs)
// do not burn the matrix pipe on padding columns.
// Staging: HBM -> VGPR (16-byte vectors, coalesced along the operand's contiguous dimension) -> LDS as [row][k]
// (k contiguous) so one ds_read_b128 (bf16) / ds_read_b32 (fp32) yields an MFMA fragment.  Operands whose contiguous
// dimension is NOT k (NN's B, TN's A and B) are transposed in registers on the way (4x4 fp32 / 4x8 bf16 micro-tiles),
// with a lane->micro-tile map chosen so the transposed LDS stores are bank-conflict-free.  Loads of K-tile t+1 are
// issued before the MFMAs of K-tile t.  All fast-path loads are branch-free: M/N edges clamp onto valid memory (their
// products only reach never-stored rows/columns); only a ragged last K-tile takes the zero-filling path.
//
// MODE 0: fp32 storage, fp32 LDS, v_mfma_f32_16x16x4_f32 (exact fp32 = k-ordered fmaf chain): the parity mode.
// MODE 1: bf16 storage, bf16 LDS, v_mfma_f32_16x16x32_bf16.
// MODE 2: fp32 storage rounded to bf16 while staged, bf16 LDS, v_mfma_f32_16x16x32_bf16.
// MODE 3: fp32 storage SPLIT into two bf16 while staged (x = hi + lo, hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits), two bf16 LDS planes,
//         three v_mfma_f32_16x16x32_bf16 per product (lo*hi + hi*lo + hi*hi, fp32 accumulation; lo*lo <= 2^-16 of the term is dropped): products good
//         to ~1e-5 relative at a sixth of the matrix-pipe time of the exact fp32 MFMA (xva_gemm_set_fp32_products(1); the exact mode stays the default).
#pragma once
#include "xva_common.h"
#include "../../include/xva_gemm.h"

namespace xva_gemm_impl {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#ifndef XVA_BK
#define XVA_BK 32
#endif
constexpr int BM = 128, BK = XVA_BK, NTHREADS = 256;
constexpr int KH = BK / 32;   // 32-deep K halves per tile

template <int MODE> struct Cfg;
template <> struct Cfg<0> { typedef float S; typedef float L; static constexpr int VE = 4, LD = BK + 4; static constexpr bool BF = false; };
template <> struct Cfg<1> { typedef uint16_t S; typedef __bf16 L; static constexpr int VE = 8, LD = BK + 8; static constexpr bool BF = true; };
template <> struct Cfg<2> { typedef float S; typedef __bf16 L; static constexpr int VE = 4, LD = BK + 8; static constexpr bool BF = true; };
template <> struct Cfg<3> { typedef float S; typedef __bf16 L; static constexpr int VE = 4, LD = BK + 8; static constexpr bool BF = true; };

__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
__device__ __forceinline__ uint32_t pack_bf2(float a, float b);
__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float lrelu(float v, float s) { return v > 0.f ? v : v * s; }
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {   // round-to-nearest-even pair -> packed bf16x2
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float ld_elem(const void* p, int64_t idx, int dtype) {
    return dtype == XVA_BF16 ? bf2f(reinterpret_cast<const uint16_t*>(p)[idx]) : reinterpret_cast<const float*>(p)[idx];
}

// x = hi + lo in bf16 pairs: (a, b) -> packed hi pair, packed lo pair
__device__ __forceinline__ void split_bf2(float a, float b, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf2(a, b);
    lo = pack_bf2(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
// ---- a 16-byte staged vector and its LDS store ------------------------------------------------------------
// fp32 storage: 4 k-consecutive floats.  bf16 storage: 8 k-consecutive bf16.
// plane: elements between the hi and the lo image of an operand tile (MODE 3)
template <int MODE, bool ACT>
__device__ __forceinline__ void st_vec_k(typename Cfg<MODE>::L* dst, uint4 raw, float slope, int plane = 0) {
    if constexpr (MODE == 1) {
        if constexpr (ACT) {
            uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float lo = lrelu(__uint_as_float(w[i] << 16), slope), hi = lrelu(__uint_as_float(w[i] & 0xffff0000u), slope);
                w[i] = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
            }
            raw = make_uint4(w[0], w[1], w[2], w[3]);
        }
        *reinterpret_cast<uint4*>(dst) = raw;
    } else {
        float a = __uint_as_float(raw.x), b = __uint_as_float(raw.y), c = __uint_as_float(raw.z), d = __uint_as_float(raw.w);
        if constexpr (ACT) { a = lrelu(a, slope); b = lrelu(b, slope); c = lrelu(c, slope); d = lrelu(d, slope); }
        if constexpr (MODE == 2) {
            *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf2(a, b), pack_bf2(c, d));
        } else if constexpr (MODE == 3) {
            uint32_t h0, l0, h1, l1;
            split_bf2(a, b, h0, l0); split_bf2(c, d, h1, l1);
            *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(dst + plane) = make_uint2(l0, l1);
        } else {
            *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
        }
    }
}
// 4 k-consecutive values given as floats
template <int MODE>
__device__ __forceinline__ void st_quad(typename Cfg<MODE>::L* dst, float a, float b, float c, float d, int plane = 0) {
    if constexpr (MODE == 3) {
        uint32_t h0, l0, h1, l1;
        split_bf2(a, b, h0, l0); split_bf2(c, d, h1, l1);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dst + plane) = make_uint2(l0, l1);
    } else if constexpr (Cfg<MODE>::BF) {
        *reinterpret_cast<uint2*>(dst) = make_uint2(pack_bf2(a, b), pack_bf2(c, d));
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
    }
}

// zero the elements of a staged vector whose k index is >= Kbound (ragged last K-tile)
template <int MODE>
__device__ __forceinline__ uint4 mask_tail(uint4 raw, int k, int Kbound) {
    if constexpr (MODE == 1) {
        uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (k + 2 * i >= Kbound) w[i] = 0u;
            else if (k + 2 * i + 1 >= Kbound) w[i] &= 0x0000ffffu;
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    } else {
        if (k + 0 >= Kbound) raw.x = 0u;
        if (k + 1 >= Kbound) raw.y = 0u;
        if (k + 2 >= Kbound) raw.z = 0u;
        if (k + 3 >= Kbound) raw.w = 0u;
        return raw;
    }
}

// ---- KC stager: global X[i][k], k contiguous ------------------------------------------------------------------
template <int ROWS, int MODE>
struct KcStage {
    typedef Cfg<MODE> C;
    static constexpr int VPR = BK / C::VE;               // vectors per row: 8 (fp32) or 4 (bf16)
    static constexpr int NV = ROWS * VPR;
    static constexpr int IT = (NV + NTHREADS - 1) / NTHREADS;
    uint4 v[IT];

    // segj / segrem: this thread's tap index and offset inside the tap for the current K-tile (A segments only)
    template <bool TAIL>
    __device__ __forceinline__ void load(const typename C::S* __restrict__ X, int64_t ld, int i0, int Ibound, int k0, int Kbound,
                                         int64_t segoff) {
        const int t = threadIdx.x;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int vid = t + it * NTHREADS;
            if (NV % NTHREADS != 0 && vid >= NV) continue;
            const int row = vid / VPR, kv = vid % VPR;
            const int i = min(i0 + row, Ibound - 1);
            const int k = k0 + kv * C::VE;
            const typename C::S* src = X + (int64_t)i * ld + k + segoff;
            if constexpr (TAIL) {
                uint4 r = make_uint4(0u, 0u, 0u, 0u);
                if (k < Kbound) r = mask_tail<MODE>(*reinterpret_cast<const uint4*>(src), k, Kbound);
                v[it] = r;
            } else {
                v[it] = *reinterpret_cast<const uint4*>(src);
            }
        }
    }
    template <bool ACT>
    __device__ __forceinline__ void store(typename C::L* Xs, float slope) const {
        const int t = threadIdx.x;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int vid = t + it * NTHREADS;
            if (NV % NTHREADS != 0 && vid >= NV) continue;
            const int row = vid / VPR, kv = vid % VPR;
            st_vec_k<MODE, ACT>(Xs + row * C::LD + kv * C::VE, v[it], slope, ROWS * C::LD);
        }
    }
};

// ---- IC stager: global X[k][i], i contiguous; transposed in registers -----------------------------------------
// fp32 storage: 4(k) x 4(i) micro-tiles, lane -> (kg = l & 7, iv = wave * 8 + (l >> 3)) : 16 consecutive lanes write
//               8 k-groups x 2 i-vectors = all 32 banks once (conflict-free ds_write_b64 / b128).
// bf16 storage: 4(k) x 8(i) micro-tiles on threads [T0, T0 + ROWS): u -> (kg = u & 7, iv8 = u >> 3).
template <int ROWS, int MODE, int T0>
struct IcStage {
    typedef Cfg<MODE> C;
    uint4 v[4 * KH];

    __device__ __forceinline__ static bool active() {
        if constexpr (MODE == 1) return (int)threadIdx.x >= T0 && (int)threadIdx.x < T0 + ROWS;
        else if constexpr (ROWS == 128) return true;
        else return (int)(threadIdx.x >> 6) * 8 < ROWS / 4;
    }
    __device__ __forceinline__ static void coords(int& kg, int& icol) {
        if constexpr (MODE == 1) { const int u = threadIdx.x - T0; kg = u & 7; icol = (u >> 3) * 8; }
        else { const int l = threadIdx.x & 63, w = threadIdx.x >> 6; kg = l & 7; icol = (w * 8 + (l >> 3)) * 4; }
    }
    // rowptr(k) = rowbase + k * ld ; column offset `coloff` (segment adjusted, per thread constant)
    template <bool TAIL>
    __device__ __forceinline__ void load(const typename C::S* const (&rowbases)[KH], int64_t ld, int64_t col, int k0, int Kbound,
                                         int kb_len = 0, int64_t kb_stride = 0) {
        if (!active()) return;
        int kg, icol;
        coords(kg, icol);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const typename C::S* rowbase = rowbases[kh];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + kh * 32 + kg * 4 + j;
                int64_t roff = (int64_t)k * ld;
                if (kb_len > 0) {   // K runs over blocks of kb_len rows (batch items): k -> (block, row in block)
                    const int kc = k < Kbound ? k : Kbound - 1;
                    int blk = (int)(((float)kc + 0.5f) / (float)kb_len);
                    if ((int64_t)blk * kb_len > kc) --blk;
                    if ((int64_t)(blk + 1) * kb_len <= kc) ++blk;
                    roff = (int64_t)blk * kb_stride + (int64_t)(kc - blk * kb_len) * ld;
                }
                if constexpr (TAIL) {
                    v[kh * 4 + j] = (k < Kbound) ? *reinterpret_cast<const uint4*>(rowbase + roff + col) : make_uint4(0u, 0u, 0u, 0u);
                } else {
                    v[kh * 4 + j] = *reinterpret_cast<const uint4*>(rowbase + roff + col);
                }
            }
        }
    }
    template <bool ACT>
    __device__ __forceinline__ void store(typename C::L* Xs, float slope) const {
        if (!active()) return;
        int kg, icol;
        coords(kg, icol);
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            typename C::L* d = Xs + icol * C::LD + kh * 32 + kg * 4;
            const uint4* vv = v + kh * 4;
            if constexpr (MODE == 1) {
                const uint32_t w[4][4] = {{vv[0].x, vv[0].y, vv[0].z, vv[0].w}, {vv[1].x, vv[1].y, vv[1].z, vv[1].w},
                                          {vv[2].x, vv[2].y, vv[2].z, vv[2].w}, {vv[3].x, vv[3].y, vv[3].z, vv[3].w}};
#pragma unroll
                for (int e = 0; e < 8; ++e) {   // column e of the micro-tile -> 4 k-consecutive bf16
                    uint16_t h[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        uint32_t word = w[j][e >> 1];
                        h[j] = (e & 1) ? (uint16_t)(word >> 16) : (uint16_t)(word & 0xffffu);
                        if constexpr (ACT) h[j] = f2bf(lrelu(bf2f(h[j]), slope));
                    }
                    uint2 o = make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
                    *reinterpret_cast<uint2*>(d + e * C::LD) = o;
                }
            } else {
                const float f[4][4] = {{__uint_as_float(vv[0].x), __uint_as_float(vv[0].y), __uint_as_float(vv[0].z), __uint_as_float(vv[0].w)},
                                       {__uint_as_float(vv[1].x), __uint_as_float(vv[1].y), __uint_as_float(vv[1].z), __uint_as_float(vv[1].w)},
                                       {__uint_as_float(vv[2].x), __uint_as_float(vv[2].y), __uint_as_float(vv[2].z), __uint_as_float(vv[2].w)},
                                       {__uint_as_float(vv[3].x), __uint_as_float(vv[3].y), __uint_as_float(vv[3].z), __uint_as_float(vv[3].w)}};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = f[0][e], b = f[1][e], c = f[2][e], dd = f[3][e];
                    if constexpr (ACT) { a = lrelu(a, slope); b = lrelu(b, slope); c = lrelu(c, slope); dd = lrelu(dd, slope); }
                    st_quad<MODE>(d + e * C::LD, a, b, c, dd, ROWS * C::LD);
                }
            }
        }
    }
};

template <int LAYOUT, int MODE, int BN>
__global__ __launch_bounds__(NTHREADS, (MODE == 3 && BN == 128) ? 2 : 3) void xva_gemm_kernel(xva_gemm_params p) {
    typedef Cfg<MODE> C;
    typedef typename C::S ST;
    typedef typename C::L LT;
    constexpr int LD = C::LD;
    constexpr int NTN = BN / 32;   // MFMA column tiles per wave
    constexpr int PLANES = MODE == 3 ? 2 : 1;                       // MODE 3: hi image, then lo image
    __shared__ __attribute__((aligned(16))) LT As[PLANES * BM * LD];
    __shared__ __attribute__((aligned(16))) LT Bs[PLANES * BN * LD];

    // XCD-aware tile order: hardware places workgroup id on XCD id % 8 (each XCD has a private 4 MiB L2).  Give every XCD
    // a CONTIGUOUS run of logical tiles (n fastest, then m, then batch/split) so that the workgroups resident on one XCD
    // at the same time share their A row-panel and the B panel through that XCD's L2.
    const int nbx = (p.N + BN - 1) / BN, nby = (p.M + BM - 1) / BM;
    int Lg;
    {
        const unsigned total = gridDim.x, id = blockIdx.x;
        const unsigned xcd = id & 7u, slot = id >> 3, q = total >> 3, r = total & 7u;
        Lg = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot);
    }
    const int tn = Lg % nbx, tm = (Lg / nbx) % nby, z = Lg / (nbx * nby);
    const int bz = z / p.splitk, ks = z - bz * p.splitk;
    const int b2n = p.batch2 > 1 ? p.batch2 : 1;
    const int z1 = bz / b2n, z2 = bz - z1 * b2n;
    const int m0 = tm * BM, n0 = tn * BN;
    const ST* A = reinterpret_cast<const ST*>(p.A) + (int64_t)z1 * p.sA + (int64_t)z2 * p.sA2;
    const ST* B = reinterpret_cast<const ST*>(p.B) + (int64_t)z1 * p.sB + (int64_t)z2 * p.sB2;

    const int nkt_total = (p.K + BK - 1) / BK;
    const int per = (nkt_total + p.splitk - 1) / p.splitk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt_total, kt_begin + per);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x4 acc[4][NTN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NTN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- stagers ----
    KcStage<BM, MODE> ka;      // NT, NN: A
    KcStage<BN, MODE> kb;      // NT: B
    IcStage<BM, MODE, 0> ia;   // TN: A
    IcStage<BN, MODE, (LAYOUT == XVA_GEMM_TN ? 128 : 0)> ib;   // NN, TN: B

    // per-thread constants of the IC stagers: clamped column (+ TN column-segment adjustment)
    int64_t ia_col = 0, ib_col = 0;
    if constexpr (LAYOUT == XVA_GEMM_TN) {
        int kg, ic; ia.coords(kg, ic);
        const int MV = (p.M + C::VE - 1) / C::VE * C::VE;
        const int m = min(m0 + ic, MV - C::VE);
        ia_col = m;
        if (p.a_seglen > 0) ia_col = m + (int64_t)(m / p.a_seglen) * p.a_segadj;
    }
    if constexpr (LAYOUT != XVA_GEMM_NT) {
        int kg, ic; ib.coords(kg, ic);
        const int NV = (p.N + C::VE - 1) / C::VE * C::VE;
        int n = min(n0 + ic, NV - C::VE);
        ib_col = n;
        if (LAYOUT == XVA_GEMM_TN && p.seglen > 0) ib_col = p.seg0 + n + (int64_t)(n / p.seglen) * p.segstride;
    }
    // A tap-segment state for this thread's k column (KC A): advanced one K-tile at a time
    int aj = 0, arem = 0;
    if constexpr (LAYOUT != XVA_GEMM_TN) {
        if (p.a_seglen > 0) {
            const int kk = kt_begin * BK + (threadIdx.x % KcStage<BM, MODE>::VPR) * C::VE;
            aj = kk / p.a_seglen; arem = kk - aj * p.a_seglen;
        }
    }

    int bj[KH], brem[KH];
#pragma unroll
    for (int kh = 0; kh < KH; ++kh) { bj[kh] = 0; brem[kh] = 0; }
    if constexpr (LAYOUT == XVA_GEMM_NN) {
        if (p.seglen > 0) {
            int kg, ic; ib.coords(kg, ic);
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                const int kk = kt_begin * BK + kh * 32 + kg * 4;
                bj[kh] = kk / p.seglen; brem[kh] = kk - bj[kh] * p.seglen;
            }
        }
    }

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        const bool tail = (k0 + BK > p.K);
        int64_t aoff = 0;
        if constexpr (LAYOUT != XVA_GEMM_TN) {
            if (p.a_seglen > 0) {
                aoff = (int64_t)aj * p.a_segadj;
                arem += BK;
                while (arem >= p.a_seglen) { arem -= p.a_seglen; ++aj; }
            }
        }
        const ST* bbase[KH];
        const ST* abase[KH];
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) { bbase[kh] = B; abase[kh] = A; }
        if constexpr (LAYOUT == XVA_GEMM_NN) {
            if (p.seglen > 0) {   // a thread's 4 consecutive k-rows share one segment (seglen % 4 == 0, checked on the host)
#pragma unroll
                for (int kh = 0; kh < KH; ++kh) {
                    bbase[kh] = B + p.seg0 + (int64_t)bj[kh] * p.segstride - (int64_t)bj[kh] * p.seglen * p.ldb;
                    brem[kh] += BK;
                    while (brem[kh] >= p.seglen) { brem[kh] -= p.seglen; ++bj[kh]; }
                }
            }
        }
        if (!tail) {
            if constexpr (LAYOUT == XVA_GEMM_TN) ia.template load<false>(abase, p.lda, ia_col, k0, p.K, p.kb_len, p.kb_sA);
            else ka.template load<false>(A, p.lda, m0, p.M, k0, p.K, aoff);
            if constexpr (LAYOUT == XVA_GEMM_NT) kb.template load<false>(B, p.ldb, n0, p.N, k0, p.K, 0);
            else ib.template load<false>(bbase, p.ldb, ib_col, k0, p.K, LAYOUT == XVA_GEMM_TN ? p.kb_len : 0, p.kb_sB);
        } else {
            if constexpr (LAYOUT == XVA_GEMM_TN) ia.template load<true>(abase, p.lda, ia_col, k0, p.K, p.kb_len, p.kb_sA);
            else ka.template load<true>(A, p.lda, m0, p.M, k0, p.K, aoff);
            if constexpr (LAYOUT == XVA_GEMM_NT) kb.template load<true>(B, p.ldb, n0, p.N, k0, p.K, 0);
            else ib.template load<true>(bbase, p.ldb, ib_col, k0, p.K, LAYOUT == XVA_GEMM_TN ? p.kb_len : 0, p.kb_sB);
        }
    };
    const bool a_act = p.a_lrelu != 0, b_act = p.b_lrelu != 0;

    if (kt_begin < kt_end) gload(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (a_act) { if constexpr (LAYOUT == XVA_GEMM_TN) ia.template store<true>(As, p.a_slope); else ka.template store<true>(As, p.a_slope); }
        else       { if constexpr (LAYOUT == XVA_GEMM_TN) ia.template store<false>(As, 1.f); else ka.template store<false>(As, 1.f); }
        if (b_act) { if constexpr (LAYOUT == XVA_GEMM_NT) kb.template store<true>(Bs, p.b_slope); else ib.template store<true>(Bs, p.b_slope); }
        else       { if constexpr (LAYOUT == XVA_GEMM_NT) kb.template store<false>(Bs, 1.f); else ib.template store<false>(Bs, 1.f); }
        __syncthreads();
        if (kt + 1 < kt_end) gload(kt + 1);

        const LT* Aw = As + (wm * 64 + (lane & 15)) * LD;
        const LT* Bw = Bs + (wn * (BN / 2) + (lane & 15)) * LD;
        if constexpr (MODE == 3) {
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                bf16x8 ah[4], al[4], bh[NTN], bl[NTN];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ah[i] = *reinterpret_cast<const bf16x8*>(Aw + i * 16 * LD + kh * 32 + (lane >> 4) * 8);
                    al[i] = *reinterpret_cast<const bf16x8*>(Aw + BM * LD + i * 16 * LD + kh * 32 + (lane >> 4) * 8);
                }
#pragma unroll
                for (int j = 0; j < NTN; ++j) {
                    bh[j] = *reinterpret_cast<const bf16x8*>(Bw + j * 16 * LD + kh * 32 + (lane >> 4) * 8);
                    bl[j] = *reinterpret_cast<const bf16x8*>(Bw + BN * LD + j * 16 * LD + kh * 32 + (lane >> 4) * 8);
                }
                // term-major: the three MFMAs of one accumulator are 4 * NTN instructions apart (small terms first)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
        } else if constexpr (C::BF) {
#pragma unroll
            for (int kh = 0; kh < KH; ++kh) {
                bf16x8 af[4], bfr[NTN];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Aw + i * 16 * LD + kh * 32 + (lane >> 4) * 8);
#pragma unroll
                for (int j = 0; j < NTN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(Bw + j * 16 * LD + kh * 32 + (lane >> 4) * 8);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int s = 0; s < BK / 4; ++s) {
                float af[4], bfr[NTN];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = Aw[i * 16 * LD + s * 4 + (lane >> 4)];
#pragma unroll
                for (int j = 0; j < NTN; ++j) bfr[j] = Bw[j * 16 * LD + s * 4 + (lane >> 4)];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < NTN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    const int64_t coff = (int64_t)z1 * p.sC + (int64_t)z2 * p.sC2;
    const int64_t roff = (int64_t)z1 * p.sR + (int64_t)z2 * p.sR2;
    const int64_t goff = (int64_t)z1 * p.sG + (int64_t)z2 * p.sG2;
    const bool first_split = (ks == 0);
    // runtime loop over the four 16-row blocks (the fully unrolled epilogue was most of this kernel's code); the accumulators keep
    // compile-time indices: block i is selected by compares
#pragma unroll 1
    for (int i = 0; i < 4; ++i) {
        f32x4 ai[NTN];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (i == g) {
#pragma unroll
                for (int j = 0; j < NTN; ++j) ai[j] = acc[g][j];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
            if (row >= p.M) continue;
            bool live = true;
            if (p.mask_mode != XVA_MASK_NONE) {
                const int64_t rr = (int64_t)row * p.mask_mul + p.mask_add;
                const int t = (int)(rr % p.Tp);
                live = t >= p.mask_pad && t < p.mask_pad + p.mask_len;
                if (live && p.mask_mode == XVA_MASK_LEN) live = (t - p.mask_pad) < p.lens[rr / p.Tp];
            }
#pragma unroll
            for (int j = 0; j < NTN; ++j) {
                const int col = n0 + wn * (BN / 2) + j * 16 + (lane & 15);
                if (col >= p.N) continue;
                float v = ai[j][r];
                if ((p.splitk == 1 || first_split) && p.bias) v += p.bias[(int64_t)z2 * p.sbias2 + col];
                v *= p.alpha;
                if (p.drop_p > 0.f) v *= xva_dropout_scale(p.drop_p, p.drop_seed, p.drop_stream, (uint64_t)row * p.N + col);
                if (p.G) {
                    const float gv = ld_elem(p.G, goff + (int64_t)row * p.ldg + col, p.g_dtype);
                    if (p.F) { const float df = gv - ld_elem(p.F, goff + (int64_t)row * p.ldg + col, p.g_dtype); v += df > 0.f ? p.fm_c : (df < 0.f ? -p.fm_c : 0.f); }
                    v = gv > 0.f ? v : v * p.gate_slope;
                }
                if ((p.splitk == 1 || first_split) && p.R) v += p.beta * ld_elem(p.R, roff + (int64_t)row * p.ldr + col, p.r_dtype);
                switch (p.act) {
                    case XVA_ACT_RELU: v = fmaxf(v, 0.f); break;
                    case XVA_ACT_LRELU: v = lrelu(v, p.act_slope); break;
                    case XVA_ACT_TANH: v = tanhf(v); break;
                    case XVA_ACT_LOGCLAMP: v = logf(fmaxf(v, p.act_slope)); break;
                    default: break;
                }
                if (!live) v = 0.f;
                const int64_t ci = coff + (p.c_trans ? (int64_t)col * p.ldc + row : (int64_t)row * p.ldc + col);
                if (p.c_dtype == XVA_BF16) {
                    uint16_t* dst = reinterpret_cast<uint16_t*>(p.C) + ci;
                    if (p.accumulate) v += bf2f(*dst);
                    *dst = f2bf(v);
                } else {
                    float* dst = reinterpret_cast<float*>(p.C) + ci;
                    if (p.splitk > 1 || p.accumulate == 2) atomicAdd(dst, v);
                    else if (p.accumulate) *dst += v;
                    else *dst = v;
                }
            }
        }
    }
}

template <int LAYOUT, int MODE>
inline void launch_bn(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st) {
    if (bn == 32) hipLaunchKernelGGL((xva_gemm_kernel<LAYOUT, MODE, 32>), dim3(nblocks), dim3(NTHREADS), 0, st, p);
    else if (bn == 64) hipLaunchKernelGGL((xva_gemm_kernel<LAYOUT, MODE, 64>), dim3(nblocks), dim3(NTHREADS), 0, st, p);
    else hipLaunchKernelGGL((xva_gemm_kernel<LAYOUT, MODE, 128>), dim3(nblocks), dim3(NTHREADS), 0, st, p);
}
template <int MODE>
inline void launch_mode(const xva_gemm_params& p, int bn, unsigned nblocks, hipStream_t st) {
    switch (p.layout) {
        case XVA_GEMM_NT: launch_bn<XVA_GEMM_NT, MODE>(p, bn, nblocks, st); break;
        case XVA_GEMM_NN: launch_bn<XVA_GEMM_NN, MODE>(p, bn, nblocks, st); break;
        default: launch_bn<XVA_GEMM_TN, MODE>(p, bn, nblocks, st); break;
    }
}

}  // namespace xva_gemm_impl
