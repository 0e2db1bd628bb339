// gemm.hip — LDS-tiled MFMA GEMM for gfx950 (MI355X).
//
// Carries every dense contraction of the FastPitch path (reference call sites:
// fastpitch/transformer.py:59-77 conv-k3 FFN, :100-152 qkv/o_net/bmm, model.py:103-122
// predictors, :261 proj) plus the DFT and mel-filterbank products of the mel front end
// (common/stft.py:86-103, common/layers.py:135).
//
// Tile: 128x128x32 per 256-thread workgroup = 4 wave64s in a 2x2 grid, each wave owning a
// 64x64 sub-tile = 4x4 MFMA 16x16 accumulators (64 fp32 acc regs/lane).  Operands are
// staged HBM -> VGPR (float4, coalesced along the contiguous dimension of the operand)
// -> LDS as [row][k] (k contiguous) so that one ds_read_b128 (bf16) / ds_read_b32 (fp32)
// yields an MFMA fragment.  The HBM loads of K-tile t+1 are issued before the MFMAs of
// K-tile t, so HBM latency hides under the matrix pipe without a second LDS buffer.
//
// compute = bf16 : operands rounded to bf16 while staged, v_mfma_f32_16x16x32_bf16.
// compute = fp32 : exact fp32, v_mfma_f32_16x16x4_f32 (k-ordered fmaf chain; parity mode).
#include "xva_common.h"
#include "../../include/xva_gemm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define BM 128
#define BN 128
#define BK 32
#define NTHREADS 256

template <bool BF16> struct Lds;
template <> struct Lds<true> {
    typedef __bf16 T;
    static constexpr int LD = BK + 8;  // 80-byte rows: 16-B aligned fragment reads
};
template <> struct Lds<false> {
    typedef float T;
    static constexpr int LD = BK + 4;  // 144-byte rows: 16-B aligned float4 stores
};

__device__ __forceinline__ float4 ld4_pred(const float* p, int nvalid) {
    // nvalid: number of in-range elements starting at p (<=0 -> none)
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid >= 4) {
        v = *reinterpret_cast<const float4*>(p);
    } else if (nvalid > 0) {
        v.x = p[0];
        if (nvalid > 1) v.y = p[1];
        if (nvalid > 2) v.z = p[2];
    }
    return v;
}

template <bool BF16>
__device__ __forceinline__ void st4(typename Lds<BF16>::T* dst, float a, float b, float c, float d) {
    if constexpr (BF16) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 v = {a, b, c, d};
        bf16x4 r = __builtin_convertvector(v, bf16x4);
        *reinterpret_cast<bf16x4*>(dst) = r;
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(a, b, c, d);
    }
}

// ---- operand stagers -----------------------------------------------------------------
// KC: global X[i][k] (k contiguous, row stride ld). thread -> (kv = t&7, r = t>>3), rows r+32j.
struct RegTile { float4 v[4]; };

__device__ __forceinline__ void load_kc(RegTile& rt, const float* __restrict__ X, int64_t ld, int i0, int Ibound,
                                        int k0, int Kbound) {
    int t = threadIdx.x;
    int kv = t & 7, r = t >> 3;
    int k = k0 + kv * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int i = i0 + r + 32 * j;
        int nv = (i < Ibound) ? (Kbound - k) : 0;
        rt.v[j] = ld4_pred(X + (int64_t)i * ld + k, nv);
    }
}
template <bool BF16>
__device__ __forceinline__ void store_kc(const RegTile& rt, typename Lds<BF16>::T* Xs) {
    int t = threadIdx.x;
    int kv = t & 7, r = t >> 3;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        st4<BF16>(Xs + (r + 32 * j) * Lds<BF16>::LD + kv * 4, rt.v[j].x, rt.v[j].y, rt.v[j].z, rt.v[j].w);
}

// IC: global X[k][i] (i contiguous).  thread -> (iv = t&31, kg = t>>5); rows k0+4kg+j, cols i0+4iv..+3.
// The 4x4 micro-tile is transposed in registers and written as 4 k-contiguous quads.
struct SegMap {
    const float* base; int64_t ld; int seglen; int64_t seg0, segstride;
    __device__ __forceinline__ const float* row(int k) const {
        if (seglen > 0) { int s = k / seglen; return base + seg0 + (int64_t)s * segstride + (int64_t)(k - s * seglen) * ld; }
        return base + (int64_t)k * ld;
    }
};
__device__ __forceinline__ void load_ic(RegTile& rt, const SegMap& X, int i0, int Ibound, int k0, int Kbound) {
    int t = threadIdx.x;
    int iv = t & 31, kg = t >> 5;
    int i = i0 + iv * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int k = k0 + kg * 4 + j;
        int nv = (k < Kbound) ? (Ibound - i) : 0;
        rt.v[j] = ld4_pred(X.row(k < Kbound ? k : 0) + i, nv);
    }
}
template <bool BF16>
__device__ __forceinline__ void store_ic(const RegTile& rt, typename Lds<BF16>::T* Xs) {
    int t = threadIdx.x;
    int iv = t & 31, kg = t >> 5;
    typename Lds<BF16>::T* d = Xs + (iv * 4) * Lds<BF16>::LD + kg * 4;
    st4<BF16>(d + 0 * Lds<BF16>::LD, rt.v[0].x, rt.v[1].x, rt.v[2].x, rt.v[3].x);
    st4<BF16>(d + 1 * Lds<BF16>::LD, rt.v[0].y, rt.v[1].y, rt.v[2].y, rt.v[3].y);
    st4<BF16>(d + 2 * Lds<BF16>::LD, rt.v[0].z, rt.v[1].z, rt.v[2].z, rt.v[3].z);
    st4<BF16>(d + 3 * Lds<BF16>::LD, rt.v[0].w, rt.v[1].w, rt.v[2].w, rt.v[3].w);
}

template <int LAYOUT, bool BF16>
__global__ __launch_bounds__(NTHREADS) void xva_gemm_kernel(xva_gemm_params p) {
    typedef typename Lds<BF16>::T LT;
    constexpr int LD = Lds<BF16>::LD;
    __shared__ __attribute__((aligned(16))) LT As[BM * LD];
    __shared__ __attribute__((aligned(16))) LT Bs[BN * LD];

    const int z = blockIdx.z;
    const int bz = z / p.splitk, ks = z - bz * p.splitk;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const float* A = p.A + (int64_t)bz * p.sA;
    const float* B = p.B + (int64_t)bz * p.sB;

    const int nkt_total = (p.K + BK - 1) / BK;
    const int per = (nkt_total + p.splitk - 1) / p.splitk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt_total, kt_begin + per);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    SegMap amap{A, p.lda, 0, 0, 0};
    SegMap bmap{B, p.ldb, LAYOUT == XVA_GEMM_NN ? p.seglen : 0, p.seg0, p.segstride};

    RegTile ra, rb;
    auto gload = [&](int kt) {
        int k0 = kt * BK;
        if constexpr (LAYOUT == XVA_GEMM_TN) load_ic(ra, amap, m0, p.M, k0, p.K);
        else load_kc(ra, A, p.lda, m0, p.M, k0, p.K);
        if constexpr (LAYOUT == XVA_GEMM_NT) load_kc(rb, B, p.ldb, n0, p.N, k0, p.K);
        else load_ic(rb, bmap, n0, p.N, k0, p.K);
    };

    if (kt_begin < kt_end) gload(kt_begin);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if constexpr (LAYOUT == XVA_GEMM_TN) store_ic<BF16>(ra, As); else store_kc<BF16>(ra, As);
        if constexpr (LAYOUT == XVA_GEMM_NT) store_kc<BF16>(rb, Bs); else store_ic<BF16>(rb, Bs);
        __syncthreads();
        if (kt + 1 < kt_end) gload(kt + 1);

        const LT* Aw = As + (wm * 64 + (lane & 15)) * LD;
        const LT* Bw = Bs + (wn * 64 + (lane & 15)) * LD;
        if constexpr (BF16) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Aw + i * 16 * LD + (lane >> 4) * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(Bw + j * 16 * LD + (lane >> 4) * 8);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int s = 0; s < BK / 4; ++s) {
                float af[4], bfr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = Aw[i * 16 * LD + s * 4 + (lane >> 4)];
#pragma unroll
                for (int j = 0; j < 4; ++j) bfr[j] = Bw[j * 16 * LD + s * 4 + (lane >> 4)];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    float* C = p.C + (int64_t)bz * p.sC;
    const float* R = p.R ? p.R + (int64_t)bz * p.sR : nullptr;
    const float* G = p.G ? p.G + (int64_t)bz * p.sG : nullptr;
    const bool first_split = (ks == 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int row = m0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
            if (row >= p.M) continue;
            bool live = xva_row_live(p.mask_mode, p.lens, p.Tp, row);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int col = n0 + wn * 64 + j * 16 + (lane & 15);
                if (col >= p.N) continue;
                float v = acc[i][j][r] * p.alpha;
                if (p.splitk == 1 || first_split) {
                    if (p.bias) v += p.bias[col];
                    if (R) v += R[(int64_t)row * p.ldr + col];
                }
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.log_clamp > 0.f) v = logf(fmaxf(v, p.log_clamp));
                if (G) v = (G[(int64_t)row * p.ldg + col] > 0.f) ? v : 0.f;
                if (!live) v = 0.f;
                float* dst = C + (int64_t)row * p.ldc + col;
                if (p.splitk > 1) atomicAdd(dst, v);
                else if (p.accumulate) *dst += v;
                else *dst = v;
            }
        }
    }
}

bool xva_prof_is_on();
void xva_prof_begin(hipStream_t st, double flops, int variant);
void xva_prof_end(hipStream_t st);

template <int LAYOUT>
static int launch_layout(const xva_gemm_params& p, dim3 grid, hipStream_t st) {
    const bool prof = xva_prof_is_on();
    if (prof) xva_prof_begin(st, 2.0 * p.M * (double)p.N * p.K * p.batch, LAYOUT * 2 + (p.compute ? 1 : 0));
    struct End { bool on; hipStream_t s; ~End() { if (on) xva_prof_end(s); } } end_{prof, st};
    if (p.compute) hipLaunchKernelGGL((xva_gemm_kernel<LAYOUT, true>), grid, dim3(NTHREADS), 0, st, p);
    else hipLaunchKernelGGL((xva_gemm_kernel<LAYOUT, false>), grid, dim3(NTHREADS), 0, st, p);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

extern "C" int xva_gemm(const xva_gemm_params* pp, void* stream) {
    XVA_CHECK_ARG(pp != nullptr, "xva_gemm: null params");
    xva_gemm_params p = *pp;
    XVA_CHECK_ARG(p.A && p.B && p.C, "xva_gemm: null operand");
    XVA_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0, "xva_gemm: negative dim");
    if (p.M == 0 || p.N == 0) return XVA_OK;
    if (p.batch < 1) p.batch = 1;
    if (p.splitk < 1) p.splitk = 1;
    XVA_CHECK_ARG(p.lda % 4 == 0 && p.ldb % 4 == 0, "xva_gemm: lda/ldb must be multiples of 4 (lda=%ld ldb=%ld)",
                  (long)p.lda, (long)p.ldb);
    XVA_CHECK_ARG(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.B % 16) == 0, "xva_gemm: operands must be 16-byte aligned");
    XVA_CHECK_ARG(p.sA % 4 == 0 && p.sB % 4 == 0, "xva_gemm: batch strides must be multiples of 4");
    XVA_CHECK_ARG(p.seg0 % 4 == 0 && p.segstride % 4 == 0, "xva_gemm: segment offsets must be multiples of 4");
    XVA_CHECK_ARG(p.splitk == 1 || p.accumulate, "xva_gemm: splitk > 1 requires accumulate");
    XVA_CHECK_ARG(p.splitk == 1 || (!p.relu && !p.G && p.mask_mode == XVA_MASK_NONE),
                  "xva_gemm: splitk > 1 cannot carry a non-linear epilogue");
    XVA_CHECK_ARG(p.mask_mode == XVA_MASK_NONE || (p.batch == 1 && p.Tp >= 3 && (p.mask_mode == XVA_MASK_PAD || p.lens)),
                  "xva_gemm: bad row-mask arguments");
    XVA_CHECK_ARG(p.layout >= 0 && p.layout <= 2, "xva_gemm: bad layout");
    if (p.K == 0) p.splitk = 1;
    int nkt = (p.K + BK - 1) / BK;
    if (p.splitk > nkt && nkt > 0) p.splitk = nkt;
    dim3 grid(xva_cdiv(p.N, BN), xva_cdiv(p.M, BM), p.batch * p.splitk);
    XVA_CHECK_ARG(grid.y <= 65535 && grid.z <= 65535, "xva_gemm: grid too large");
    hipStream_t st = (hipStream_t)stream;
    switch (p.layout) {
        case XVA_GEMM_NT: return launch_layout<XVA_GEMM_NT>(p, grid, st);
        case XVA_GEMM_NN: return launch_layout<XVA_GEMM_NN>(p, grid, st);
        default: return launch_layout<XVA_GEMM_TN>(p, grid, st);
    }
}
