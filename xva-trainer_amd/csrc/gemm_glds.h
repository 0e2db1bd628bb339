// gemm_glds.h — second-generation main loop of the implicit-convolution GEMM for bf16-STORED operands (gfx950).
//
// What changes against gemm_core.h (which stays the general path: fp32 / mixed storage, fused LeakyReLU staging, ragged
// segment lengths):
//   * operand tiles travel HBM -> LDS directly (global_load_lds_dwordx4, 16 B per lane): no staging VGPRs, no ds_write pass;
//     the LDS image is lane-linear per wave instruction, so the bank swizzle is applied to the per-lane SOURCE address;
//   * operands whose contiguous dimension is not k (NN's B, TN's A and B) keep their natural [k][idx] image and the MFMA
//     fragments are gathered by the LDS transpose read (ds_read_b64_tr_b16); k-rows are stored with bits 2/3 of k swapped
//     and their 32-byte windows XOR-ed with the row position, which makes both the DMA writes and the transpose reads
//     conflict-free (measured: 3.2 cycles per wave read vs 16 for the plain image);
//   * K tile 64, two LDS buffers, ONE barrier per K tile: the DMA of tile t+1 is in flight while tile t feeds the MFMAs;
//   * tiles 256x256 (8 waves, 128x64 each, 128 KiB LDS, one workgroup per CU) and 128x128 (4 waves, 64x64 each);
//   * the MFMA operands are swapped (D = B-frag x A-frag) so each lane ends up with 4 CONSECUTIVE COLUMNS of one output row:
//     the epilogue loads residual / gate and stores C with 8- or 16-byte vectors.
// M / N edges clamp onto valid memory (their products reach only never-stored rows / columns); k >= K lanes read a zero page.
#pragma once
#include <type_traits>
#include "gemm_core.h"

namespace xva_glds {
using xva_gemm_impl::bf16x8;
using xva_gemm_impl::bf2f;
using xva_gemm_impl::f2bf;
using xva_gemm_impl::f32x4;
using xva_gemm_impl::ld_elem;
using xva_gemm_impl::lrelu;
using xva_gemm_impl::pack_bf2;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// ---- the two 16-bit formats ------------------------------------------------------------------------------------------------------
// Every kernel of this header exists in two flavours: bf16 (F16 = false: v_mfma_f32_16x16x32_bf16) and IEEE half (F16 = true:
// v_mfma_f32_16x16x32_f16 — the same rate, the same LDS images and DMA; 11 mantissa bits instead of 8).  All 16-bit tensors of one
// problem (A, B and whichever of C / R / G / F are not fp32) share the flavour; the LDS bytes are format-agnostic, so only the MFMA
// opcode and the fp32 <-> 16-bit conversions of the epilogue differ.
__host__ __device__ __forceinline__ bool is16(int dtype) { return dtype != XVA_F32; }
template <bool F16> __device__ __forceinline__ void unpack2(uint32_t w, float& a, float& b) {
    if constexpr (F16) { const f16x2 h = __builtin_bit_cast(f16x2, w); a = (float)h[0]; b = (float)h[1]; }
    else { a = __uint_as_float(w << 16); b = __uint_as_float(w & 0xffff0000u); }
}
template <bool F16> __device__ __forceinline__ uint32_t pack2(float a, float b) {      // round-to-nearest-even pair
    if constexpr (F16) { const f16x2 h = {(_Float16)a, (_Float16)b}; return __builtin_bit_cast(uint32_t, h); }
    else return pack_bf2(a, b);
}
template <bool F16> __device__ __forceinline__ float ld16(const void* p, int64_t idx) {
    const uint16_t u = reinterpret_cast<const uint16_t*>(p)[idx];
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, u); else return bf2f(u);
}
template <bool F16> __device__ __forceinline__ uint16_t cvt16(float v) {
    if constexpr (F16) return __builtin_bit_cast(uint16_t, (_Float16)v); else return f2bf(v);
}
template <bool F16> __device__ __forceinline__ float ld_any(const void* p, int64_t idx, int dtype) {
    return is16(dtype) ? ld16<F16>(p, idx) : reinterpret_cast<const float*>(p)[idx];
}
template <bool F16> __device__ __forceinline__ f32x4 mma16(bf16x8 x, bf16x8 y, f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0);
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define XVA_LDS __attribute__((address_space(3)))
#define XVA_GLB __attribute__((address_space(1)))

constexpr int GK = 64;            // K tile
enum { KC = 0, IC = 1 };          // operand kinds: k-contiguous rows / index-contiguous k-rows

static __device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];

// phase timestamps of every workgroup (tools/glds_timing.hip builds this header with XVA_GLDS_TIMING; the product never does)
#ifdef XVA_GLDS_TIMING
static __device__ unsigned long long* g_glds_timing;
#define XVA_T(i) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0) g_glds_timing[(size_t)blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define XVA_T(i) do { } while (0)
#endif
// ablation of the 256x256 K loop for tools/glds_timing.hip (bit 0: no MFMAs, 1: no DMA inside the loop, 2: no fragment reads inside the
// loop, 3: no epilogue); the product never defines it
#ifndef XVA_GLDS_ABLATE
#define XVA_GLDS_ABLATE 0
#endif

template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<N, I + 1>(f); }
}

__device__ __forceinline__ int swap23(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }
// position of 16-byte chunk c inside k-row `pos` of an IC image with ROWS index columns (an involution in c):
// rows of >= 256 bytes XOR the 32-byte window with pos & 7, 128-byte rows (ROWS = 64) with (pos >> 1) & 3 — either way the 8
// k-rows a half-wave transpose read touches land on 8 distinct 32-byte bank windows.
template <int ROWS>
__device__ __forceinline__ int ic_chunk(int pos, int c) {
    return ROWS >= 128 ? (c ^ ((pos & 7) << 1)) : (ROWS == 64 ? (c ^ (((pos >> 1) & 3) << 1)) : (c ^ (((pos >> 2) & 1) << 1)));
}

// ---- DMA descriptors of one operand tile ([ROWS idx] x [64 k]) for one thread ---------------------------------------------
// The tile is 8 * ROWS 16-byte chunks; wave instruction Q (0 .. ROWS/8 - 1) fills LDS bytes [Q * 1024, Q * 1024 + 1024).
template <int KIND, int ROWS, int NW>
struct Loader {
    static constexpr int NI = ROWS / (8 * NW);   // wave instructions per wave per tile
    int64_t off[NI];                              // element offset of this lane's chunk from the tile's scalar base
    int kk[NI];                                   // k index inside the tile of the chunk's first element

    // i0: first row / column of the tile; bound: number of valid rows / columns; ld: leading dimension (elements);
    // KIND == IC: column c is remapped to colmap(c) = cseg0 + c + (c / cseglen) * csegstride when cseglen > 0 (TN column segments)
    // KIND == KC: k segments (conv taps): chunk k-offset kc adds (kc / kseglen) * ksegadj (kseglen divides 64 or is a multiple of it)
    // KIND == IC: k-row kr lives at (kr / kseglen) * ksegadj + (kr % kseglen) * ld (NN row segments)
    __device__ __forceinline__ void init(int lane, int wave, int i0, int bound, int64_t ld, int cseglen, int64_t cseg0, int64_t csegstride,
                                         int kseglen = 0, int64_t ksegadj = 0) {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int Q = q * NW + wave;
            if constexpr (KIND == KC) {
                const int r = Q * 8 + (lane >> 3);
                const int c = (lane & 7) ^ (lane >> 3);             // chunk held by LDS position (lane & 7) of row r: c ^ (r & 7)
                off[q] = (int64_t)min(i0 + r, bound - 1) * ld + c * 8 + (kseglen > 0 ? (int64_t)((c * 8) / kseglen) * ksegadj : 0);
                kk[q] = c * 8;
            } else {
                constexpr int CPR = ROWS / 8;                       // chunks per k-row: 16 or 32
                constexpr int RPI = 64 / CPR;                       // k-rows per wave instruction: 4 or 2
                const int pos = Q * RPI + lane / CPR;
                const int p = lane % CPR;
                const int c = ic_chunk<ROWS>(pos, p);
                const int k = swap23(pos);
                int col = min(i0 + c * 8, bound - 8);
                int64_t cm = col;
                if (cseglen > 0) cm = cseg0 + col + (int64_t)(col / cseglen) * csegstride;
                off[q] = (kseglen > 0 ? (int64_t)(k / kseglen) * ksegadj + (int64_t)(k % kseglen) * ld : (int64_t)k * ld) + cm;
                kk[q] = k;
            }
        }
    }
    // base: operand pointer advanced to this K tile (scalar); k0: first k of the tile; K: reduction length
    __device__ __forceinline__ void issue(const uint16_t* base, int k0, int K, XVA_LDS uint8_t* tile, int wave) const {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const uint16_t* src = (k0 + kk[q] < K) ? base + off[q] : reinterpret_cast<const uint16_t*>(g_zero_page);
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)src, (XVA_LDS void*)(tile + (q * NW + wave) * 1024), 16, 0, 0);
        }
    }
    // the whole 64-deep tile lies inside K
    __device__ __forceinline__ void issue_full(const uint16_t* base, XVA_LDS uint8_t* tile, int wave) const {
#pragma unroll
        for (int q = 0; q < NI; ++q)
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)(base + off[q]), (XVA_LDS void*)(tile + (q * NW + wave) * 1024), 16, 0, 0);
    }
};

// ---- MFMA operand fragments -----------------------------------------------------------------------------------------------------
// KC operands are read by plain ds_read_b128 (the compiler tracks them).  IC operands come from LDS transpose reads, and those are issued
// as INLINE ASM: through the builtin the compiler cannot tell the read from the LDS bytes an in-flight global_load_lds is still writing
// and puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 — in the NN / TN K loops that drained the whole DMA ring once per K
// tile (found in the ISA of the first resident weight-gradient kernel, wgrad_res.h; the NT loops never had it).  The compiler then no
// longer tracks lgkmcnt for these reads: frags_wait() is the explicit wait (lgkmcnt(0) only — scalar loads share the counter and return
// out of order, so counted waits are not safe), and frag_value() ties the registers to it before anything may read them.
template <int KIND> struct Frag { bf16x8 v; };
template <> struct Frag<1> { s16x4 lo, hi; };
template <bool ANY_IC> __device__ __forceinline__ void frags_wait() { if constexpr (ANY_IC) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ bf16x8 frag_value(Frag<0>& f) { return f.v; }
__device__ __forceinline__ bf16x8 frag_value(Frag<1>& f) {
    asm volatile("" : "+v"(f.lo), "+v"(f.hi));                     // ordered after the preceding frags_wait
    s16x8 v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// ---- MFMA fragment readers ------------------------------------------------------------------------------------------------
// KC image: [ROWS][64 k] bf16, 128-byte rows, chunk c of row r at position c ^ (r & 7).
// lane -> row (l & 15) of the 16-row tile, k = kh * 32 + (l >> 4) * 8 + j
struct KcReader {
    uint32_t o0, o1;   // byte offsets of this lane inside a 16-row tile for kh = 0 / 1
    __device__ __forceinline__ void init(int lane) {
        o0 = (lane & 15) * 128 + (((lane >> 4) ^ (lane & 7)) * 16);
        o1 = o0 ^ 64;
    }
    __device__ __forceinline__ Frag<KC> read(const XVA_LDS uint8_t* tile, int row0, int kh) const {
        Frag<KC> f; f.v = *reinterpret_cast<const XVA_LDS bf16x8*>(tile + row0 * 128 + (kh ? o1 : o0)); return f;
    }
};
// IC image: [64 kpos][ROWS idx] bf16, kpos = k with bits 2/3 swapped, 16-byte chunk c of a row at position ic_chunk(kpos, c).
template <int ROWS, int TILES>
struct IcReader {
    uint32_t o[TILES];   // byte offset of this lane for idx tile t (16 columns), kh = 0, half 0
    __device__ __forceinline__ void init(int lane, int w0) {
        const int g = lane >> 4, i = lane & 15;
        const int posl = (g >> 1) * 16 + (g & 1) * 4 + (i >> 2);               // kpos without the (kh, half) bits (they do not enter the swizzle)
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            o[t] = posl * (ROWS * 2) + ic_chunk<ROWS>(posl, w0 / 8 + t * 2 + ((i >> 1) & 1)) * 16 + (i & 1) * 8;
    }
    // The two transpose reads of one MFMA operand, as inline asm: see Frag below
    __device__ __forceinline__ Frag<IC> read(const XVA_LDS uint8_t* tile, int t, int kh) const {
        const XVA_LDS uint8_t* a = tile + o[t] + kh * (32 * ROWS * 2);
        Frag<IC> f;
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"((uint32_t)(uintptr_t)a));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"((uint32_t)(uintptr_t)a), "n"(8 * ROWS * 2));
        return f;
    }
};

// LeakyReLU of an operand fragment (x -> x > 0 ? x : slope * x), the "activation fused into the consumer" of HiFi-GAN's
// generator (models.py:62-66,115-126): applied to the 8 bf16 values a lane feeds to one MFMA
template <bool F16 = false>
__device__ __forceinline__ bf16x8 lrelu_frag(bf16x8 f, float slope) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 w = __builtin_bit_cast(u32x4, f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (w[e] & 0x80008000u) {   // at least one negative value in the pair
            float lo, hi; unpack2<F16>(w[e], lo, hi);
            w[e] = pack2<F16>(lrelu(lo, slope), lrelu(hi, slope));
        }
    }
    return __builtin_bit_cast(bf16x8, w);
}

// output activation of N values: ONE uniform branch on p.act around the whole group (a switch per element costs a scalar branch
// chain per value: measured +45 us on the 27584 x 1536 FastPitch conv1 forward)
template <int N>
__device__ __forceinline__ void apply_act(const xva_gemm_params& p, float (&v)[N]) {
    if (p.act == XVA_ACT_NONE) return;
    if (p.act == XVA_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = fmaxf(v[e], 0.f);
    } else if (p.act == XVA_ACT_LRELU) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = lrelu(v[e], p.act_slope);
    } else if (p.act == XVA_ACT_TANH) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = tanhf(v[e]);
    } else if (p.act == XVA_ACT_LOGCLAMP) {
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = logf(fmaxf(v[e], p.act_slope));
    }
}

// ---- epilogue -------------------------------------------------------------------------------------------------------------
// v[0..3]: columns col .. col + 3 of row `row` (all inside N when VEC).  Order (include/xva_gemm.h):
// v = alpha * (acc + bias) ; dropout ; gate ; + beta * R ; act ; row mask ; store / accumulate.
template <bool VEC, bool F16 = false>
__device__ __forceinline__ void epilogue4(const xva_gemm_params& p, f32x4 a, int row, int col, bool live, bool lin_first, int z2,
                                          int64_t coff, int64_t roff, int64_t goff) {
    float v[4] = {a[0], a[1], a[2], a[3]};
    const int nv = VEC ? 4 : min(4, p.N - col);
    if (lin_first && p.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (e < nv) v[e] += p.bias[(int64_t)z2 * p.sbias2 + col + e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] *= p.alpha;
    if (p.drop_p > 0.f) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] *= xva_dropout_scale(p.drop_p, p.drop_seed, p.drop_stream, (uint64_t)row * p.N + col + e);
    }
    if (p.G) {
        float g[4] = {1.f, 1.f, 1.f, 1.f};
        const int64_t gi = goff + (int64_t)row * p.ldg + col;
        if (VEC) {
            if (is16(p.g_dtype)) {
                uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.G) + gi);
                unpack2<F16>(r.x, g[0], g[1]); unpack2<F16>(r.y, g[2], g[3]);
            } else {
                float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.G) + gi);
                g[0] = r.x; g[1] = r.y; g[2] = r.z; g[3] = r.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nv) g[e] = ld_any<F16>(p.G, gi + e, p.g_dtype);
        }
        if (p.F) {      // feature-matching term, before the gate: v += fm_c * sign(g - f)
            float f[4] = {0.f, 0.f, 0.f, 0.f};
            if (VEC) {
                if (is16(p.g_dtype)) {
                    uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.F) + gi);
                    unpack2<F16>(r.x, f[0], f[1]); unpack2<F16>(r.y, f[2], f[3]);
                } else {
                    float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.F) + gi);
                    f[0] = r.x; f[1] = r.y; f[2] = r.z; f[3] = r.w;
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) if (e < nv) f[e] = ld_any<F16>(p.F, gi + e, p.g_dtype);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float df = g[e] - f[e]; v[e] += df > 0.f ? p.fm_c : (df < 0.f ? -p.fm_c : 0.f); }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = g[e] > 0.f ? v[e] : v[e] * p.gate_slope;
    }
    if (lin_first && p.R) {
        const int64_t ri = roff + (int64_t)row * p.ldr + col;
        float r4[4] = {0.f, 0.f, 0.f, 0.f};
        if (VEC) {
            if (is16(p.r_dtype)) {
                uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.R) + ri);
                unpack2<F16>(r.x, r4[0], r4[1]); unpack2<F16>(r.y, r4[2], r4[3]);
            } else {
                float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.R) + ri);
                r4[0] = r.x; r4[1] = r.y; r4[2] = r.z; r4[3] = r.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (e < nv) r4[e] = ld_any<F16>(p.R, ri + e, p.r_dtype);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += p.beta * r4[e];
    }
    apply_act<4>(p, v);
    if (!live) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
    if (VEC && !p.c_trans) {
        const int64_t ci = coff + (int64_t)row * p.ldc + col;
        if (is16(p.c_dtype)) {
            uint2* dst = reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + ci);
            if (p.accumulate) {
                uint2 o = *dst;
                float o4[4]; unpack2<F16>(o.x, o4[0], o4[1]); unpack2<F16>(o.y, o4[2], o4[3]);
                v[0] += o4[0]; v[1] += o4[1]; v[2] += o4[2]; v[3] += o4[3];
            }
            const uint2 hi = make_uint2(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]));
            *dst = hi;
            if (p.c_plane) {      // the lo plane of a split-bf16 pair: bf16(v - hi)
                float h4[4]; unpack2<F16>(hi.x, h4[0], h4[1]); unpack2<F16>(hi.y, h4[2], h4[3]);
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C) + ci + p.c_plane) = make_uint2(pack2<F16>(v[0] - h4[0], v[1] - h4[1]), pack2<F16>(v[2] - h4[2], v[3] - h4[3]));
            }
            if (p.C2) *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.C2) + ci) =
                make_uint2(pack2<F16>(lrelu(v[0], p.c2_slope), lrelu(v[1], p.c2_slope)), pack2<F16>(lrelu(v[2], p.c2_slope), lrelu(v[3], p.c2_slope)));
        } else {
            float* dst = reinterpret_cast<float*>(p.C) + ci;
            if (p.splitk > 1 || p.accumulate == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(dst + e, v[e]);
            } else {
                float4* d4 = reinterpret_cast<float4*>(dst);
                if (p.accumulate) { float4 o = *d4; v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w; }
                *d4 = make_float4(v[0], v[1], v[2], v[3]);
                if (p.C2) *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C2) + ci) =
                    make_float4(lrelu(v[0], p.c2_slope), lrelu(v[1], p.c2_slope), lrelu(v[2], p.c2_slope), lrelu(v[3], p.c2_slope));
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e >= nv) break;
            const int64_t ci = coff + (p.c_trans ? (int64_t)(col + e) * p.ldc + row : (int64_t)row * p.ldc + col + e);
            if (is16(p.c_dtype)) {
                uint16_t* dst = reinterpret_cast<uint16_t*>(p.C) + ci;
                float x = v[e];
                if (p.accumulate) x += ld16<F16>(dst, 0);
                *dst = cvt16<F16>(x);
                if (p.c_plane) dst[p.c_plane] = cvt16<F16>(x - ld16<F16>(dst, 0));
                if (p.C2) reinterpret_cast<uint16_t*>(p.C2)[ci] = cvt16<F16>(lrelu(x, p.c2_slope));
            } else {
                float* dst = reinterpret_cast<float*>(p.C) + ci;
                if (p.splitk > 1 || p.accumulate == 2) atomicAdd(dst, v[e]);
                else {
                    float x = v[e];
                    if (p.accumulate) x += *dst;
                    *dst = x;
                    if (p.C2) reinterpret_cast<float*>(p.C2)[ci] = lrelu(x, p.c2_slope);
                }
            }
        }
    }
}

// ---- epilogue of one wave tile: acc[i][j][e] = C[rbase + i*16][cbase + j*16 + e]  (rbase includes lane & 15, cbase (lane >> 4) * 4)
// (compile-time indices: a rolled loop would index `acc` dynamically and push the accumulators to scratch)
template <int MI, int NJ, bool F16 = false>
__device__ __forceinline__ void tile_epilogue(const xva_gemm_params& p, f32x4 (&acc)[MI][NJ], int vec_epi, int rbase, int cbase, int z1, int z2,
                                              int bz, int ks) {
    const int64_t coff = (int64_t)z1 * p.sC + (int64_t)z2 * p.sC2;
    const int64_t roff = (int64_t)z1 * p.sR + (int64_t)z2 * p.sR2;
    const int64_t goff = (int64_t)z1 * p.sG + (int64_t)z2 * p.sG2;
    const bool lin_first = (p.splitk == 1) || (ks == 0);
    // runtime loops over the row blocks (code size: see tile_epilogue_rows); the accumulators keep compile-time indices behind a compare
    auto run = [&](auto vec_tag) {
        constexpr bool VEC = decltype(vec_tag)::value;
#pragma unroll 1
        for (int i = 0; i < MI; ++i) {
            f32x4 a[NJ];
            static_for<MI>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (i == g) static_for<NJ>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = acc[g][j]; });
            });
            const int row = rbase + i * 16;
            if (row < p.M) {
                bool live = true;
                if (p.mask_mode != XVA_MASK_NONE) {
                    const int64_t rr = (int64_t)row * p.mask_mul + p.mask_add;
                    const int t = (int)(rr % p.Tp);
                    live = t >= p.mask_pad && t < p.mask_pad + p.mask_len;
                    if (live && p.mask_mode == XVA_MASK_LEN) live = (t - p.mask_pad) < p.lens[rr / p.Tp];
                }
                static_for<NJ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int col = cbase + j * 16;
                    if (col < p.N) epilogue4<VEC, F16>(p, a[j], row, col, live, lin_first, z2, coff, roff, goff);
                });
            }
        }
    };
    if (p.splitk > 1 && p.sk_ws) {
        // split-K slab: raw partial sums of this K range, [split][M][N] fp32; xva_gemm_splitk_reduce applies the epilogue
        float* slab = reinterpret_cast<float*>(p.sk_ws) + ((int64_t)bz * p.splitk + ks) * (int64_t)p.M * p.N;
#pragma unroll 1
        for (int i = 0; i < MI; ++i) {
            f32x4 a[NJ];
            static_for<MI>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (i == g) static_for<NJ>([&](auto jc) { constexpr int j = decltype(jc)::value; a[j] = acc[g][j]; });
            });
            const int row = rbase + i * 16;
            if (row < p.M) {
                static_for<NJ>([&](auto jc) {
                    constexpr int j = decltype(jc)::value;
                    const int col = cbase + j * 16;
                    if (col < p.N) {
                        float* dst = slab + (int64_t)row * p.N + col;
                        if (vec_epi) *reinterpret_cast<float4*>(dst) = make_float4(a[j][0], a[j][1], a[j][2], a[j][3]);
                        else for (int e = 0; e < 4 && col + e < p.N; ++e) dst[e] = a[j][e];
                    }
                });
            }
        }
        return;
    }
    if (vec_epi) run(std::true_type{}); else run(std::false_type{});
}


// ---- row-contiguous epilogue (vec_epi == 2) ---------------------------------------------------------------------------------
// The MFMA result layout gives a lane 4 columns of a row, i.e. 8-byte (bf16) pieces: a wave store touches 16 rows x 32 bytes, and
// every residual / gate load sits behind the previous piece's store (C, R and G may alias, so the compiler keeps the order) —
// measured 17 us of a 25 us HiFi-GAN resblock workgroup.  Here each 16-row block of the wave tile is transposed through a private
// LDS scratch (pitch WN + 4 floats: conflict-free b128 writes and reads) so that a lane owns 8 CONSECUTIVE columns of a row:
// 16-byte bf16 loads / stores, full 128-byte row segments per wave, and all loads of a group of row blocks issued before its
// first store.  Same operation order as epilogue4.  Host-verified (vec_epi == 2): N % 8 == 0, 8-element row alignment of C / R / G.
// (residual / gate / accumulated-into tensors are bf16 here: 8 values = one 16-byte load; fp32 ones take the 4-column epilogue)
template <bool F16 = false>
__device__ __forceinline__ void unpack8(const uint4& raw, float (&o)[8]) {
    const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) unpack2<F16>(w[e], o[2 * e], o[2 * e + 1]);
}
__device__ __forceinline__ uint4 load8(const void* base, int64_t idx) { return *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(base) + idx); }

// order the wave's LDS scratch writes and reads for the COMPILER only: LDS instructions of one wave execute in order, so no wait is
// needed — and a fence would also wait for the global stores in flight (vmcnt), serialising every 16-row block behind the previous
// block's stores (measured: 21 us of a 50 us 256x256 workgroup)
__device__ __forceinline__ void wave_lds_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// EPI (round 6): which optional epilogue features are COMPILED IN — bit 0: residual / gate / accumulate-into-C reads, bit 1: dropout, bit 2: the rest (feature-matching
// term, second output, split-pair output, non-temporal stores).  The full epilogue is ~10 KB of code per pass and four passes per loop iteration: with everything
// compiled in, a 256 x 256 workgroup spent 8.8 us in it whatever the problem used (6.4 us with the global stores removed: tools/glds_timing.hip -DXVA_GLDS_ABLATE=16),
// 5.7 us with the unused paths compiled out.  The 256 x 256 / 384 x 128 kernels are instantiated for EPI 0, 1, 3 and 7 and the launcher picks the smallest that
// covers the problem (epi_variant).
template <int MI, int NJ, bool F16 = false, int EPI = 7>
__device__ __forceinline__ void tile_epilogue_rows(const xva_gemm_params& p, f32x4 (&acc)[MI][NJ], XVA_LDS float* scr, int r0, int c0, int lane,
                                                   int z1, int z2, int bz, int ks, int nt_store = 0) {
    constexpr int WN = NJ * 16, PITCH = WN + 4, LPR = WN / 8, RPP = 64 / LPR, NPASS = 16 / RPP;
    constexpr int CH = MI < 2 ? MI : 2;                                   // 16-row blocks whose loads are in flight together
    const int rr = lane / LPR, col = c0 + (lane % LPR) * 8;
    const bool col_ok = col < p.N;
    const bool slab = p.splitk > 1 && p.sk_ws;
    const int64_t coff = (int64_t)z1 * p.sC + (int64_t)z2 * p.sC2;
    const int64_t roff = (int64_t)z1 * p.sR + (int64_t)z2 * p.sR2;
    const int64_t goff = (int64_t)z1 * p.sG + (int64_t)z2 * p.sG2;
    float* slab_base = slab ? reinterpret_cast<float*>(p.sk_ws) + ((int64_t)bz * p.splitk + ks) * (int64_t)p.M * p.N : nullptr;
    float bias[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!slab && p.bias && col_ok) {
        const float4* bq = reinterpret_cast<const float4*>(p.bias + (int64_t)z2 * p.sbias2 + col);
        const float4 b0 = bq[0], b1 = bq[1];
        bias[0] = b0.x; bias[1] = b0.y; bias[2] = b0.z; bias[3] = b0.w; bias[4] = b1.x; bias[5] = b1.y; bias[6] = b1.z; bias[7] = b1.w;
    }
    constexpr bool E_RG = (EPI & 1) != 0, E_DROP = (EPI & 2) != 0, E_X = (EPI & 4) != 0;
    constexpr bool E_R32 = (EPI & 8) != 0;      // the residual is an fp32 tensor (fp16-operand / split-products FastPitch: the fp32 residual stream): two 16-byte loads per piece
    const bool want_r = E_RG && !slab && p.R, want_g = E_RG && !slab && p.G, want_c = E_RG && !slab && p.accumulate, want_f = E_X && want_g && p.F;
    const bool mask32 = (int64_t)p.M * p.mask_mul + p.mask_add < (1ll << 31) && p.mask_add >= 0 && p.mask_mul >= 0;   // mapped row indices fit 32 bits
    // The loop over groups of CH row blocks is a RUNTIME loop: unrolled, the epilogue of the 256x256 kernel alone was ~400 KB of
    // code (the instruction cache holds 64 KB) and took 21 us of a 50 us workgroup whatever the memory traffic.  Only the
    // accumulator -> LDS writes need compile-time register indices; they sit behind a compare per group.
#pragma unroll 1
    for (int ch = 0; ch < MI / CH; ++ch) {
        const int i0 = ch * CH;
        uint4 rraw[CH * NPASS], graw[CH * NPASS], craw[CH * NPASS], fraw[CH * NPASS], rraw2[E_R32 ? CH * NPASS : 1];
        static_for<CH * NPASS>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int row = r0 + (i0 + q / NPASS) * 16 + (q % NPASS) * RPP + rr;
            if (row < p.M && col_ok) {
                if (want_r) {
                    if constexpr (E_R32) {
                        const float* rp = reinterpret_cast<const float*>(p.R) + roff + (int64_t)row * p.ldr + col;
                        rraw[q] = *reinterpret_cast<const uint4*>(rp); rraw2[q] = *reinterpret_cast<const uint4*>(rp + 4);
                    } else rraw[q] = load8(p.R, roff + (int64_t)row * p.ldr + col);
                }
                if (want_g) graw[q] = load8(p.G, goff + (int64_t)row * p.ldg + col);
                if (want_f) fraw[q] = load8(p.F, goff + (int64_t)row * p.ldg + col);
                if (want_c) craw[q] = load8(p.C, coff + (int64_t)row * p.ldc + col);
            }
        });
        static_for<CH>([&](auto ic) {
            constexpr int ii = decltype(ic)::value;
            const int i = i0 + ii;
            wave_lds_order();
            static_for<MI / CH>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                if (ch == g) {
                    static_for<NJ>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        *reinterpret_cast<XVA_LDS f32x4*>(scr + (lane & 15) * PITCH + j * 16 + (lane >> 4) * 4) = acc[g * CH + ii][j];
                    });
                }
            });
            wave_lds_order();
            static_for<NPASS>([&](auto pc) {
                constexpr int ps = decltype(pc)::value, q = ii * NPASS + ps;
                const int row = r0 + i * 16 + ps * RPP + rr;
                const XVA_LDS f32x4* src = reinterpret_cast<const XVA_LDS f32x4*>(scr + (ps * RPP + rr) * PITCH + (lane % LPR) * 8);
                const f32x4 lo = src[0], hi = src[1];
                if (!(row < p.M && col_ok)) return;
                float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                if (slab) {
                    float4* dst = reinterpret_cast<float4*>(slab_base + (int64_t)row * p.N + col);
                    dst[0] = make_float4(v[0], v[1], v[2], v[3]); dst[1] = make_float4(v[4], v[5], v[6], v[7]);
                    return;
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] + bias[e]) * p.alpha;
                if (E_DROP && p.drop_p > 0.f) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] *= xva_dropout_scale(p.drop_p, p.drop_seed, p.drop_stream, (uint64_t)row * p.N + col + e);
                }
                if (want_g) {
                    float g[8]; unpack8<F16>(graw[q], g);
                    if (want_f) {      // feature-matching term, before the gate
                        float f[8]; unpack8<F16>(fraw[q], f);
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float df = g[e] - f[e]; v[e] += df > 0.f ? p.fm_c : (df < 0.f ? -p.fm_c : 0.f); }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = g[e] > 0.f ? v[e] : v[e] * p.gate_slope;
                }
                if (want_r) {
                    float r8[8];
                    if constexpr (E_R32) {
                        r8[0] = __uint_as_float(rraw[q].x); r8[1] = __uint_as_float(rraw[q].y); r8[2] = __uint_as_float(rraw[q].z); r8[3] = __uint_as_float(rraw[q].w);
                        r8[4] = __uint_as_float(rraw2[q].x); r8[5] = __uint_as_float(rraw2[q].y); r8[6] = __uint_as_float(rraw2[q].z); r8[7] = __uint_as_float(rraw2[q].w);
                    } else unpack8<F16>(rraw[q], r8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += p.beta * r8[e];
                }
                apply_act<8>(p, v);
                if (p.mask_mode != XVA_MASK_NONE) {
                    const int64_t rm = (int64_t)row * p.mask_mul + p.mask_add;
                    int t, item;
                    if (mask32) { item = (int)((uint32_t)rm / (uint32_t)p.Tp); t = (int)((uint32_t)rm - (uint32_t)item * (uint32_t)p.Tp); }   // 32-bit division: ~5x fewer instructions
                    else { item = (int)(rm / p.Tp); t = (int)(rm - (int64_t)item * p.Tp); }
                    bool live = t >= p.mask_pad && t < p.mask_pad + p.mask_len;
                    if (live && p.mask_mode == XVA_MASK_LEN) live = (t - p.mask_pad) < p.lens[item];
                    if (!live) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] = 0.f;
                    }
                }
                if (want_c) {
                    float o[8]; unpack8<F16>(craw[q], o);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += o[e];
                }
                const int64_t ci = coff + (int64_t)row * p.ldc + col;
                if (is16(p.c_dtype)) {
                    if (E_X && nt_store) {      // a streamed output (larger than the L2s): do not evict the operand panels the next rounds re-read
                        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                        const u32x4 pk = {pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]), pack2<F16>(v[6], v[7])};
                        __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(reinterpret_cast<uint16_t*>(p.C) + ci));
                    } else {
                        const uint4 hi = make_uint4(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]), pack2<F16>(v[6], v[7]));
                        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + ci) = hi;
                        if (E_X && p.c_plane) {      // the lo plane of a split-bf16 pair
                            float hv[8]; unpack8<F16>(hi, hv);
                            *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C) + ci + p.c_plane) =
                                make_uint4(pack2<F16>(v[0] - hv[0], v[1] - hv[1]), pack2<F16>(v[2] - hv[2], v[3] - hv[3]), pack2<F16>(v[4] - hv[4], v[5] - hv[5]),
                                           pack2<F16>(v[6] - hv[6], v[7] - hv[7]));
                        }
                    }
                } else {
                    float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C) + ci);
                    dst[0] = make_float4(v[0], v[1], v[2], v[3]); dst[1] = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (E_X && p.C2) {   // the activated copy next to the raw one
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = lrelu(v[e], p.c2_slope);
                    if (is16(p.c_dtype)) {
                        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.C2) + ci) =
                            make_uint4(pack2<F16>(v[0], v[1]), pack2<F16>(v[2], v[3]), pack2<F16>(v[4], v[5]), pack2<F16>(v[6], v[7]));
                    } else {
                        float4* dst = reinterpret_cast<float4*>(reinterpret_cast<float*>(p.C2) + ci);
                        dst[0] = make_float4(v[0], v[1], v[2], v[3]); dst[1] = make_float4(v[4], v[5], v[6], v[7]);
                    }
                }
            });
        });
    }
}
// can the row-contiguous epilogue serve this problem? (no transposed / atomic stores)
__host__ __device__ __forceinline__ bool rows_epilogue_ok(const xva_gemm_params& p, int vec_epi) {
    if (vec_epi != 2) return false;
    const bool slab = p.splitk > 1 && p.sk_ws;
    if (slab) return true;                       // raw partial sums [split][M][N]: C's own layout (c_trans, strides) is the reduce kernel's business
    if (p.c_trans) return false;
    if (!slab && p.c_dtype == XVA_F32 && (p.splitk > 1 || p.accumulate == 2)) return false;
    if (!slab && p.splitk > 1) return false;
    if (!slab && ((p.R && !is16(p.r_dtype)) || (p.G && !is16(p.g_dtype)) || (p.accumulate && !is16(p.c_dtype)))) return false;
    return true;
}
constexpr int epi_scratch_bytes(int WN) { return 16 * (WN + 4) * 4; }
// the smallest compiled epilogue variant (EPI of tile_epilogue_rows) that covers this launch; 7 = everything, incl. the 4-column fallback epilogue
inline int epi_variant(const xva_gemm_params& p, int vec_flags, bool r32_variants = false) {
    const int vec_epi = vec_flags & 15;
    if (r32_variants && p.R && p.r_dtype == XVA_F32 && vec_epi == 2 && !p.c_trans && !(p.splitk > 1) && p.accumulate == 0 && (!p.G || is16(p.g_dtype)) &&
        !p.F && !p.C2 && !p.c_plane && !(vec_flags >> 4) && p.ldr % 4 == 0 && p.sR % 4 == 0 && p.sR2 % 4 == 0 && ((uintptr_t)p.R % 16) == 0)
        return (p.drop_p > 0.f ? 3 : 1) | 8;        // the row-contiguous epilogue with an fp32 residual (variants instantiated for the fp16 flavour)
    if (!rows_epilogue_ok(p, vec_epi)) return 7;
    if (p.splitk > 1 && p.sk_ws) return 0;                              // raw partial sums into the slabs: no epilogue features at all
    if (p.F || p.C2 || p.c_plane || (vec_flags >> 4)) return 7;
    const bool rg = p.R || p.G || p.accumulate;
    if (p.drop_p > 0.f) return 3;
    return rg ? 1 : 0;
}

// ---- the kernel -----------------------------------------------------------------------------------------------------------
// vec_epi: 1 = host-verified that N % 4 == 0 and C / R / G rows are 4-element aligned (vector epilogue allowed); 2 = 8-element
// granularity as well (row-contiguous epilogue through LDS)
template <int LAYOUT, int BM, int BN, int WM, int WN, bool F16 = false, int EPI = 7>
__global__ __launch_bounds__((BM / WM) * (BN / WN) * 64, (2 * (BM + BN) * GK * 2 > 80 * 1024) ? 1 : (BM * BN >= 128 * 128 ? 2 : 3)) void xva_gemm_glds_kernel(xva_gemm_params p, int vec_flags) {
    const int vec_epi = vec_flags & 15, nt_store = vec_flags >> 4;     // bit 4: non-temporal C stores (gemm_glds.hip)
    constexpr int NWN = BN / WN, NW = (BM / WM) * NWN;
    constexpr int MI = WM / 16, NJ = WN / 16;
    constexpr int AK = LAYOUT == XVA_GEMM_TN ? IC : KC;
    constexpr int BKD = LAYOUT == XVA_GEMM_NT ? KC : IC;
    constexpr int A_BYTES = BM * GK * 2, B_BYTES = BN * GK * 2, BUF = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;

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
    const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (int64_t)z1 * p.sA + (int64_t)z2 * p.sA2;
    const uint16_t* B = reinterpret_cast<const uint16_t*>(p.B) + (int64_t)z1 * p.sB + (int64_t)z2 * p.sB2;

    const int tpb = p.kb_len > 0 ? (p.kb_len + GK - 1) / GK : 1;                                  // K tiles per K block (TN)
    const int nkt1 = p.kb_len > 0 ? (p.K / p.kb_len) * tpb : (p.K + GK - 1) / GK;
    const int nkt_total = p.planes ? 3 * nkt1 : nkt1;                                             // split-bf16 planes: three passes over the K tiles (hi hi, hi lo, lo hi)
    const int per = (nkt_total + p.splitk - 1) / p.splitk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt_total, kt_begin + per);

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    XVA_T(0);

    Loader<AK, BM, NW> la;
    Loader<BKD, BN, NW> lb;
    if constexpr (AK == KC) la.init(lane, wave, m0, p.M, p.lda, 0, 0, 0, p.a_seglen, p.a_segadj);
    else la.init(lane, wave, m0, (p.M + 7) & ~7, p.lda, p.a_seglen, 0, p.a_segadj);          // TN: A column m -> m + (m / a_seglen) * a_segadj; bound: see xva_gemm_glds_eligible
    if constexpr (BKD == KC) lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0);
    else if constexpr (LAYOUT == XVA_GEMM_TN) lb.init(lane, wave, n0, p.N, p.ldb, p.seglen, p.seg0, p.segstride);
    else lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0, p.seglen, p.segstride);      // NN: row segments (tile base carries the tile's first segment)

    auto a_base = [&](int k0) -> const uint16_t* {
        if constexpr (AK == KC) return A + k0 + (p.a_seglen > 0 ? (int64_t)(k0 / p.a_seglen) * p.a_segadj : 0);
        else return A + (int64_t)k0 * p.lda;
    };
    auto b_base = [&](int k0) -> const uint16_t* {
        if constexpr (BKD == KC) return B + k0;
        else if constexpr (LAYOUT == XVA_GEMM_NN)
            return p.seglen > 0 ? B + p.seg0 + (int64_t)(k0 / p.seglen) * p.segstride + (int64_t)(k0 % p.seglen) * p.ldb : B + (int64_t)k0 * p.ldb;
        else return B + (int64_t)k0 * p.ldb;
    };
    auto issue = [&](int kt, int buf) {
        if constexpr (LAYOUT == XVA_GEMM_TN) {
            if (p.kb_len > 0) {   // K blocks (one per item, any length): tile kt = (block, 64-row piece of it); rows past the block end are zero
                const int blk = kt / tpb, kl = (kt - blk * tpb) * GK;
                la.issue(A + (int64_t)blk * p.kb_sA + (int64_t)kl * p.lda, kl, p.kb_len, smem + buf * BUF, wave);
                lb.issue(B + (int64_t)blk * p.kb_sB + (int64_t)kl * p.ldb, kl, p.kb_len, smem + buf * BUF + A_BYTES, wave);
                return;
            }
        }
        if (p.planes) {
            const int pass = kt / nkt1, k0 = (kt - pass * nkt1) * GK;
            la.issue(a_base(k0) + (pass == 2 ? p.a_plane : 0), k0, p.K, smem + buf * BUF, wave);
            lb.issue(b_base(k0) + (pass == 1 ? p.b_plane : 0), k0, p.K, smem + buf * BUF + A_BYTES, wave);
            return;
        }
        const int k0 = kt * GK;
        la.issue(a_base(k0), k0, p.K, smem + buf * BUF, wave);
        lb.issue(b_base(k0), k0, p.K, smem + buf * BUF + A_BYTES, wave);
    };

    KcReader kra, krb;
    IcReader<BM, MI> ira;
    IcReader<BN, NJ> irb;
    if constexpr (AK == KC) kra.init(lane); else ira.init(lane, wm * WM);
    if constexpr (BKD == KC) krb.init(lane); else irb.init(lane, wn * WN);

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K loop, two LDS stages, ONE AND A HALF tiles in flight: the DMA of tile kt + 2 is issued in the middle of tile kt — as soon as every
    // wave has read the second half of tile kt out of its buffer (barrier A) — and only tile kt + 1 has to have landed at the end of the
    // iteration (`vmcnt(LOADS)` leaves the youngest tile's loads outstanding; barrier B).  Inside a training step the operands come from
    // HBM (the producer's writes do not stay in the Infinity Cache) and one tile of prefetch distance did not cover that latency.
    constexpr int LOADS = Loader<AK, BM, NW>::NI + Loader<BKD, BN, NW>::NI;          // DMA instructions per wave per tile
    constexpr int WAIT_YOUNGEST = 0x0F70 | (LOADS & 15) | ((LOADS >> 4) << 14);       // s_waitcnt vmcnt(LOADS)
    constexpr bool ANY_IC = AK == IC || BKD == IC;
    auto read_frags = [&](const XVA_LDS uint8_t* At, const XVA_LDS uint8_t* Bt, int kh, Frag<AK> (&af)[MI], Frag<BKD> (&bfr)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (BKD == KC) bfr[j] = krb.read(Bt, wn * WN + j * 16, kh);
            else bfr[j] = irb.read(Bt, j, kh);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if constexpr (AK == KC) af[i] = kra.read(At, wm * WM + i * 16, kh);
            else af[i] = ira.read(At, i, kh);
        }
    };
    // wait for the transpose reads (if any), take the operand values (LeakyReLU of an operand applied here), multiply
    auto mfma_all = [&](Frag<AK> (&afr)[MI], Frag<BKD> (&bfrr)[NJ]) {
        frags_wait<ANY_IC>();
        bf16x8 af[MI], bfr[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = frag_value(afr[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = frag_value(bfrr[j]);
        if (p.a_lrelu) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = lrelu_frag<F16>(af[i], p.a_slope);
        }
        if (p.b_lrelu) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = lrelu_frag<F16>(bfr[j], p.b_slope);
        }
        if constexpr (XVA_GLDS_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(bfr[j]));
            return;
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = mma16<F16>(bfr[j], af[i], acc[i][j]);   // swapped: lane = 4 columns of a row
    };
    if (kt_begin < kt_end) issue(kt_begin, 0);
    if (kt_begin + 1 < kt_end) { issue(kt_begin + 1, 1); __builtin_amdgcn_s_waitcnt(WAIT_YOUNGEST); }
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    XVA_T(1);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const int cur = (kt - kt_begin) & 1;
        const XVA_LDS uint8_t* At = smem + cur * BUF;
        const XVA_LDS uint8_t* Bt = At + A_BYTES;
        Frag<AK> af[MI];
        Frag<BKD> bfr[NJ];
        read_frags(At, Bt, 0, af, bfr);
        mfma_all(af, bfr);
        read_frags(At, Bt, 1, af, bfr);
        frags_wait<ANY_IC>();
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): this wave's reads of the buffer are in registers
        __builtin_amdgcn_s_barrier();                            // A: the buffer of tile kt is free
        const bool more = (XVA_GLDS_ABLATE & 2) ? false : kt + 2 < kt_end;
        if (more) issue(kt + 2, cur);
        mfma_all(af, bfr);
        if (more) __builtin_amdgcn_s_waitcnt(WAIT_YOUNGEST); else __builtin_amdgcn_s_waitcnt(0x0F70);   // tile kt + 1 has landed (tile kt + 2 may be in flight)
        __builtin_amdgcn_s_barrier();                            // B
    }
    XVA_T(2);
    if constexpr (XVA_GLDS_ABLATE & 8) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else if constexpr (EPI != 7)   // host-checked (epi_variant): see xva_gemm_glds8_kernel
        tile_epilogue_rows<MI, NJ, F16, EPI>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, bz, ks, nt_store);
    else if (rows_epilogue_ok(p, vec_epi))
        tile_epilogue_rows<MI, NJ, F16>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, bz, ks, nt_store);
    else
        tile_epilogue<MI, NJ, F16>(p, acc, vec_epi, m0 + wm * WM + (lane & 15), n0 + wn * WN + (lane >> 4) * 4, z1, z2, bz, ks);
    XVA_T(3);
}

// ---- 32-deep K tiles -------------------------------------------------------------------------------------------------------------
constexpr int GK3 = 32;
#ifndef XVA_GLDS8_WHOLE
#define XVA_GLDS8_WHOLE 1          // NT products on the 256 x 256 tile: the loop with whole-line DMA pieces (xva_gemm_glds8w_kernel) when the geometry allows
#endif
#ifdef XVA_GLDS_TIMING
static inline int xva_gemm_glds_wholeline() { return XVA_GLDS8_WHOLE; }     // tools/glds_timing.hip: fixed at compile time
#else
int xva_gemm_glds_wholeline();                                              // gemm_glds.hip (xva_gemm_set_wholeline)
#endif
#ifndef XVA_GLDS8_SLOTS
#define XVA_GLDS8_SLOTS 4          // ring slots of the staggered 256x256 K loop (4 or 5): 4 x 32 KiB, three tiles (96 k) in flight; 5 slots (all 160 KiB)
                                   // measured +2 % warm, nothing inside the training steps
#endif
__device__ __forceinline__ int kc32_f(int row) { return (4 - ((row >> 2) & 3)) & 3; }
template <int KIND, int ROWS, int NW>
struct Loader32 {
    static constexpr int NI = ROWS * GK3 * 2 / 1024 / NW;   // wave instructions (1024 bytes each) per wave per tile
    int64_t off[NI];
    int kk[NI];
    __device__ __forceinline__ void init(int lane, int wave, int i0, int bound, int64_t ld, int cseglen, int64_t cseg0, int64_t csegstride,
                                         int kseglen = 0, int64_t ksegadj = 0) {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int Q = q * NW + wave;
            if constexpr (KIND == KC) {
                const int r = Q * 16 + (lane >> 2);
                const int c = (lane & 3) ^ kc32_f(lane >> 2);
                off[q] = (int64_t)min(i0 + r, bound - 1) * ld + c * 8 + (kseglen > 0 ? (int64_t)((c * 8) / kseglen) * ksegadj : 0);
                kk[q] = c * 8;
            } else {
                constexpr int CPR = ROWS / 8;
                constexpr int RPI = 64 / CPR;
                const int pos = Q * RPI + lane / CPR;
                const int pch = lane % CPR;
                const int c = ic_chunk<ROWS>(pos, pch);
                const int k = swap23(pos);
                int col = min(i0 + c * 8, bound - 8);
                int64_t cm = col;
                if (cseglen > 0) cm = cseg0 + col + (int64_t)(col / cseglen) * csegstride;
                off[q] = (kseglen > 0 ? (int64_t)(k / kseglen) * ksegadj + (int64_t)(k % kseglen) * ld : (int64_t)k * ld) + cm;
                kk[q] = k;
            }
        }
    }
    __device__ __forceinline__ void issue(const uint16_t* base, int k0, int K, XVA_LDS uint8_t* tile, int wave) const {
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const uint16_t* src = (k0 + kk[q] < K) ? base + off[q] : reinterpret_cast<const uint16_t*>(g_zero_page);
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)src, (XVA_LDS void*)(tile + (q * NW + wave) * 1024), 16, 0, 0);
        }
    }
    // the whole 32-deep tile lies inside K: no zero-page select
    __device__ __forceinline__ void issue_full(const uint16_t* base, XVA_LDS uint8_t* tile, int wave) const {
#pragma unroll
        for (int q = 0; q < NI; ++q)
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)(base + off[q]), (XVA_LDS void*)(tile + (q * NW + wave) * 1024), 16, 0, 0);
    }
};
struct KcReader32 {
    uint32_t o;
    __device__ __forceinline__ void init(int lane) { o = (lane & 15) * 64 + (((lane >> 4) ^ kc32_f(lane & 15)) * 16); }
    __device__ __forceinline__ Frag<KC> read(const XVA_LDS uint8_t* tile, int row0) const {
        Frag<KC> f; f.v = *reinterpret_cast<const XVA_LDS bf16x8*>(tile + row0 * 64 + o); return f;
    }
};


// ---- 256 x 256 tile, staggered wave groups -----------------------------------------------------------------------------------------
// Same operands and epilogues as xva_gemm_glds_kernel<256, 256, 128, 64>; a different K loop.  There, the barriers keep all 8 waves in
// the same phase: every SIMD first waits for its two waves' LDS fragment reads (measured 0.5 us per 64-deep K tile) and then runs their
// MFMAs (0.97 us) — the sum, not the maximum (tools/glds_timing.hip ablations: MFMAs alone 17.5 us per 18 tiles, reads + DMA + barriers
// alone 16.7, everything 30.8).  Here the two row halves of the tile (wave groups wm = 0 / 1; each SIMD holds one wave of either) run ONE
// BARRIER APART: a phase is {read the 12 fragments of a 32-deep K tile, issue the DMA of tile kt + 3 | barrier | 32 MFMAs | barrier},
// and group 1 enters the loop through one extra barrier, so that between any two barriers one group computes while the other reads LDS
// and issues DMA — the matrix pipe of a SIMD alternates between its two waves instead of idling through the reads.
// LDS: a ring of four 32-deep tiles (4 x 32 KiB): three tiles = 96 k in flight.  Hazards (interval I(2t+1): group 0 reads tile t, group 1
// computes tile t - 1; I(2t+2): group 1 reads tile t, group 0 computes it):
//   * tile t - 1 is last read in I(2t) and every read slot ends with lgkmcnt(0) before its barrier; tile t + 3 is written to its slot
//     from I(2t+1) on: after a barrier every reader has passed;
//   * tile t + 1 is first read in I(2t+3); every wave waits for its own tile t + 1 loads (vmcnt leaves tiles t + 2, t + 3 outstanding) in
//     its read slot of tile t (I(2t+1) / I(2t+2)), i.e. before a barrier the first reader passes.
// The same loop serves a 384 x 128 tile (BM = 384, BN = 128, 8 waves of 96 x 64: products with 256 < N <= 384 — FastPitch's d_model — NT / NN only, a
// 384-wide index-contiguous image is not laid out): the ring slots have the same 32 KiB (24 + 8), a phase is 10 fragment reads | 24 MFMAs, and the
// two groups are the waves 0 - 3 / 4 - 7 (row quarters 0, 1 / 2, 3) — what matters is that every SIMD holds one wave of either group.  Before, that
// tile ran the lock-step loop of xva_gemm_glds_kernel (1.78 us per 64-deep K tile against an MFMA floor of 0.74).
template <int LAYOUT, int BM = 256, int BN = 256, int WM = 128, int WN = 64, bool F16 = false, int EPI = 7>
__global__ __launch_bounds__(512, 1) void xva_gemm_glds8_kernel(xva_gemm_params p, int vec_flags) {
    const int vec_epi = vec_flags & 15, nt_store = vec_flags >> 4;     // bit 4: non-temporal C stores (gemm_glds.hip)
    constexpr int NWN = BN / WN, NW = (BM / WM) * NWN;
    static_assert(NW == 8 && (BM + BN) == 512, "8 waves, 32 KiB ring slots");
    static_assert(LAYOUT != XVA_GEMM_TN || BM == 256, "TN: A is an index-contiguous image (power-of-two widths only)");
    constexpr int MI = WM / 16, NJ = WN / 16;
    constexpr int AK = LAYOUT == XVA_GEMM_TN ? IC : KC;
    constexpr int BKD = LAYOUT == XVA_GEMM_NT ? KC : IC;
    constexpr int A_BYTES = BM * GK3 * 2, B_BYTES = BN * GK3 * 2, BUF = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;

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
    const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (int64_t)z1 * p.sA + (int64_t)z2 * p.sA2;
    const uint16_t* B = reinterpret_cast<const uint16_t*>(p.B) + (int64_t)z1 * p.sB + (int64_t)z2 * p.sB2;

    const int tpb = p.kb_len > 0 ? (p.kb_len + GK3 - 1) / GK3 : 1;
    const int nkt1 = p.kb_len > 0 ? (p.K / p.kb_len) * tpb : (p.K + GK3 - 1) / GK3;
    const int nkt_total = p.planes ? 3 * nkt1 : nkt1;            // split-bf16 planes: three passes over the K tiles (hi hi, hi lo, lo hi); host-checked: no segments / K blocks
    const int per = (nkt_total + p.splitk - 1) / p.splitk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt_total, kt_begin + per);

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int grp = wave >> 2;                                   // wave group: waves g and g + 4 share a SIMD
    XVA_T(0);

    Loader32<AK, BM, NW> la;
    Loader32<BKD, BN, NW> lb;
    if constexpr (AK == KC) la.init(lane, wave, m0, p.M, p.lda, 0, 0, 0, p.a_seglen, p.a_segadj);
    else la.init(lane, wave, m0, (p.M + 7) & ~7, p.lda, p.a_seglen, 0, p.a_segadj);
    if constexpr (BKD == KC) lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0);
    else if constexpr (LAYOUT == XVA_GEMM_TN) lb.init(lane, wave, n0, p.N, p.ldb, p.seglen, p.seg0, p.segstride);
    else lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0, p.seglen, p.segstride);

    // Tiles are issued strictly in order (kt_begin, kt_begin + 1, ...): the operand bases of the NEXT tile are kept incrementally — the
    // divisions by the segment / K-block lengths cost ~60 scalar instructions per tile in the read slot, which has to fit under the
    // other wave group's 32 MFMAs (512 cycles).
    const uint16_t* nA; const uint16_t* nB;      // bases of the next tile to issue
    int n_k0, n_akin = 0, n_bkin = 0;            // its first k (inside its K block for TN K blocks); position inside the A / B segment
    int n_pass = 0;                              // split-bf16 planes: the pass the next tile belongs to
    const uint16_t* const A0 = A; const uint16_t* const B0 = B;
    const bool kblocks = LAYOUT == XVA_GEMM_TN && p.kb_len > 0;
    const int Kb = kblocks ? p.kb_len : p.K;     // bound of the k index the loaders compare against
    {
        if (kblocks) {
            const int blk = kt_begin / tpb, kl = (kt_begin - blk * tpb) * GK3;
            n_k0 = kl;
            nA = A + (int64_t)blk * p.kb_sA + (int64_t)kl * p.lda;
            nB = B + (int64_t)blk * p.kb_sB + (int64_t)kl * p.ldb;
        } else {
            n_pass = p.planes ? kt_begin / nkt1 : 0;
            const int k0 = (kt_begin - n_pass * nkt1) * GK3;
            n_k0 = k0;
            if (p.planes) { A += n_pass == 2 ? p.a_plane : 0; B += n_pass == 1 ? p.b_plane : 0; }
            if constexpr (AK == KC) {
                nA = A + k0 + (p.a_seglen > 0 ? (int64_t)(k0 / p.a_seglen) * p.a_segadj : 0);
                n_akin = p.a_seglen > 0 ? k0 % p.a_seglen : 0;
            } else nA = A + (int64_t)k0 * p.lda;
            if constexpr (BKD == KC) nB = B + k0;
            else if constexpr (LAYOUT == XVA_GEMM_NN) {
                if (p.seglen > 0) { nB = B + p.seg0 + (int64_t)(k0 / p.seglen) * p.segstride; n_bkin = k0 % p.seglen; nB += (int64_t)n_bkin * p.ldb; }
                else nB = B + (int64_t)k0 * p.ldb;
            } else nB = B + (int64_t)k0 * p.ldb;
        }
    }
    const int64_t a_small = (AK == KC && p.a_seglen > 0 && p.a_seglen < GK3) ? (int64_t)(GK3 / p.a_seglen) * p.a_segadj : 0;   // segments shorter than a tile
    auto issue_next = [&](int slot) {
        XVA_LDS uint8_t* st = smem + slot * BUF;
        if (n_k0 + GK3 <= Kb) { la.issue_full(nA, st, wave); lb.issue_full(nB, st + A_BYTES, wave); }
        else { la.issue(nA, n_k0, Kb, st, wave); lb.issue(nB, n_k0, Kb, st + A_BYTES, wave); }
        // advance to the next tile
        n_k0 += GK3;
        if (p.planes) {                           // plain operands (no segments): the next pass restarts at k = 0 on the other plane of A or B
            if (n_k0 >= Kb) {
                ++n_pass; n_k0 = 0;
                nA = A0 + (n_pass == 2 ? p.a_plane : 0);
                nB = B0 + (n_pass == 1 ? p.b_plane : 0);
            } else {
                if constexpr (AK == KC) nA += GK3; else nA += (int64_t)GK3 * p.lda;
                if constexpr (BKD == KC) nB += GK3; else nB += (int64_t)GK3 * p.ldb;
            }
            return;
        }
        if (kblocks) {
            if (n_k0 >= tpb * GK3) {              // next K block
                nA += p.kb_sA - (int64_t)(n_k0 - GK3) * p.lda; nB += p.kb_sB - (int64_t)(n_k0 - GK3) * p.ldb; n_k0 = 0;
            } else { nA += (int64_t)GK3 * p.lda; nB += (int64_t)GK3 * p.ldb; }
            return;
        }
        if constexpr (AK == KC) {
            nA += GK3 + a_small;
            if (p.a_seglen >= GK3) { n_akin += GK3; if (n_akin >= p.a_seglen) { n_akin -= p.a_seglen; nA += p.a_segadj; } }
        } else nA += (int64_t)GK3 * p.lda;
        if constexpr (BKD == KC) nB += GK3;
        else if constexpr (LAYOUT == XVA_GEMM_NN) {
            if (p.seglen >= GK3) {
                n_bkin += GK3;
                if (n_bkin >= p.seglen) { n_bkin -= p.seglen; nB += p.segstride - (int64_t)(p.seglen - GK3) * p.ldb; }
                else nB += (int64_t)GK3 * p.ldb;
            } else if (p.seglen > 0) nB += (int64_t)(GK3 / p.seglen) * p.segstride;
            else nB += (int64_t)GK3 * p.ldb;
        } else nB += (int64_t)GK3 * p.ldb;
    };

    KcReader32 kra, krb;
    IcReader<BM, MI> ira;
    IcReader<BN, NJ> irb;
    if constexpr (AK == KC) kra.init(lane); else ira.init(lane, wm * WM);
    if constexpr (BKD == KC) krb.init(lane); else irb.init(lane, wn * WN);

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int LOADS = Loader32<AK, BM, NW>::NI + Loader32<BKD, BN, NW>::NI;      // DMA instructions per wave per 32-deep tile (4)
    constexpr int NS = XVA_GLDS8_SLOTS;                                               // ring slots; NS - 1 tiles in flight
    static_assert((NS - 2) * LOADS < 16, "vmcnt immediates below");
    constexpr int WAIT_VM3 = 0x0F70 | (3 * LOADS), WAIT_VM2 = 0x0F70 | (2 * LOADS), WAIT_VM1 = 0x0F70 | LOADS, WAIT_VM0 = 0x0F70;
#define XVA_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    const int ntl = kt_end - kt_begin;
    if (ntl > 0) issue_next(0);
    if (ntl > 1) issue_next(1);
    if (ntl > 2) issue_next(2);
    if (NS > 4 && ntl > 3) issue_next(3);
    if (NS > 4 && ntl > 3) __builtin_amdgcn_s_waitcnt(WAIT_VM3);
    else if (ntl > 2) __builtin_amdgcn_s_waitcnt(WAIT_VM2); else if (ntl > 1) __builtin_amdgcn_s_waitcnt(WAIT_VM1); else __builtin_amdgcn_s_waitcnt(WAIT_VM0);
    XVA_BAR();
    if (grp == 1) XVA_BAR();                                     // group 1 runs one barrier behind group 0
    XVA_T(1);
    int slot = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const XVA_LDS uint8_t* At = smem + slot * BUF;
        const XVA_LDS uint8_t* Bt = At + A_BYTES;
        Frag<AK> afr[MI];
        Frag<BKD> bfrr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (BKD == KC) bfrr[j] = krb.read(Bt, wn * WN + j * 16);
            else bfrr[j] = irb.read(Bt, j, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if constexpr (AK == KC) afr[i] = kra.read(At, wm * WM + i * 16);
            else afr[i] = ira.read(At, i, 0);
        }
        const int ahead = (XVA_GLDS_ABLATE & 2) ? 0 : kt_end - kt;     // tiles left including this one
        // tile kt + NS - 1 goes into the slot of tile kt - 1; then this wave's part of tile kt + 1 must have landed
        if (ahead > NS - 1) { issue_next(slot == 0 ? NS - 1 : slot - 1); __builtin_amdgcn_s_waitcnt(NS > 4 ? WAIT_VM3 : WAIT_VM2); }
        else if (NS > 4 && ahead > 3) __builtin_amdgcn_s_waitcnt(WAIT_VM2);
        else if (ahead > 2) __builtin_amdgcn_s_waitcnt(WAIT_VM1);
        else __builtin_amdgcn_s_waitcnt(WAIT_VM0);
        frags_wait<AK == IC || BKD == IC>();                     // the transpose reads (inline asm: see Frag)
        bf16x8 af[MI], bfr[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = frag_value(afr[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = frag_value(bfrr[j]);
        if (p.a_lrelu) {                                         // one uniform branch per read slot
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = lrelu_frag<F16>(af[i], p.a_slope);
        }
        if (p.b_lrelu) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = lrelu_frag<F16>(bfr[j], p.b_slope);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
        XVA_BAR();
        if constexpr (XVA_GLDS_ABLATE & 1) {
#pragma unroll
            for (int i = 0; i < MI; ++i) asm volatile("" :: "v"(af[i]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(bfr[j]));
        } else {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = mma16<F16>(bfr[j], af[i], acc[i][j]);
            __builtin_amdgcn_s_setprio(0);
        }
        XVA_BAR();
        slot = slot == NS - 1 ? 0 : slot + 1;
    }
    if (grp == 0) XVA_BAR();
#undef XVA_BAR
    XVA_T(2);
    if constexpr (EPI != 7)         // host-checked (epi_variant): the row-contiguous epilogue serves this launch and needs no more than EPI's features
        tile_epilogue_rows<MI, NJ, F16, EPI>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, bz, ks, nt_store);
    else if (rows_epilogue_ok(p, vec_epi))
        tile_epilogue_rows<MI, NJ, F16>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, bz, ks, nt_store);
    else
        tile_epilogue<MI, NJ, F16>(p, acc, vec_epi, m0 + wm * WM + (lane & 15), n0 + wn * WN + (lane >> 4) * 4, z1, z2, bz, ks);
    XVA_T(3);
}

// ---- 256 x 256 tile, staggered wave groups, WHOLE-LINE DMA pieces (NT) -----------------------------------------------------------------------
// The loop of xva_gemm_glds8_kernel for two K-contiguous operands (NT), with one change in how the operands reach the LDS.  There a DMA piece (one
// global_load_lds_dwordx4 of a wave, 1 KiB) gathers 16 rows x 64 bytes — a 32-deep bf16 K tile — and the CU's address unit spends one cycle per 128-byte line a
// piece touches: 16 per piece, 64 B/clk (tools/dma_issue_probe.hip: 130 ticks per piece with eight waves issuing, against 86 for 8 rows x 128 bytes).  Four waves of
// a group issue their 16 pieces into one address unit, so the read slot of the staggered loop carried ~300 cycles of DMA issue (profiles/r06_kloop_ablation.txt) and
// was longer than the other group's 32 MFMAs.  Here a piece is 8 rows x 128 bytes (whole lines, the 64-deep image of xva_gemm_glds_kernel: Loader / KcReader), the
// ring holds FIVE 32 KiB units — one operand's 256 rows x 64 k each, in the order A(p), B(p), A(p + 1), ... — and a 32-deep phase reads one 64-byte half of its
// pair's rows.  A pair is consumed over two tiles; the units of pair p - 1 are free from tile 2p on, so tile i issues unit i + 3 (even tiles a B unit — the
// weights, warm in L2 — one tile-time before its first reader; odd tiles an A unit two tile-times ahead).  Host-checked (launch_tile8): NT, K % 64 == 0, tap
// segments of A a multiple of 64 (or none), no planes, no K blocks.
template <bool F16 = false, int EPI = 7>
__global__ __launch_bounds__(512, 1) void xva_gemm_glds8w_kernel(xva_gemm_params p, int vec_flags) {
    const int vec_epi = vec_flags & 15, nt_store = vec_flags >> 4;
    constexpr int BM = 256, BN = 256, WM = 128, WN = 64;
    constexpr int NWN = BN / WN, NW = 8;
    constexpr int MI = WM / 16, NJ = WN / 16;
    constexpr int UNIT = 256 * 128, NU = 5;                      // 32 KiB per operand pair-unit; five of them = all 160 KiB
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;

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
    const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (int64_t)z1 * p.sA + (int64_t)z2 * p.sA2;
    const uint16_t* B = reinterpret_cast<const uint16_t*>(p.B) + (int64_t)z1 * p.sB + (int64_t)z2 * p.sB2;

    const int npair = p.K / 64;                                  // host-checked: K % 64 == 0
    const int per = (npair + p.splitk - 1) / p.splitk;
    const int pb = ks * per, pe = min(npair, pb + per);
    const int ntl = pe > pb ? 2 * (pe - pb) : 0;                 // 32-deep tiles of this workgroup
    const int nun = ntl;                                         // units (one A and one B per pair)

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int grp = wave >> 2;
    XVA_T(0);

    Loader<KC, BM, NW> la;
    Loader<KC, BN, NW> lb;
    la.init(lane, wave, m0, p.M, p.lda, 0, 0, 0, p.a_seglen, p.a_segadj);
    lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0);
    // operand bases of the next A / B unit to issue
    const int k0 = pb * 64;
    const uint16_t* nA = A + k0 + (p.a_seglen > 0 ? (int64_t)(k0 / p.a_seglen) * p.a_segadj : 0);
    const uint16_t* nB = B + k0;
    int n_akin = p.a_seglen > 0 ? k0 % p.a_seglen : 0;
    const int a_seg = p.a_seglen > 0 ? p.a_seglen : 0x7fffffff;
    auto issue_unit = [&](int u) {                               // u: local unit index (even: A of pair u / 2, odd: B); slot u % 5
        XVA_LDS uint8_t* st = smem + (u % NU) * UNIT;
        if (u & 1) { lb.issue_full(nB, st, wave); nB += 64; }
        else {
            la.issue_full(nA, st, wave);
            n_akin += 64;
            const bool cross = n_akin >= a_seg;
            n_akin = cross ? n_akin - a_seg : n_akin;
            nA += 64 + (cross ? p.a_segadj : (int64_t)0);
        }
    };
    KcReader kra, krb;
    kra.init(lane); krb.init(lane);

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int UL = Loader<KC, BM, NW>::NI;                    // DMA pieces per wave per unit (4)
    static_assert(UL == 4 && Loader<KC, BN, NW>::NI == 4, "vmcnt immediates below");
#define XVA_BAR() do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); } while (0)
    {   // prologue: units 0 .. 4; units 0 and 1 (the first pair) have to land
        const int n0u = nun < NU ? nun : NU;
        for (int u = 0; u < n0u; ++u) issue_unit(u);
        if (n0u >= 5) __builtin_amdgcn_s_waitcnt(0x0F70 | 12); else if (n0u == 4) __builtin_amdgcn_s_waitcnt(0x0F70 | 8);
        else if (n0u == 3) __builtin_amdgcn_s_waitcnt(0x0F70 | 4); else __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    XVA_BAR();
    if (grp == 1) XVA_BAR();                                     // group 1 runs one barrier behind group 0
    XVA_T(1);
    int ua = 0;                                                  // slot of the current pair's A unit (0 .. 4); its B unit sits in the next slot
    for (int i = 0; i < ntl; ++i) {
        const int h = i & 1;
        const XVA_LDS uint8_t* At = smem + ua * UNIT;
        const XVA_LDS uint8_t* Bt = smem + (ua == NU - 1 ? 0 : ua + 1) * UNIT;
        Frag<KC> afr[MI], bfrr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfrr[j] = krb.read(Bt, wn * WN + j * 16, h);
#pragma unroll
        for (int i2 = 0; i2 < MI; ++i2) afr[i2] = kra.read(At, wm * WM + i2 * 16, h);
        // tile i >= 2 issues unit i + 3 (the slot of unit i - 2, whose pair every reader has left); an odd tile then waits for the next pair's units (<= i + 2)
        const bool more = !(XVA_GLDS_ABLATE & 2) && i + 3 < nun;
        if (i >= 2 && more) issue_unit(i + 3);
        if (h) { if (more) __builtin_amdgcn_s_waitcnt(0x0F70 | UL); else __builtin_amdgcn_s_waitcnt(0x0F70); }
        bf16x8 af[MI], bfr[NJ];
#pragma unroll
        for (int i2 = 0; i2 < MI; ++i2) af[i2] = frag_value(afr[i2]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = frag_value(bfrr[j]);
        if (p.a_lrelu) {
#pragma unroll
            for (int i2 = 0; i2 < MI; ++i2) af[i2] = lrelu_frag<F16>(af[i2], p.a_slope);
        }
        if (p.b_lrelu) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = lrelu_frag<F16>(bfr[j], p.b_slope);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
        XVA_BAR();
        if constexpr (XVA_GLDS_ABLATE & 1) {
#pragma unroll
            for (int i2 = 0; i2 < MI; ++i2) asm volatile("" :: "v"(af[i2]));
#pragma unroll
            for (int j = 0; j < NJ; ++j) asm volatile("" :: "v"(bfr[j]));
        } else {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i2 = 0; i2 < MI; ++i2)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i2][j] = mma16<F16>(bfr[j], af[i2], acc[i2][j]);
            __builtin_amdgcn_s_setprio(0);
        }
        XVA_BAR();
        if (h) ua = ua + 2 >= NU ? ua + 2 - NU : ua + 2;
    }
    if (grp == 0) XVA_BAR();
#undef XVA_BAR
    XVA_T(2);
    if constexpr (EPI != 7)
        tile_epilogue_rows<MI, NJ, F16, EPI>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, bz, ks, nt_store);
    else if (rows_epilogue_ok(p, vec_epi))
        tile_epilogue_rows<MI, NJ, F16>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, bz, ks, nt_store);
    else
        tile_epilogue<MI, NJ, F16>(p, acc, vec_epi, m0 + wm * WM + (lane & 15), n0 + wn * WN + (lane >> 4) * 4, z1, z2, bz, ks);
    XVA_T(3);
}

// ---- 256 x 128 tile, K tile 32, three LDS stages, TWO workgroups per CU ------------------------------------------------------------
// The 256 x 256 kernels hold one workgroup per CU: nothing runs under its 9 us epilogue (all 256 workgroups of a round store their
// 128 KiB at the same time), under its prologue, or while its waves sit at a barrier.  Here a workgroup is 4 waves (one per SIMD) with
// the same 128 x 64 wave tile (same LDS bytes per flop), a 32-deep K tile of (256 + 128) rows = 24 KiB and a ring of three of them
// (72 KiB), so that TWO independent workgroups share a CU: each SIMD holds one wave of either, their barriers and epilogues are not
// synchronised, and whatever one workgroup waits for, the other's MFMAs fill.  Two K tiles (64 k) are in flight, one barrier per tile.
// KC image: [ROWS][32 k] bf16 = 64-byte rows, 16-byte chunk c of row r at position c ^ f(r), f(r) = (4 - ((r >> 2) & 3)) & 3: the four
// 16-lane groups of a ds_read_b128 fragment read ({rows v, 12 + v: chunk c}, {rows 4 + v, 8 + v: chunk c ^ 1}) each cover the 64 banks once.
// IC image: the first 32 k-rows of the 64-deep image above (same swizzles, same transpose reads).
template <int LAYOUT, int BM, int BN, bool F16 = false>
__global__ __launch_bounds__((BM / 128) * (BN / 64) * 64, 2) void xva_gemm_glds3_kernel(xva_gemm_params p, int vec_epi) {
    constexpr int WM = 128, WN = 64;
    constexpr int NWN = BN / WN, NW = (BM / WM) * NWN;
    static_assert(NW == 4, "4 waves");
    constexpr int MI = WM / 16, NJ = WN / 16;
    constexpr int AK = LAYOUT == XVA_GEMM_TN ? IC : KC;
    constexpr int BKD = LAYOUT == XVA_GEMM_NT ? KC : IC;
    constexpr int A_BYTES = BM * GK3 * 2, B_BYTES = BN * GK3 * 2, BUF = A_BYTES + B_BYTES;
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;

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
    const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (int64_t)z1 * p.sA + (int64_t)z2 * p.sA2;
    const uint16_t* B = reinterpret_cast<const uint16_t*>(p.B) + (int64_t)z1 * p.sB + (int64_t)z2 * p.sB2;

    const int tpb = p.kb_len > 0 ? (p.kb_len + GK3 - 1) / GK3 : 1;
    const int nkt_total = p.kb_len > 0 ? (p.K / p.kb_len) * tpb : (p.K + GK3 - 1) / GK3;
    const int per = (nkt_total + p.splitk - 1) / p.splitk;
    const int kt_begin = ks * per;
    const int kt_end = min(nkt_total, kt_begin + per);

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    Loader32<AK, BM, NW> la;
    Loader32<BKD, BN, NW> lb;
    if constexpr (AK == KC) la.init(lane, wave, m0, p.M, p.lda, 0, 0, 0, p.a_seglen, p.a_segadj);
    else la.init(lane, wave, m0, (p.M + 7) & ~7, p.lda, p.a_seglen, 0, p.a_segadj);
    if constexpr (BKD == KC) lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0);
    else if constexpr (LAYOUT == XVA_GEMM_TN) lb.init(lane, wave, n0, p.N, p.ldb, p.seglen, p.seg0, p.segstride);
    else lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0, p.seglen, p.segstride);

    auto issue = [&](int kt, int slot) {
        XVA_LDS uint8_t* st = smem + slot * BUF;
        if constexpr (LAYOUT == XVA_GEMM_TN) {
            if (p.kb_len > 0) {
                const int blk = kt / tpb, kl = (kt - blk * tpb) * GK3;
                la.issue(A + (int64_t)blk * p.kb_sA + (int64_t)kl * p.lda, kl, p.kb_len, st, wave);
                lb.issue(B + (int64_t)blk * p.kb_sB + (int64_t)kl * p.ldb, kl, p.kb_len, st + A_BYTES, wave);
                return;
            }
        }
        const int k0 = kt * GK3;
        if constexpr (AK == KC) la.issue(A + k0 + (p.a_seglen > 0 ? (int64_t)(k0 / p.a_seglen) * p.a_segadj : 0), k0, p.K, st, wave);
        else la.issue(A + (int64_t)k0 * p.lda, k0, p.K, st, wave);
        const uint16_t* bb;
        if constexpr (BKD == KC) bb = B + k0;
        else if constexpr (LAYOUT == XVA_GEMM_NN)
            bb = p.seglen > 0 ? B + p.seg0 + (int64_t)(k0 / p.seglen) * p.segstride + (int64_t)(k0 % p.seglen) * p.ldb : B + (int64_t)k0 * p.ldb;
        else bb = B + (int64_t)k0 * p.ldb;
        lb.issue(bb, k0, p.K, st + A_BYTES, wave);
    };

    KcReader32 kra, krb;
    IcReader<BM, MI> ira;
    IcReader<BN, NJ> irb;
    if constexpr (AK == KC) kra.init(lane); else ira.init(lane, wm * WM);
    if constexpr (BKD == KC) krb.init(lane); else irb.init(lane, wn * WN);

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int LOADS = Loader32<AK, BM, NW>::NI + Loader32<BKD, BN, NW>::NI;
    constexpr int WAIT_YOUNGEST = 0x0F70 | (LOADS & 15) | ((LOADS >> 4) << 14);
    if (kt_begin < kt_end) issue(kt_begin, 0);
    if (kt_begin + 1 < kt_end) issue(kt_begin + 1, 1);
    int slot = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        if (kt + 1 < kt_end) __builtin_amdgcn_s_waitcnt(WAIT_YOUNGEST); else __builtin_amdgcn_s_waitcnt(0x0F70);   // tile kt has landed (this wave's part)
        __builtin_amdgcn_s_barrier();           // ... every wave's part; and every wave has read tile kt - 1 out of its slot
        if (kt + 2 < kt_end) issue(kt + 2, slot >= 1 ? slot - 1 : 2);
        const XVA_LDS uint8_t* At = smem + slot * BUF;
        const XVA_LDS uint8_t* Bt = At + A_BYTES;
        Frag<AK> afr[MI];
        Frag<BKD> bfrr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (BKD == KC) bfrr[j] = krb.read(Bt, wn * WN + j * 16);
            else bfrr[j] = irb.read(Bt, j, 0);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            if constexpr (AK == KC) afr[i] = kra.read(At, wm * WM + i * 16);
            else afr[i] = ira.read(At, i, 0);
        }
        frags_wait<AK == IC || BKD == IC>();
        bf16x8 af[MI], bfr[NJ];
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = frag_value(afr[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = frag_value(bfrr[j]);
        if (p.a_lrelu) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = lrelu_frag<F16>(af[i], p.a_slope);
        }
        if (p.b_lrelu) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) bfr[j] = lrelu_frag<F16>(bfr[j], p.b_slope);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);     // lgkmcnt(0): the slot's fragments are in registers before this wave reaches the next barrier
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = mma16<F16>(bfr[j], af[i], acc[i][j]);
        slot = slot == 2 ? 0 : slot + 1;
    }
    __builtin_amdgcn_s_barrier();               // the epilogue scratch overlays the ring
    if (rows_epilogue_ok(p, vec_epi))
        tile_epilogue_rows<MI, NJ, F16>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, bz, ks);
    else
        tile_epilogue<MI, NJ, F16>(p, acc, vec_epi, m0 + wm * WM + (lane & 15), n0 + wn * WN + (lane >> 4) * 4, z1, z2, bz, ks);
}

// ---- stride-1 convolution with a RESIDENT input tile ------------------------------------------------------------------------
// A k-tap (dilation d) convolution over CIN channels is the GEMM above with K = k * CIN in tap segments; there every 64-deep K tile
// re-fetches its 128 input rows shifted by the tap, i.e. the input is streamed k times through L2 -> LDS (the measured bound of
// HiFi-GAN's 32/64/128-channel stages: 6.7 TB/s of L2 traffic for 47 MB of input).  Here the workgroup's 128 output rows load
// their 128 + (k-1) d input rows ONCE (global_load_lds, 16-byte chunks XOR-swizzled by the row so that the shifted 16-row MFMA
// fragments stay (nearly) conflict-free); only the small weight tiles stream through the usual double buffer.
// NT: forward (weights [Cout][k * CIN]).  NN: backward-data (weights read through the B row segments, tap-reversed).
constexpr int RES_HALO = 64;
template <int CIN> __device__ __forceinline__ int res_swz(int row) {
    return CIN == 128 ? (row & 15) : (CIN == 64 ? ((row >> 1) & 7) : (CIN == 32 ? ((row >> 2) & 3) : (CIN == 16 ? ((row >> 3) & 1) : 0)));
}

// stride (1, 2, 4: HiFi-GAN's strided discriminator convs) = input rows per output row; rowpitch = elements between consecutive input
// rows (> CIN for a grouped conv: the tile holds one group's channels); the problem's group index is the second batch level (z2).
// minimum waves per SIMD the register allocation has to leave room for (narrow outputs: small accumulators, latency-bound workgroups)
#ifndef XVA_CONV_RES_WAVES
#define XVA_CONV_RES_WAVES(BN) ((BN) <= 32 ? 5 : ((BN) <= 64 ? 3 : 2))
#endif
constexpr int res_a_bytes(int cin, int stride) { return (((stride * 128 + RES_HALO) * cin * 2) + 1023) & ~1023; }
template <int LAYOUT, int CIN, int BN, int WM, int WN, bool F16 = false, int EPI = 7>
__global__ __launch_bounds__((128 / WM) * (BN / WN) * 64, XVA_CONV_RES_WAVES(BN)) void xva_conv_res_kernel(xva_gemm_params p, int vec_epi, int dstep, int stride,
                                                                                       int64_t rowpitch) {
    constexpr int BM = 128;
    constexpr int NWN = BN / WN, NW = (BM / WM) * NWN;
    static_assert(NW == 4, "resident-input conv: 4 waves");
    constexpr int MI = WM / 16, NJ = WN / 16;
    constexpr int BKD = LAYOUT == XVA_GEMM_NT ? KC : IC;
    constexpr int B_BYTES = BN * GK * 2;
    const int A_BYTES = res_a_bytes(CIN, stride);
    constexpr int CPR = CIN / 8, RPI = 64 / CPR;            // 16-byte chunks per input row, rows per DMA instruction
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;

    const int nbx = (p.N + BN - 1) / BN, nby = (p.M + BM - 1) / BM;
    int Lg;
    {
        const unsigned total = gridDim.x, id = blockIdx.x;
        const unsigned xcd = id & 7u, slot = id >> 3, q = total >> 3, r = total & 7u;
        Lg = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot);
    }
    const int tn = Lg % nbx, tm = (Lg / nbx) % nby, z = Lg / (nbx * nby);
    const int b2n = p.batch2 > 1 ? p.batch2 : 1;
    const int z1 = z / b2n, z2 = z - z1 * b2n;
    const int m0 = tm * BM, n0 = tn * BN;
    const uint16_t* A = reinterpret_cast<const uint16_t*>(p.A) + (int64_t)z1 * p.sA + (int64_t)z2 * p.sA2;
    const uint16_t* B = reinterpret_cast<const uint16_t*>(p.B) + (int64_t)z1 * p.sB + (int64_t)z2 * p.sB2;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    XVA_T(0);
    const int ntaps = p.K / CIN;
    const int halo = (ntaps - 1) * (dstep < 0 ? -dstep : dstep);
    const int lo = dstep < 0 ? -halo : 0;                     // taps step forwards (forward conv) or backwards (backward-data)
    const int nkt = (p.K + GK - 1) / GK;

    // resident input rows stride * m0 + lo .. stride * (m0 + BM - 1) + lo + halo (valid input rows: lo .. stride * (M - 1) + lo + halo)
    {
        const int nrows = stride * (BM - 1) + 1 + halo, rmax = stride * (p.M - 1) + lo + halo;
        const int ninstr = (nrows + RPI - 1) / RPI;
        for (int q = wave; q < ninstr; q += NW) {
            const int r = q * RPI + lane / CPR, pch = lane % CPR;
            const int c = pch ^ res_swz<CIN>(r);
            const uint16_t* src = A + (int64_t)min(stride * m0 + lo + r, rmax) * rowpitch + c * 8;
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)src, (XVA_LDS void*)(smem + q * 1024), 16, 0, 0);
        }
    }
    Loader<BKD, BN, NW> lb;
    if constexpr (BKD == KC) lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0);
    else lb.init(lane, wave, n0, p.N, p.ldb, 0, 0, 0, p.seglen, p.segstride);
    auto b_base = [&](int k0) -> const uint16_t* {
        if constexpr (BKD == KC) return B + k0;
        else return p.seglen > 0 ? B + p.seg0 + (int64_t)(k0 / p.seglen) * p.segstride + (int64_t)(k0 % p.seglen) * p.ldb : B + (int64_t)k0 * p.ldb;
    };
    KcReader krb;
    IcReader<BN, NJ> irb;
    if constexpr (BKD == KC) krb.init(lane); else irb.init(lane, wn * WN);
    const int g = lane >> 4;
    const int arow = stride * (wm * WM + (lane & 15)) - lo;

    f32x4 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weight tiles: two LDS stages, one and a half tiles in flight (see xva_gemm_glds_kernel)
    constexpr int LOADS = Loader<BKD, BN, NW>::NI;
    constexpr int WAIT_YOUNGEST = 0x0F70 | (LOADS & 15) | ((LOADS >> 4) << 14);       // s_waitcnt vmcnt(LOADS)
    auto read_frags = [&](const XVA_LDS uint8_t* Bt, int kt, int kh, bf16x8 (&af)[MI], Frag<BKD> (&bfr)[NJ]) {
        const int kl = kt * GK + kh * 32 + g * 8;           // this lane's first k: one MFMA k-step spans 32 / CIN taps when CIN < 32
        const int tap = min(kl / CIN, ntaps - 1);           // a ragged last K tile multiplies zero weights: stay inside the tile
        const int ch = (kl % CIN) / 8;
        const int shift = tap * dstep;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if constexpr (BKD == KC) bfr[j] = krb.read(Bt, wn * WN + j * 16, kh);
            else bfr[j] = irb.read(Bt, j, kh);
        }
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int r = arow + stride * (i * 16) + shift;
            af[i] = *reinterpret_cast<const XVA_LDS bf16x8*>(smem + r * (CIN * 2) + ((ch ^ res_swz<CIN>(r)) << 4));
        }
        if (p.a_lrelu) {
#pragma unroll
            for (int i = 0; i < MI; ++i) af[i] = lrelu_frag<F16>(af[i], p.a_slope);
        }
    };
    auto mfma_all = [&](const bf16x8 (&af)[MI], Frag<BKD> (&bfrr)[NJ]) {
        frags_wait<BKD == IC>();                            // the weight fragments of backward-data are transpose reads (inline asm: see Frag)
        bf16x8 bfr[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bfr[j] = frag_value(bfrr[j]);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                acc[i][j] = mma16<F16>(bfr[j], af[i], acc[i][j]);
    };
    lb.issue(b_base(0), 0, p.K, smem + A_BYTES, wave);
    if (nkt > 1) { lb.issue(b_base(GK), GK, p.K, smem + A_BYTES + B_BYTES, wave); __builtin_amdgcn_s_waitcnt(WAIT_YOUNGEST); }
    else __builtin_amdgcn_s_waitcnt(0x0F70);
    __builtin_amdgcn_s_barrier();
    XVA_T(1);
    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        const XVA_LDS uint8_t* Bt = smem + A_BYTES + cur * B_BYTES;
        bf16x8 af[MI];
        Frag<BKD> bfr[NJ];
        read_frags(Bt, kt, 0, af, bfr);
        mfma_all(af, bfr);
        read_frags(Bt, kt, 1, af, bfr);
        frags_wait<BKD == IC>();
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();                            // A: the weight buffer of tile kt is free
        const bool more = kt + 2 < nkt;
        if (more) lb.issue(b_base((kt + 2) * GK), (kt + 2) * GK, p.K, smem + A_BYTES + cur * B_BYTES, wave);
        mfma_all(af, bfr);
        if (more) __builtin_amdgcn_s_waitcnt(WAIT_YOUNGEST); else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();                            // B: tile kt + 1 has landed
    }
    XVA_T(2);
    if constexpr (EPI != 7)          // host-checked (launch_conv_res): the row-contiguous epilogue serves this launch within EPI's features
        tile_epilogue_rows<MI, NJ, F16, EPI>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, z, 0);
    else if (rows_epilogue_ok(p, vec_epi))
        tile_epilogue_rows<MI, NJ, F16>(p, acc, reinterpret_cast<XVA_LDS float*>(smem + wave * epi_scratch_bytes(WN)), m0 + wm * WM, n0 + wn * WN, lane, z1, z2, z, 0);
    else
        tile_epilogue<MI, NJ, F16>(p, acc, vec_epi, m0 + wm * WM + (lane & 15), n0 + wn * WN + (lane >> 4) * 4, z1, z2, z, 0);
    XVA_T(3);
}

template <int LAYOUT, int CIN, int BN, int WM, int WN, bool F16 = false>
inline int launch_conv_res(const xva_gemm_params& p, int vec_epi, int dstep, int stride, int64_t rowpitch, hipStream_t st) {
    constexpr int LDS_FULL = res_a_bytes(CIN, 4) + 2 * BN * GK * 2;
    constexpr int LDS_MAX = LDS_FULL > 160 * 1024 ? 160 * 1024 : LDS_FULL;      // the plan never admits a (CIN, stride) pair beyond the 160 KiB of a CU
    if (res_a_bytes(CIN, stride) + 2 * BN * GK * 2 > LDS_MAX) return -1;
    const int LDS = res_a_bytes(CIN, stride) + 2 * BN * GK * 2;
    const long nblocks = (long)xva_cdiv(p.N, BN) * xva_cdiv(p.M, 128) * p.batch * p.batch2;
    auto go = [&](auto kern, bool& attr_set) {
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_MAX) != hipSuccess) return -1;
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), LDS, st, p, vec_epi, dstep, stride, rowpitch);
        return 0;
    };
    static bool a5 = false, a7 = false;
    // HiFi-GAN's convolutions never drop out: everything but the dropout hash (and the 4-column fallback epilogue) compiled in — EPI 5 — unless the launch needs them
    if (p.drop_p <= 0.f && rows_epilogue_ok(p, vec_epi & 15)) return go(xva_conv_res_kernel<LAYOUT, CIN, BN, WM, WN, F16, 5>, a5);
    return go(xva_conv_res_kernel<LAYOUT, CIN, BN, WM, WN, F16, 7>, a7);
}

template <int LAYOUT, int BM, int BN, bool F16 = false>
inline int launch_tile3(const xva_gemm_params& p, int vec_epi, hipStream_t st) {
    constexpr int LDS = 3 * (BM + BN) * GK3 * 2;
    auto kern = xva_gemm_glds3_kernel<LAYOUT, BM, BN, F16>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
        attr_set = true;
    }
    long nblocks = (long)xva_cdiv(p.N, BN) * xva_cdiv(p.M, BM) * p.batch * p.batch2 * p.splitk;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(256), LDS, st, p, vec_epi);
    return 0;
}

template <int LAYOUT, int BM = 256, int BN = 256, int WM = 128, int WN = 64, bool F16 = false>
inline int launch_tile8(const xva_gemm_params& p, int vec_epi, hipStream_t st) {
    constexpr int LDS = XVA_GLDS8_SLOTS * (BM + BN) * GK3 * 2;
    const long nblocks = (long)xva_cdiv(p.N, BN) * xva_cdiv(p.M, BM) * p.batch * p.batch2 * p.splitk;
    auto go = [&](auto kern, bool& attr_set) {
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(512), LDS, st, p, vec_epi);
        return 0;
    };
    static bool a0 = false, a1 = false, a3 = false, a7 = false;
    static bool a9 = false, a11 = false;
    const int ev = epi_variant(p, vec_epi, F16);
    if constexpr (LAYOUT == XVA_GEMM_NT && BM == 256 && BN == 256) {
        // two K-contiguous operands: the loop with whole-line DMA pieces (xva_gemm_glds8w_kernel) whenever the geometry allows
        if (xva_gemm_glds_wholeline() && p.K % 64 == 0 && !p.planes && p.kb_len == 0 && (p.a_seglen == 0 || p.a_seglen % 64 == 0)) {
            constexpr int LDSW = 5 * 256 * 128;
            auto gow = [&](auto kern, bool& attr_set) {
                if (!attr_set) {
                    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDSW) != hipSuccess) return -1;
                    attr_set = true;
                }
                hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(512), LDSW, st, p, vec_epi);
                return 0;
            };
            static bool w0 = false, w1 = false, w3 = false, w7 = false, w9 = false, w11 = false;
            if constexpr (F16) {
                if (ev == 9) return gow(xva_gemm_glds8w_kernel<F16, 9>, w9);
                if (ev == 11) return gow(xva_gemm_glds8w_kernel<F16, 11>, w11);
            }
            switch (ev) {
                case 0: return gow(xva_gemm_glds8w_kernel<F16, 0>, w0);
                case 1: return gow(xva_gemm_glds8w_kernel<F16, 1>, w1);
                case 3: return gow(xva_gemm_glds8w_kernel<F16, 3>, w3);
                default: return gow(xva_gemm_glds8w_kernel<F16, 7>, w7);
            }
        }
    }
    if constexpr (F16) {
        if (ev == 9) return go(xva_gemm_glds8_kernel<LAYOUT, BM, BN, WM, WN, F16, 9>, a9);
        if (ev == 11) return go(xva_gemm_glds8_kernel<LAYOUT, BM, BN, WM, WN, F16, 11>, a11);
    }
    switch (ev) {      // (vec_epi carries the flag bits of gemm_glds.hip: bit 4 = non-temporal stores)
        case 0: return go(xva_gemm_glds8_kernel<LAYOUT, BM, BN, WM, WN, F16, 0>, a0);
        case 1: return go(xva_gemm_glds8_kernel<LAYOUT, BM, BN, WM, WN, F16, 1>, a1);
        case 3: return go(xva_gemm_glds8_kernel<LAYOUT, BM, BN, WM, WN, F16, 3>, a3);
        default: return go(xva_gemm_glds8_kernel<LAYOUT, BM, BN, WM, WN, F16, 7>, a7);
    }
}

template <int LAYOUT, int BM, int BN, int WM, int WN, bool F16 = false>
inline int launch_tile(const xva_gemm_params& p, int vec_epi, hipStream_t st) {
    constexpr int NT = (BM / WM) * (BN / WN) * 64;
    constexpr int LDS = 2 * (BM + BN) * GK * 2;
    const long nblocks = (long)xva_cdiv(p.N, BN) * xva_cdiv(p.M, BM) * p.batch * p.batch2 * p.splitk;
    auto go = [&](auto kern, bool& attr_set) {
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1;
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(NT), LDS, st, p, vec_epi);
        return 0;
    };
    static bool a0 = false, a1 = false, a3 = false, a7 = false;
    if constexpr (BM == 128 && BN == 128) {      // the 128 x 128 tile gets the compiled-down epilogue variants too (the other small tiles keep the full one: build time)
        static bool a9 = false, a11 = false;
        const int ev = epi_variant(p, vec_epi, F16);
        if constexpr (F16) {
            if (ev == 9) return go(xva_gemm_glds_kernel<LAYOUT, BM, BN, WM, WN, F16, 9>, a9);
            if (ev == 11) return go(xva_gemm_glds_kernel<LAYOUT, BM, BN, WM, WN, F16, 11>, a11);
        }
        switch (ev) {
            case 0: return go(xva_gemm_glds_kernel<LAYOUT, BM, BN, WM, WN, F16, 0>, a0);
            case 1: return go(xva_gemm_glds_kernel<LAYOUT, BM, BN, WM, WN, F16, 1>, a1);
            case 3: return go(xva_gemm_glds_kernel<LAYOUT, BM, BN, WM, WN, F16, 3>, a3);
            default: break;
        }
    }
    return go(xva_gemm_glds_kernel<LAYOUT, BM, BN, WM, WN, F16, 7>, a7);
}

}  // namespace xva_glds
