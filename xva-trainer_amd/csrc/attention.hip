// attention.hip — fused single-head attention of the FastPitch FFT blocks for bf16-stored activations (gfx950).
//
// Replaces, in the throughput mode, the chain  S = scale * Q K^T -> masked softmax -> dropout -> P V  (transformer.py:109-130:
// MultiHeadAttn with n_head = 1, d_head = 64) and its backward by three flash-style kernels that never write the (T x T)
// probability matrix to HBM:
//   forward            : one workgroup per 64 query rows, online softmax over 64-key blocks, saves logsumexp per row;
//   backward dK / dV   : one workgroup per 64 keys, loops over query blocks, recomputes P from the saved logsumexp;
//   backward dQ        : one workgroup per 64 query rows, loops over key blocks.
// Both backward kernels are atomics-free and deterministic.  Dropout masks are the stateless hash of xva_common.h on the index
// (b * Tp + i) * Tp + j — the same function the unfused path (fp_ops.hip softmax) and the oracle use.
//
// Data layout: qkv (B, Tp, 192) bf16 = [Q | K | V] per padded row; key j of item b is valid iff 1 <= j <= lens[b].
// Tiles of 64 rows x 64 head dims (8 KiB) travel HBM -> LDS by global_load_lds; the LDS image XORs the 32-byte window of a row
// with (row >> 1) & 3, which makes BOTH access patterns conflict-free: ds_read_b128 MFMA fragments along the head dim (Q K^T,
// dO V^T) and ds_read_b64_tr_b16 transposed fragments along the row dim (P V, dS^T Q, P^T dO, dS K).
// MFMA convention used throughout: mfma(X, Y) with X = fragment indexed a, Y = fragment indexed b gives every lane the outputs
// [a = (lane >> 4) * 4 + r][b = lane & 15]; operands are ordered so that b is the dimension whose statistics are lane-local
// (query row in forward / dQ, key in dK / dV) and the probability registers feed the next MFMA without any cross-lane movement
// (the reduction index of that MFMA is permuted identically in both of its operands).
#include "gemm_glds.h"
#include "../../include/xva_hip.h"

namespace {
using namespace xva_glds;

constexpr int TILE = 64 * 64 * 2;   // bytes of one LDS tile

__device__ __forceinline__ uint32_t chunk_off(int r, int c) { return r * 128 + (((((c >> 1) ^ ((r >> 1) & 3)) << 1) | (c & 1)) << 4); }

// DMA a [64 rows][64 cols] bf16 tile (rows row0 .. row0 + 63 clamped to rmax) into LDS; 4 waves x 2 instructions
__device__ __forceinline__ void tile_dma(const uint16_t* base, int64_t ld, int row0, int rmax, XVA_LDS uint8_t* tile, int lane, int wave) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int Q = q * 4 + wave;
        const int r = Q * 8 + (lane >> 3), p = lane & 7;
        const int c = (((p >> 1) ^ ((r >> 1) & 3)) << 1) | (p & 1);
        const uint16_t* src = base + (int64_t)min(row0 + r, rmax) * ld + c * 8;
        __builtin_amdgcn_global_load_lds((const XVA_GLB void*)src, (XVA_LDS void*)(tile + Q * 1024), 16, 0, 0);
    }
}

struct Frag {
    uint32_t kc[2];   // ds_read_b128 fragment (index = row tile, k = head dim): lane offset for kh = 0, 1
    uint32_t tr[4];   // transpose-read fragment (index = column tile dt, k = rows): lane offset for dt = 0..3
    __device__ __forceinline__ void init(int lane) {
        const int g = lane >> 4, i = lane & 15;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) kc[kh] = chunk_off(i, kh * 4 + g);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) tr[dt] = chunk_off(g * 4 + (i >> 2), dt * 2 + ((i & 3) >> 1)) + (i & 1) * 8;
    }
    // rows it*16 .. +15 of the tile as an MFMA operand indexed by row, k = kh*32 + (lane>>4)*8 + e
    __device__ __forceinline__ bf16x8 read_kc(const XVA_LDS uint8_t* tile, int it, int kh) const {
        return *reinterpret_cast<const XVA_LDS bf16x8*>(tile + it * 2048 + kc[kh]);
    }
    // columns dt*16 .. +15 as an MFMA operand indexed by column; k-step t covers rows 32t .. 32t+31 in the PERMUTED order
    // e < 4: row 32t + (lane>>4)*4 + e ; e >= 4: row 32t + 16 + (lane>>4)*4 + (e-4)   (matches pack_rows below)
    __device__ __forceinline__ bf16x8 read_tr(const XVA_LDS uint8_t* tile, int dt, int t) const {
        const XVA_LDS uint8_t* a = tile + t * 4096 + tr[dt];
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((XVA_LDS s16x4*)a);
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((XVA_LDS s16x4*)(a + 2048));
        return __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
};

// two accumulator tiles (rows / keys 32t + g*4 + r and 32t + 16 + g*4 + r) -> one bf16 operand in the permuted k order
// (F16: the 16-bit format of every tensor and MFMA operand of the kernel — false bf16, true IEEE half; gemm_glds.h)
template <bool F16 = false>
__device__ __forceinline__ bf16x8 pack_rows(f32x4 a, f32x4 b) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = {pack2<F16>(a[0], a[1]), pack2<F16>(a[2], a[3]), pack2<F16>(b[0], b[1]), pack2<F16>(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 ld_frag(const uint16_t* p) { return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(p)); }
template <bool F16 = false>
__device__ __forceinline__ void st4(uint16_t* dst, f32x4 v, float s) {
    *reinterpret_cast<uint2*>(dst) = make_uint2(pack2<F16>(v[0] * s, v[1] * s), pack2<F16>(v[2] * s, v[3] * s));
}
#define MFMA(a, b, c) mma16<F16>(a, b, c)

// ================================================================================ forward ====
template <bool DROP, bool F16 = false>
__global__ __launch_bounds__(256, 2) void attn_fwd_kernel(const uint16_t* __restrict__ qkv, const int* __restrict__ lens,
                                                          uint16_t* __restrict__ av, float* __restrict__ lse, int Tp, float scale,
                                                          float pdrop, uint64_t seed, uint32_t stream_id) {
    __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[4 * TILE];   // K0 V0 K1 V1
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;
    const int b = blockIdx.y, q0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const uint16_t* base = qkv + (int64_t)b * Tp * 192;
    const int len = lens[b];
    const int nkb = (len + 1 + 63) / 64;   // keys 1 .. len
    const int row = q0 + wave * 16 + i, rowc = min(row, Tp - 1);
    Frag fr; fr.init(lane);
    bf16x8 qf[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) qf[kh] = ld_frag(base + (int64_t)rowc * 192 + kh * 32 + g * 8);
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;
    const uint64_t drow = ((uint64_t)b * Tp + row) * Tp;

    tile_dma(base + 64, 192, 0, Tp - 1, smem, lane, wave);
    tile_dma(base + 128, 192, 0, Tp - 1, smem + TILE, lane, wave);
    __syncthreads();
    for (int jb = 0; jb < nkb; ++jb) {
        const int cur = jb & 1;
        if (jb + 1 < nkb) {
            tile_dma(base + 64, 192, (jb + 1) * 64, Tp - 1, smem + (cur ^ 1) * 2 * TILE, lane, wave);
            tile_dma(base + 128, 192, (jb + 1) * 64, Tp - 1, smem + (cur ^ 1) * 2 * TILE + TILE, lane, wave);
        }
        const XVA_LDS uint8_t* Kt = smem + cur * 2 * TILE;
        const XVA_LDS uint8_t* Vt = Kt + TILE;
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) s[kt] = MFMA(fr.read_kc(Kt, kt, kh), qf[kh], s[kt]);   // [key g*4+r][row lane&15]
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jb * 64 + kt * 16 + g * 4 + r;
                const float v = (j >= 1 && j <= len) ? s[kt][r] * scale : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m - m_use);
        lsum *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __expf(s[kt][r] - m_use);
                lsum += p;
                if (DROP) p *= xva_dropout_scale(pdrop, seed, stream_id, drow + (uint64_t)(jb * 64 + kt * 16 + g * 4 + r));
                s[kt][r] = p;
            }
        m = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = pack_rows<F16>(s[2 * t], s[2 * t + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) o[dt] = MFMA(fr.read_tr(Vt, dt, t), pf, o[dt]);           // [d g*4+r][row lane&15]
        }
        __syncthreads();
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (row < Tp) {
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        uint16_t* dst = av + ((int64_t)b * Tp + row) * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) st4<F16>(dst + dt * 16, o[dt], inv);
        if (g == 0) lse[(int64_t)b * Tp + row] = lsum > 0.f ? m + __logf(lsum) : INFINITY;
    }
}

// D[row] = sum_d dO[row][d] * O[row][d]   (8 lanes per row)
template <bool F16 = false>
__global__ void attn_bwd_prep_kernel(const uint16_t* __restrict__ O, const uint16_t* __restrict__ dO, float* __restrict__ D, int64_t rows) {
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int c = (threadIdx.x & 7) * 8;
    float s = 0.f;
    if (row < rows) {
        uint4 a = *reinterpret_cast<const uint4*>(O + row * 64 + c), d = *reinterpret_cast<const uint4*>(dO + row * 64 + c);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, dw[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float a0, a1, d0, d1; unpack2<F16>(aw[e], a0, a1); unpack2<F16>(dw[e], d0, d1);
            s += a0 * d0 + a1 * d1;
        }
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (row < rows && (threadIdx.x & 7) == 0) D[row] = s;
}

// ================================================================================ backward: dK, dV ====
template <bool DROP, bool F16 = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dO,
                                                              const float* __restrict__ lse, const float* __restrict__ Dv,
                                                              const int* __restrict__ lens, uint16_t* __restrict__ dqkv, int Tp,
                                                              float scale, float pdrop, uint64_t seed, uint32_t stream_id) {
    __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[4 * TILE + 1024];   // Q0 dO0 Q1 dO1 | logsumexp / D of the two query blocks
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;
    const int b = blockIdx.y, k0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const uint16_t* base = qkv + (int64_t)b * Tp * 192;
    const uint16_t* dob = dO + (int64_t)b * Tp * 64;
    const int len = lens[b];
    const int key = k0 + wave * 16 + i, keyc = min(key, Tp - 1);
    uint16_t* outk = dqkv + ((int64_t)b * Tp + key) * 192 + 64 + g * 4;
    if (k0 > len) {   // a block of dead keys (uniform per workgroup): zero gradients
        if (key < Tp) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { st4<F16>(outk + dt * 16, z, 1.f); st4<F16>(outk + 64 + dt * 16, z, 1.f); }
        }
        return;
    }
    const bool key_ok = key >= 1 && key <= len;
    Frag fr; fr.init(lane);
    bf16x8 kf[2], vf[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        kf[kh] = ld_frag(base + (int64_t)keyc * 192 + 64 + kh * 32 + g * 8);
        vf[kh] = ld_frag(base + (int64_t)keyc * 192 + 128 + kh * 32 + g * 8);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int nqb = (min(len + 1, Tp) + 63) / 64;   // rows above len carry dO = 0: no contribution
    const float* Lb = lse + (int64_t)b * Tp;
    const float* Db = Dv + (int64_t)b * Tp;
    // per-row statistics of a query block travel with its tiles: wave 0 DMAs 64 logsumexp values, wave 1 the 64 D values (4 bytes per lane)
    XVA_LDS uint8_t* stat = smem + 4 * TILE;
    auto stat_dma = [&](int ib, int buf) {
        if (wave < 2) {
            const float* src = (wave == 0 ? Lb : Db) + min(ib * 64 + lane, Tp - 1);
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)src, (XVA_LDS void*)(stat + buf * 512 + wave * 256), 4, 0, 0);
        }
    };

    tile_dma(base, 192, 0, Tp - 1, smem, lane, wave);
    tile_dma(dob, 64, 0, Tp - 1, smem + TILE, lane, wave);
    stat_dma(0, 0);
    __syncthreads();
    for (int ib = 0; ib < nqb; ++ib) {
        const int cur = ib & 1;
        if (ib + 1 < nqb) {
            tile_dma(base, 192, (ib + 1) * 64, Tp - 1, smem + (cur ^ 1) * 2 * TILE, lane, wave);
            tile_dma(dob, 64, (ib + 1) * 64, Tp - 1, smem + (cur ^ 1) * 2 * TILE + TILE, lane, wave);
            stat_dma(ib + 1, cur ^ 1);
        }
        const XVA_LDS uint8_t* Qt = smem + cur * 2 * TILE;
        const XVA_LDS uint8_t* Ot = Qt + TILE;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            s[rt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                s[rt] = MFMA(fr.read_kc(Qt, rt, kh), kf[kh], s[rt]);      // [row g*4+r][key lane&15]
                dp[rt] = MFMA(fr.read_kc(Ot, rt, kh), vf[kh], dp[rt]);    // dPd = dO V^T
            }
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const f32x4 L4 = *reinterpret_cast<const XVA_LDS f32x4*>(stat + cur * 512 + (rt * 16 + g * 4) * 4);
            const f32x4 D4 = *reinterpret_cast<const XVA_LDS f32x4*>(stat + cur * 512 + 256 + (rt * 16 + g * 4) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = ib * 64 + rt * 16 + g * 4 + r;
                const bool ok = key_ok && ir < Tp;
                const float p = ok ? __expf(s[rt][r] * scale - L4[r]) : 0.f;
                float dr = 1.f;
                if (DROP) dr = xva_dropout_scale(pdrop, seed, stream_id, ((uint64_t)b * Tp + ir) * Tp + key);
                s[rt][r] = p * dr;                                           // dropped probabilities
                dp[rt][r] = p * (dp[rt][r] * dr - D4[r]) * scale;            // dS (scale of S folded in)
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 pf = pack_rows<F16>(s[2 * t], s[2 * t + 1]);
            const bf16x8 df = pack_rows<F16>(dp[2 * t], dp[2 * t + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                dv[dt] = MFMA(fr.read_tr(Ot, dt, t), pf, dv[dt]);            // dV[d][key] += dO[row][d] Pd[row][key]
                dk[dt] = MFMA(fr.read_tr(Qt, dt, t), df, dk[dt]);            // dK[d][key] += Q[row][d] dS[row][key]
            }
        }
        __syncthreads();
    }
    if (key < Tp) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { st4<F16>(outk + dt * 16, dk[dt], 1.f); st4<F16>(outk + 64 + dt * 16, dv[dt], 1.f); }
    }
}

// ================================================================================ backward: dQ ====
template <bool DROP, bool F16 = false>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const uint16_t* __restrict__ qkv, const uint16_t* __restrict__ dO,
                                                             const float* __restrict__ lse, const float* __restrict__ Dv,
                                                             const int* __restrict__ lens, uint16_t* __restrict__ dqkv, int Tp,
                                                             float scale, float pdrop, uint64_t seed, uint32_t stream_id) {
    __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[4 * TILE];   // K0 V0 K1 V1
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;
    const int b = blockIdx.y, q0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const uint16_t* base = qkv + (int64_t)b * Tp * 192;
    const int len = lens[b];
    const int nkb = (len + 1 + 63) / 64;
    const int row = q0 + wave * 16 + i, rowc = min(row, Tp - 1);
    Frag fr; fr.init(lane);
    bf16x8 qf[2], of[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        qf[kh] = ld_frag(base + (int64_t)rowc * 192 + kh * 32 + g * 8);
        of[kh] = ld_frag(dO + ((int64_t)b * Tp + rowc) * 64 + kh * 32 + g * 8);
    }
    const float L = lse[(int64_t)b * Tp + rowc], Dr = Dv[(int64_t)b * Tp + rowc];
    const uint64_t drow = ((uint64_t)b * Tp + row) * Tp;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    tile_dma(base + 64, 192, 0, Tp - 1, smem, lane, wave);
    tile_dma(base + 128, 192, 0, Tp - 1, smem + TILE, lane, wave);
    __syncthreads();
    for (int jb = 0; jb < nkb; ++jb) {
        const int cur = jb & 1;
        if (jb + 1 < nkb) {
            tile_dma(base + 64, 192, (jb + 1) * 64, Tp - 1, smem + (cur ^ 1) * 2 * TILE, lane, wave);
            tile_dma(base + 128, 192, (jb + 1) * 64, Tp - 1, smem + (cur ^ 1) * 2 * TILE + TILE, lane, wave);
        }
        const XVA_LDS uint8_t* Kt = smem + cur * 2 * TILE;
        const XVA_LDS uint8_t* Vt = Kt + TILE;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                s[kt] = MFMA(fr.read_kc(Kt, kt, kh), qf[kh], s[kt]);      // [key g*4+r][row lane&15]
                dp[kt] = MFMA(fr.read_kc(Vt, kt, kh), of[kh], dp[kt]);    // dPd
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jb * 64 + kt * 16 + g * 4 + r;
                const float p = (j >= 1 && j <= len) ? __expf(s[kt][r] * scale - L) : 0.f;
                float dr = 1.f;
                if (DROP) dr = xva_dropout_scale(pdrop, seed, stream_id, drow + (uint64_t)j);
                dp[kt][r] = p * (dp[kt][r] * dr - Dr) * scale;
            }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bf16x8 df = pack_rows<F16>(dp[2 * t], dp[2 * t + 1]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) dq[dt] = MFMA(fr.read_tr(Kt, dt, t), df, dq[dt]);   // dQ[d][row] += K[key][d] dS[row][key]
        }
        __syncthreads();
    }
    if (row < Tp) {
        uint16_t* dst = dqkv + ((int64_t)b * Tp + row) * 192 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) st4<F16>(dst + dt * 16, dq[dt], 1.f);
    }
}
#undef MFMA
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)


// ================================================================================ split-bf16 pairs (fp32 mode, split products) ====
// The same three kernels for tensors stored as split-bf16 PAIRS (include/xva_gemm.h: hi = bf16(x), lo = bf16(x - hi), the lo plane `plane` elements after the
// hi plane): every product is hi.hi + hi.lo + lo.hi in the fp32 accumulator — Q K^T, dO V^T from the stored pairs, P V, dS^T Q, P^T dO, dS K with the
// probabilities / score gradients split in registers — and every output leaves as a pair.  Replaces, for the FastPitch split-products mode, the unfused
// chain (scores and probabilities through HBM in fp32: two T x T tensors per layer and direction) of fastpitch_engine.hip:attention_*_planes.
// LDS: the hi and lo images of both streamed tiles, double buffered = 64 KB (+ 1 KB of row statistics in dK / dV): two workgroups per CU.
using xva_gemm_impl::split_bf2;
__device__ __forceinline__ void pack_rows_split(f32x4 a, f32x4 b, bf16x8& hi, bf16x8& lo) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
    split_bf2(a[0], a[1], h0, l0); split_bf2(a[2], a[3], h1, l1);
    split_bf2(b[0], b[1], h2, l2); split_bf2(b[2], b[3], h3, l3);
    const u32x4 h = {h0, h1, h2, h3}, l = {l0, l1, l2, l3};
    hi = __builtin_bit_cast(bf16x8, h); lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ void st4_pair(uint16_t* dst, int64_t plane, f32x4 v, float s) {
    uint32_t h0, l0, h1, l1;
    split_bf2(v[0] * s, v[1] * s, h0, l0); split_bf2(v[2] * s, v[3] * s, h1, l1);
    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(dst + plane) = make_uint2(l0, l1);
}
// acc += Xh Yh + Xh Yl + Xl Yh (small terms first)
#define MFMA3(xh, xl, yh, yl, acc) do { acc = MFMA(xl, yh, acc); acc = MFMA(xh, yl, acc); acc = MFMA(xh, yh, acc); } while (0)
constexpr int SPLIT_LDS = 8 * TILE;

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_fwd_split_kernel(const uint16_t* __restrict__ qkv, int64_t qplane, const int* __restrict__ lens,
                                                                uint16_t* __restrict__ av, int64_t avplane, float* __restrict__ lse, int Tp, float scale,
                                                                float pdrop, uint64_t seed, uint32_t stream_id) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_dyn[];   // 2 x {Kh Kl Vh Vl}
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_dyn;
    const int b = blockIdx.y, q0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const uint16_t* base = qkv + (int64_t)b * Tp * 192;
    const uint16_t* basl = base + qplane;
    const int len = lens[b];
    const int nkb = (len + 1 + 63) / 64;   // keys 1 .. len
    const int row = q0 + wave * 16 + i, rowc = min(row, Tp - 1);
    Frag fr; fr.init(lane);
    bf16x8 qh[2], ql[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        qh[kh] = ld_frag(base + (int64_t)rowc * 192 + kh * 32 + g * 8);
        ql[kh] = ld_frag(basl + (int64_t)rowc * 192 + kh * 32 + g * 8);
    }
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, lsum = 0.f;
    const uint64_t drow = ((uint64_t)b * Tp + row) * Tp;
    auto dma = [&](int jb, int buf) {
        XVA_LDS uint8_t* t = smem + buf * 4 * TILE;
        tile_dma(base + 64, 192, jb * 64, Tp - 1, t, lane, wave);
        tile_dma(basl + 64, 192, jb * 64, Tp - 1, t + TILE, lane, wave);
        tile_dma(base + 128, 192, jb * 64, Tp - 1, t + 2 * TILE, lane, wave);
        tile_dma(basl + 128, 192, jb * 64, Tp - 1, t + 3 * TILE, lane, wave);
    };
    dma(0, 0);
    __syncthreads();
    for (int jb = 0; jb < nkb; ++jb) {
        const int cur = jb & 1;
        if (jb + 1 < nkb) dma(jb + 1, cur ^ 1);
        const XVA_LDS uint8_t* Kh = smem + cur * 4 * TILE;
        const XVA_LDS uint8_t* Kl = Kh + TILE;
        const XVA_LDS uint8_t* Vh = Kh + 2 * TILE;
        const XVA_LDS uint8_t* Vl = Kh + 3 * TILE;
        f32x4 s[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const bf16x8 kfh = fr.read_kc(Kh, kt, kh), kfl = fr.read_kc(Kl, kt, kh);
                MFMA3(kfh, kfl, qh[kh], ql[kh], s[kt]);                                              // [key g*4+r][row lane&15]
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jb * 64 + kt * 16 + g * 4 + r;
                const float v = (j >= 1 && j <= len) ? s[kt][r] * scale : -INFINITY;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float m_use = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = expf(m - m_use);
        lsum *= alpha;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = expf(s[kt][r] - m_use);
                lsum += p;
                if (DROP) p *= xva_dropout_scale(pdrop, seed, stream_id, drow + (uint64_t)(jb * 64 + kt * 16 + g * 4 + r));
                s[kt][r] = p;
            }
        m = m_new;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 ph, pl;
            pack_rows_split(s[2 * t], s[2 * t + 1], ph, pl);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 vfh = fr.read_tr(Vh, dt, t), vfl = fr.read_tr(Vl, dt, t);
                MFMA3(vfh, vfl, ph, pl, o[dt]);                                                      // [d g*4+r][row lane&15]
            }
        }
        __syncthreads();
    }
    lsum += __shfl_xor(lsum, 16, 64);
    lsum += __shfl_xor(lsum, 32, 64);
    if (row < Tp) {
        const float inv = lsum > 0.f ? 1.f / lsum : 0.f;
        uint16_t* dst = av + ((int64_t)b * Tp + row) * 64 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) st4_pair(dst + dt * 16, avplane, o[dt], inv);
        if (g == 0) lse[(int64_t)b * Tp + row] = lsum > 0.f ? m + logf(lsum) : INFINITY;
    }
}

// D[row] = sum_d dO[row][d] * O[row][d] of the pairs' values (8 lanes per row)
__global__ void attn_bwd_prep_split_kernel(const uint16_t* __restrict__ O, int64_t oplane, const uint16_t* __restrict__ dO, int64_t dplane,
                                           float* __restrict__ D, int64_t rows) {
    const int64_t row = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3);
    const int c = (threadIdx.x & 7) * 8;
    float s = 0.f;
    if (row < rows) {
        const uint4 a = *reinterpret_cast<const uint4*>(O + row * 64 + c), al = *reinterpret_cast<const uint4*>(O + oplane + row * 64 + c);
        const uint4 d = *reinterpret_cast<const uint4*>(dO + row * 64 + c), dl = *reinterpret_cast<const uint4*>(dO + dplane + row * 64 + c);
        const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, alw[4] = {al.x, al.y, al.z, al.w}, dw[4] = {d.x, d.y, d.z, d.w}, dlw[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float o0 = __uint_as_float(aw[e] << 16) + __uint_as_float(alw[e] << 16), o1 = __uint_as_float(aw[e] & 0xffff0000u) + __uint_as_float(alw[e] & 0xffff0000u);
            const float g0 = __uint_as_float(dw[e] << 16) + __uint_as_float(dlw[e] << 16), g1 = __uint_as_float(dw[e] & 0xffff0000u) + __uint_as_float(dlw[e] & 0xffff0000u);
            s += o0 * g0 + o1 * g1;
        }
    }
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    if (row < rows && (threadIdx.x & 7) == 0) D[row] = s;
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_split_kernel(const uint16_t* __restrict__ qkv, int64_t qplane, const uint16_t* __restrict__ dO,
                                                                    int64_t dplane, const float* __restrict__ lse, const float* __restrict__ Dv,
                                                                    const int* __restrict__ lens, uint16_t* __restrict__ dqkv, int64_t gplane, int Tp,
                                                                    float scale, float pdrop, uint64_t seed, uint32_t stream_id) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_dyn[];   // 2 x {Qh Ql dOh dOl} | logsumexp / D of the two query blocks
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_dyn;
    const int b = blockIdx.y, k0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const uint16_t* base = qkv + (int64_t)b * Tp * 192;
    const uint16_t* basl = base + qplane;
    const uint16_t* dob = dO + (int64_t)b * Tp * 64;
    const uint16_t* dol = dob + dplane;
    const int len = lens[b];
    const int key = k0 + wave * 16 + i, keyc = min(key, Tp - 1);
    uint16_t* outk = dqkv + ((int64_t)b * Tp + key) * 192 + 64 + g * 4;
    if (k0 > len) {   // a block of dead keys (uniform per workgroup): zero gradients
        if (key < Tp) {
            const f32x4 z = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) { st4_pair(outk + dt * 16, gplane, z, 1.f); st4_pair(outk + 64 + dt * 16, gplane, z, 1.f); }
        }
        return;
    }
    const bool key_ok = key >= 1 && key <= len;
    Frag fr; fr.init(lane);
    bf16x8 kfh[2], kfl[2], vfh[2], vfl[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        kfh[kh] = ld_frag(base + (int64_t)keyc * 192 + 64 + kh * 32 + g * 8);
        kfl[kh] = ld_frag(basl + (int64_t)keyc * 192 + 64 + kh * 32 + g * 8);
        vfh[kh] = ld_frag(base + (int64_t)keyc * 192 + 128 + kh * 32 + g * 8);
        vfl[kh] = ld_frag(basl + (int64_t)keyc * 192 + 128 + kh * 32 + g * 8);
    }
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) { dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const int nqb = (min(len + 1, Tp) + 63) / 64;   // rows above len carry dO = 0: no contribution
    const float* Lb = lse + (int64_t)b * Tp;
    const float* Db = Dv + (int64_t)b * Tp;
    XVA_LDS uint8_t* stat = smem + SPLIT_LDS;
    auto dma = [&](int ib, int buf) {
        XVA_LDS uint8_t* t = smem + buf * 4 * TILE;
        tile_dma(base, 192, ib * 64, Tp - 1, t, lane, wave);
        tile_dma(basl, 192, ib * 64, Tp - 1, t + TILE, lane, wave);
        tile_dma(dob, 64, ib * 64, Tp - 1, t + 2 * TILE, lane, wave);
        tile_dma(dol, 64, ib * 64, Tp - 1, t + 3 * TILE, lane, wave);
        if (wave < 2) {
            const float* src = (wave == 0 ? Lb : Db) + min(ib * 64 + lane, Tp - 1);
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)src, (XVA_LDS void*)(stat + buf * 512 + wave * 256), 4, 0, 0);
        }
    };
    dma(0, 0);
    __syncthreads();
    for (int ib = 0; ib < nqb; ++ib) {
        const int cur = ib & 1;
        if (ib + 1 < nqb) dma(ib + 1, cur ^ 1);
        const XVA_LDS uint8_t* Qh = smem + cur * 4 * TILE;
        const XVA_LDS uint8_t* Ql = Qh + TILE;
        const XVA_LDS uint8_t* Oh = Qh + 2 * TILE;
        const XVA_LDS uint8_t* Ol = Qh + 3 * TILE;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            s[rt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const bf16x8 a = fr.read_kc(Qh, rt, kh), al = fr.read_kc(Ql, rt, kh);
                MFMA3(a, al, kfh[kh], kfl[kh], s[rt]);                       // [row g*4+r][key lane&15]
                const bf16x8 d = fr.read_kc(Oh, rt, kh), dl = fr.read_kc(Ol, rt, kh);
                MFMA3(d, dl, vfh[kh], vfl[kh], dp[rt]);                      // dPd = dO V^T
            }
        }
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) {
            const f32x4 L4 = *reinterpret_cast<const XVA_LDS f32x4*>(stat + cur * 512 + (rt * 16 + g * 4) * 4);
            const f32x4 D4 = *reinterpret_cast<const XVA_LDS f32x4*>(stat + cur * 512 + 256 + (rt * 16 + g * 4) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ir = ib * 64 + rt * 16 + g * 4 + r;
                const bool ok = key_ok && ir < Tp;
                const float p = ok ? expf(s[rt][r] * scale - L4[r]) : 0.f;
                float dr = 1.f;
                if (DROP) dr = xva_dropout_scale(pdrop, seed, stream_id, ((uint64_t)b * Tp + ir) * Tp + key);
                s[rt][r] = p * dr;                                           // dropped probabilities
                dp[rt][r] = p * (dp[rt][r] * dr - D4[r]) * scale;            // dS (scale of S folded in)
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 ph, pl, dh, dl;
            pack_rows_split(s[2 * t], s[2 * t + 1], ph, pl);
            pack_rows_split(dp[2 * t], dp[2 * t + 1], dh, dl);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 oth = fr.read_tr(Oh, dt, t), otl = fr.read_tr(Ol, dt, t);
                MFMA3(oth, otl, ph, pl, dv[dt]);                             // dV[d][key] += dO[row][d] Pd[row][key]
                const bf16x8 qth = fr.read_tr(Qh, dt, t), qtl = fr.read_tr(Ql, dt, t);
                MFMA3(qth, qtl, dh, dl, dk[dt]);                             // dK[d][key] += Q[row][d] dS[row][key]
            }
        }
        __syncthreads();
    }
    if (key < Tp) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) { st4_pair(outk + dt * 16, gplane, dk[dt], 1.f); st4_pair(outk + 64 + dt * 16, gplane, dv[dt], 1.f); }
    }
}

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_split_kernel(const uint16_t* __restrict__ qkv, int64_t qplane, const uint16_t* __restrict__ dO,
                                                                   int64_t dplane, const float* __restrict__ lse, const float* __restrict__ Dv,
                                                                   const int* __restrict__ lens, uint16_t* __restrict__ dqkv, int64_t gplane, int Tp,
                                                                   float scale, float pdrop, uint64_t seed, uint32_t stream_id) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_dyn[];   // 2 x {Kh Kl Vh Vl}
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_dyn;
    const int b = blockIdx.y, q0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, i = lane & 15;
    const uint16_t* base = qkv + (int64_t)b * Tp * 192;
    const uint16_t* basl = base + qplane;
    const int len = lens[b];
    const int nkb = (len + 1 + 63) / 64;
    const int row = q0 + wave * 16 + i, rowc = min(row, Tp - 1);
    Frag fr; fr.init(lane);
    bf16x8 qh[2], ql[2], oh[2], ol[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        qh[kh] = ld_frag(base + (int64_t)rowc * 192 + kh * 32 + g * 8);
        ql[kh] = ld_frag(basl + (int64_t)rowc * 192 + kh * 32 + g * 8);
        oh[kh] = ld_frag(dO + ((int64_t)b * Tp + rowc) * 64 + kh * 32 + g * 8);
        ol[kh] = ld_frag(dO + dplane + ((int64_t)b * Tp + rowc) * 64 + kh * 32 + g * 8);
    }
    const float L = lse[(int64_t)b * Tp + rowc], Dr = Dv[(int64_t)b * Tp + rowc];
    const uint64_t drow = ((uint64_t)b * Tp + row) * Tp;
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto dma = [&](int jb, int buf) {
        XVA_LDS uint8_t* t = smem + buf * 4 * TILE;
        tile_dma(base + 64, 192, jb * 64, Tp - 1, t, lane, wave);
        tile_dma(basl + 64, 192, jb * 64, Tp - 1, t + TILE, lane, wave);
        tile_dma(base + 128, 192, jb * 64, Tp - 1, t + 2 * TILE, lane, wave);
        tile_dma(basl + 128, 192, jb * 64, Tp - 1, t + 3 * TILE, lane, wave);
    };
    dma(0, 0);
    __syncthreads();
    for (int jb = 0; jb < nkb; ++jb) {
        const int cur = jb & 1;
        if (jb + 1 < nkb) dma(jb + 1, cur ^ 1);
        const XVA_LDS uint8_t* Kh = smem + cur * 4 * TILE;
        const XVA_LDS uint8_t* Kl = Kh + TILE;
        const XVA_LDS uint8_t* Vh = Kh + 2 * TILE;
        const XVA_LDS uint8_t* Vl = Kh + 3 * TILE;
        f32x4 s[4], dp[4];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
            s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f}; dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                const bf16x8 kfh = fr.read_kc(Kh, kt, kh), kfl = fr.read_kc(Kl, kt, kh);
                MFMA3(kfh, kfl, qh[kh], ql[kh], s[kt]);                      // [key g*4+r][row lane&15]
                const bf16x8 vfh = fr.read_kc(Vh, kt, kh), vfl = fr.read_kc(Vl, kt, kh);
                MFMA3(vfh, vfl, oh[kh], ol[kh], dp[kt]);                     // dPd
            }
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = jb * 64 + kt * 16 + g * 4 + r;
                const float p = (j >= 1 && j <= len) ? expf(s[kt][r] * scale - L) : 0.f;
                float dr = 1.f;
                if (DROP) dr = xva_dropout_scale(pdrop, seed, stream_id, drow + (uint64_t)j);
                dp[kt][r] = p * (dp[kt][r] * dr - Dr) * scale;
            }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            bf16x8 dh, dl;
            pack_rows_split(dp[2 * t], dp[2 * t + 1], dh, dl);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 kth = fr.read_tr(Kh, dt, t), ktl = fr.read_tr(Kl, dt, t);
                MFMA3(kth, ktl, dh, dl, dq[dt]);                             // dQ[d][row] += K[key][d] dS[row][key]
            }
        }
        __syncthreads();
    }
    if (row < Tp) {
        uint16_t* dst = dqkv + ((int64_t)b * Tp + row) * 192 + g * 4;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) st4_pair(dst + dt * 16, gplane, dq[dt], 1.f);
    }
}

template <typename K>
static int raise_lds(K kernel, int bytes, bool& done) {
    if (done) return XVA_OK;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) {
        xva_set_error("attention (split pairs): cannot raise the dynamic LDS limit"); return XVA_ERR_HIP;
    }
    done = true;
    return XVA_OK;
}

}  // namespace

// av (B, Tp, 64) = dropout(softmax(scale * Q K^T, keys 1..len)) V ; lse (B, Tp) fp32 = logsumexp of the masked scaled scores
template <bool F16>
static int attention_fwd_any(const void* qkv, const int32_t* lens, void* av, float* lse, int B, int Tp, float scale, float p_drop, uint64_t seed, uint32_t stream_id,
                             void* stream) {
    XVA_CHECK_ARG(qkv && lens && av && lse && B > 0 && Tp > 0, "attention_fwd: bad args");
    XVA_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)av % 8) == 0, "attention_fwd: misaligned tensors");
    dim3 grid(xva_cdiv(Tp, 64), B), block(256);
    if (p_drop > 0.f)
        hipLaunchKernelGGL((attn_fwd_kernel<true, F16>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)qkv, lens, (uint16_t*)av, lse, Tp,
                           scale, p_drop, seed, stream_id);
    else
        hipLaunchKernelGGL((attn_fwd_kernel<false, F16>), grid, block, 0, (hipStream_t)stream, (const uint16_t*)qkv, lens, (uint16_t*)av, lse, Tp,
                           scale, p_drop, seed, stream_id);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_attention_fwd(const void* qkv, const int32_t* lens, void* av, float* lse, int B, int Tp, float scale,
                                    float p_drop, uint64_t seed, uint32_t stream_id, void* stream) {
    return attention_fwd_any<false>(qkv, lens, av, lse, B, Tp, scale, p_drop, seed, stream_id, stream);
}

// d_qkv (B, Tp, 192) = gradients of Q | K | V given d_av; `dscratch` holds B * Tp floats
template <bool F16>
static int attention_bwd_any(const void* qkv, const void* av, const void* d_av, const float* lse, float* dscratch,
                                    const int32_t* lens, void* d_qkv, int B, int Tp, float scale, float p_drop, uint64_t seed,
                                    uint32_t stream_id, void* stream) {
    XVA_CHECK_ARG(qkv && av && d_av && lse && dscratch && lens && d_qkv && B > 0 && Tp > 0, "attention_bwd: bad args");
    XVA_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)av % 16) == 0 && ((uintptr_t)d_av % 16) == 0 && ((uintptr_t)d_qkv % 8) == 0,
                  "attention_bwd: misaligned tensors");
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL((attn_bwd_prep_kernel<F16>), dim3((unsigned)xva_cdiv(rows, 32)), dim3(256), 0, st, (const uint16_t*)av, (const uint16_t*)d_av,
                       dscratch, rows);
    dim3 grid(xva_cdiv(Tp, 64), B), block(256);
    if (p_drop > 0.f) {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<true, F16>), grid, block, 0, st, (const uint16_t*)qkv, (const uint16_t*)d_av, lse, dscratch, lens,
                           (uint16_t*)d_qkv, Tp, scale, p_drop, seed, stream_id);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<true, F16>), grid, block, 0, st, (const uint16_t*)qkv, (const uint16_t*)d_av, lse, dscratch, lens,
                           (uint16_t*)d_qkv, Tp, scale, p_drop, seed, stream_id);
    } else {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<false, F16>), grid, block, 0, st, (const uint16_t*)qkv, (const uint16_t*)d_av, lse, dscratch, lens,
                           (uint16_t*)d_qkv, Tp, scale, p_drop, seed, stream_id);
        hipLaunchKernelGGL((attn_bwd_dq_kernel<false, F16>), grid, block, 0, st, (const uint16_t*)qkv, (const uint16_t*)d_av, lse, dscratch, lens,
                           (uint16_t*)d_qkv, Tp, scale, p_drop, seed, stream_id);
    }
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_attention_bwd(const void* qkv, const void* av, const void* d_av, const float* lse, float* dscratch,
                                    const int32_t* lens, void* d_qkv, int B, int Tp, float scale, float p_drop, uint64_t seed,
                                    uint32_t stream_id, void* stream) {
    return attention_bwd_any<false>(qkv, av, d_av, lse, dscratch, lens, d_qkv, B, Tp, scale, p_drop, seed, stream_id, stream);
}

// The same on split-bf16 pairs (fp32 mode with split products): qkv / av / d_av / d_qkv are the hi planes, their lo planes `*_plane` ELEMENTS after them.
// Plane offsets of 0 (all of them): the tensors are single IEEE-half tensors (XVA_F16) — the fp16-operand mode of the same schedule (include/xva_hip.h).
extern "C" int xva_fp_attention_fwd_pairs(const void* qkv, int64_t qkv_plane, const int32_t* lens, void* av, int64_t av_plane, float* lse, int B, int Tp,
                                          float scale, float p_drop, uint64_t seed, uint32_t stream_id, void* stream) {
    if (qkv_plane == 0 && av_plane == 0) return attention_fwd_any<true>(qkv, lens, av, lse, B, Tp, scale, p_drop, seed, stream_id, stream);
    XVA_CHECK_ARG(qkv && lens && av && lse && B > 0 && Tp > 0, "attention_fwd_pairs: bad args");
    XVA_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)av % 8) == 0 && qkv_plane % 8 == 0 && av_plane % 4 == 0, "attention_fwd_pairs: misaligned tensors");
    static bool a0 = false, a1 = false;
    XVA_TRY(raise_lds(attn_fwd_split_kernel<true>, SPLIT_LDS, a0));
    XVA_TRY(raise_lds(attn_fwd_split_kernel<false>, SPLIT_LDS, a1));
    dim3 grid(xva_cdiv(Tp, 64), B), block(256);
    if (p_drop > 0.f)
        hipLaunchKernelGGL((attn_fwd_split_kernel<true>), grid, block, SPLIT_LDS, (hipStream_t)stream, (const uint16_t*)qkv, qkv_plane, lens, (uint16_t*)av,
                           av_plane, lse, Tp, scale, p_drop, seed, stream_id);
    else
        hipLaunchKernelGGL((attn_fwd_split_kernel<false>), grid, block, SPLIT_LDS, (hipStream_t)stream, (const uint16_t*)qkv, qkv_plane, lens, (uint16_t*)av,
                           av_plane, lse, Tp, scale, p_drop, seed, stream_id);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

extern "C" int xva_fp_attention_bwd_pairs(const void* qkv, int64_t qkv_plane, const void* av, int64_t av_plane, const void* d_av, int64_t d_av_plane,
                                          const float* lse, float* dscratch, const int32_t* lens, void* d_qkv, int64_t d_qkv_plane, int B, int Tp, float scale,
                                          float p_drop, uint64_t seed, uint32_t stream_id, void* stream) {
    if (qkv_plane == 0 && av_plane == 0 && d_av_plane == 0 && d_qkv_plane == 0)
        return attention_bwd_any<true>(qkv, av, d_av, lse, dscratch, lens, d_qkv, B, Tp, scale, p_drop, seed, stream_id, stream);
    XVA_CHECK_ARG(qkv && av && d_av && lse && dscratch && lens && d_qkv && B > 0 && Tp > 0, "attention_bwd_pairs: bad args");
    XVA_CHECK_ARG(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)av % 16) == 0 && ((uintptr_t)d_av % 16) == 0 && ((uintptr_t)d_qkv % 8) == 0 && qkv_plane % 8 == 0 &&
                      av_plane % 8 == 0 && d_av_plane % 8 == 0 && d_qkv_plane % 4 == 0, "attention_bwd_pairs: misaligned tensors");
    static bool a0 = false, a1 = false, a2 = false, a3 = false;
    XVA_TRY(raise_lds(attn_bwd_dkv_split_kernel<true>, SPLIT_LDS + 1024, a0));
    XVA_TRY(raise_lds(attn_bwd_dkv_split_kernel<false>, SPLIT_LDS + 1024, a1));
    XVA_TRY(raise_lds(attn_bwd_dq_split_kernel<true>, SPLIT_LDS, a2));
    XVA_TRY(raise_lds(attn_bwd_dq_split_kernel<false>, SPLIT_LDS, a3));
    hipStream_t st = (hipStream_t)stream;
    const int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(attn_bwd_prep_split_kernel, dim3((unsigned)xva_cdiv(rows, 32)), dim3(256), 0, st, (const uint16_t*)av, av_plane, (const uint16_t*)d_av,
                       d_av_plane, dscratch, rows);
    dim3 grid(xva_cdiv(Tp, 64), B), block(256);
    if (p_drop > 0.f) {
        hipLaunchKernelGGL((attn_bwd_dkv_split_kernel<true>), grid, block, SPLIT_LDS + 1024, st, (const uint16_t*)qkv, qkv_plane, (const uint16_t*)d_av, d_av_plane,
                           lse, dscratch, lens, (uint16_t*)d_qkv, d_qkv_plane, Tp, scale, p_drop, seed, stream_id);
        hipLaunchKernelGGL((attn_bwd_dq_split_kernel<true>), grid, block, SPLIT_LDS, st, (const uint16_t*)qkv, qkv_plane, (const uint16_t*)d_av, d_av_plane, lse,
                           dscratch, lens, (uint16_t*)d_qkv, d_qkv_plane, Tp, scale, p_drop, seed, stream_id);
    } else {
        hipLaunchKernelGGL((attn_bwd_dkv_split_kernel<false>), grid, block, SPLIT_LDS + 1024, st, (const uint16_t*)qkv, qkv_plane, (const uint16_t*)d_av, d_av_plane,
                           lse, dscratch, lens, (uint16_t*)d_qkv, d_qkv_plane, Tp, scale, p_drop, seed, stream_id);
        hipLaunchKernelGGL((attn_bwd_dq_split_kernel<false>), grid, block, SPLIT_LDS, st, (const uint16_t*)qkv, qkv_plane, (const uint16_t*)d_av, d_av_plane, lse,
                           dscratch, lens, (uint16_t*)d_qkv, d_qkv_plane, Tp, scale, p_drop, seed, stream_id);
    }
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
