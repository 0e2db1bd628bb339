// fp_ops.hip — the non-GEMM kernels of the FastPitch forward/backward hot loop (gfx950).
//
// All sequence tensors use the padded token-major layout (B, Tp = T + 2, C), fp32: row 0 and
// row Tp-1 of every item are structural zero rows (they ARE the zero padding of the k=3
// convolutions, so a conv is a plain overlapping-row GEMM), rows 1..len are live.  Every
// kernel here is HBM-bound: one pass over its operands, float4 / wave-per-row accesses,
// reductions by wave64 shuffles.  Reference call sites are cited per kernel.
#include "xva_common.h"
#include <type_traits>
#include "../../include/xva_gemm.h"
#include "../../include/xva_hip.h"

#define WAVES_PER_BLOCK 4

// activation element access: dt = XVA_F32 (parity mode), XVA_BF16 (bf16 training mode) or XVA_F16 (IEEE half: the single-plane operand copies of the
// fp16-operand mode); arithmetic is always fp32
typedef _Float16 xva_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_f16(float x, float y) { const xva_h2 h = {(_Float16)x, (_Float16)y}; return __builtin_bit_cast(uint32_t, h); }
__device__ __forceinline__ float a_ld(const void* p, int64_t i, int dt) {
    if (dt == XVA_F16) return (float)reinterpret_cast<const _Float16*>(p)[i];
    return dt == XVA_BF16 ? __uint_as_float(((uint32_t) reinterpret_cast<const uint16_t*>(p)[i]) << 16)
                          : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void a_st(void* p, int64_t i, int dt, float v) {
    if (dt == XVA_F16) { reinterpret_cast<_Float16*>(p)[i] = (_Float16)v; return; }
    if (dt == XVA_BF16) {
        uint32_t u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);
        reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)(u >> 16);
    } else {
        reinterpret_cast<float*>(p)[i] = v;
    }
}

// two adjacent elements (i even): one 4-byte (bf16) or 8-byte (fp32) access
__device__ __forceinline__ void a_ld2(const void* p, int64_t i, int dt, float& x, float& y) {
    if (dt == XVA_F16) {
        const xva_h2 h = *reinterpret_cast<const xva_h2*>(reinterpret_cast<const uint16_t*>(p) + i);
        x = (float)h[0]; y = (float)h[1];
    } else if (dt == XVA_BF16) {
        uint32_t u = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(p) + i);
        x = __uint_as_float(u << 16); y = __uint_as_float(u & 0xffff0000u);
    } else {
        float2 f = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(p) + i);
        x = f.x; y = f.y;
    }
}
__device__ __forceinline__ void a_st2(void* p, int64_t i, int dt, float x, float y) {
    if (dt == XVA_F16) {
        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(p) + i) = pk_f16(x, y);
    } else if (dt == XVA_BF16) {
        uint32_t a = __float_as_uint(x), b = __float_as_uint(y);
        a += 0x7fffu + ((a >> 16) & 1u); b += 0x7fffu + ((b >> 16) & 1u);
        *reinterpret_cast<uint32_t*>(reinterpret_cast<uint16_t*>(p) + i) = (a >> 16) | (b & 0xffff0000u);
    } else {
        *reinterpret_cast<float2*>(reinterpret_cast<float*>(p) + i) = make_float2(x, y);
    }
}

// =====================================================================================
// Embedding + positional embedding  (transformer.py:212-227: word_emb(ids) + pos_emb * mask)
// out[b, t', :] = emb[id] + (id != 0 ? pos[t'-1] : 0) for 1 <= t' <= T ; structural rows = 0
// =====================================================================================
__global__ void embed_fwd_kernel(const int* __restrict__ ids, const float* __restrict__ emb, const float* __restrict__ pos,
                                 void* __restrict__ out, int dt, int B, int T, int C) {
    int Tp = T + 2;
    int64_t r = blockIdx.x;
    int b = (int)(r / Tp), tp = (int)(r % Tp);
    if (tp == 0 || tp == Tp - 1) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) a_st(out, r * C + c, dt, 0.f);
        return;
    }
    int id = ids[b * T + tp - 1];
    const float* e = emb + (int64_t)id * C;
    const float* p = pos + (int64_t)(tp - 1) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) a_st(out, r * C + c, dt, e[c] + (id != 0 ? p[c] : 0.f));
}
// dEmb[id] += dX[row] for id != 0 (padding_idx = 0 receives no gradient: nn.Embedding(padding_idx=0))
__global__ void embed_bwd_kernel(const int* __restrict__ ids, const void* __restrict__ dX, int dt, float* __restrict__ dEmb, int B,
                                 int T, int C) {
    int Tp = T + 2;
    int64_t r = blockIdx.x;
    int b = (int)(r / Tp), tp = (int)(r % Tp);
    if (tp == 0 || tp == Tp - 1) return;
    int id = ids[b * T + tp - 1];
    if (id == 0) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(dEmb + (int64_t)id * C + c, a_ld(dX, r * C + c, dt));
}

extern "C" int xva_fp_embed_fwd(const int32_t* ids, const float* emb, const float* pos, void* out, int dt, int B, int T, int C,
                                void* stream) {
    XVA_CHECK_ARG(ids && emb && pos && out && C % 4 == 0, "embed_fwd: bad args");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(B * (T + 2)), dim3(128), 0, (hipStream_t)stream, ids, emb, pos, out, dt, B, T, C);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_embed_bwd(const int32_t* ids, const void* dX, int dt, float* dEmb, int B, int T, int C, void* stream) {
    XVA_CHECK_ARG(ids && dX && dEmb, "embed_bwd: bad args");
    hipLaunchKernelGGL(embed_bwd_kernel, dim3(B * (T + 2)), dim3(128), 0, (hipStream_t)stream, ids, dX, dt, dEmb, B, T, C);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// =====================================================================================
// Masked softmax over keys, in place  (transformer.py:118-127: masked_fill(-inf) + softmax)
// S: (B, Tp, Ts) rows of Tp scores (Ts = padded row stride). Key j is valid iff 1 <= j <= len[b].
// One wave64 per query row.  Optional dropout on the probabilities (dropatt).
// =====================================================================================
#define SM_MAXPL 32  // up to 64*32 = 2048 keys per row
// Writes the probabilities P in place over S; with attention dropout (dropatt, transformer.py:128) ALSO writes the dropped
// copy Pd = P * m / (1 - p) that the P.V product consumes (backward needs the undropped P for the softmax Jacobian).
// pair_plane > 0 (fp32 S): Pd is written as a split-bf16 pair (hi plane at Pd, lo plane pair_plane elements after it: the operand of xva_gemm `planes`)
__device__ __forceinline__ void sm_store_pair(void* base, int64_t i, int64_t plane, float v) {
    if (plane < 0) { reinterpret_cast<_Float16*>(base)[i] = (_Float16)v; return; }      // one IEEE-half tensor instead of a pair
    uint32_t u = __float_as_uint(v); u += 0x7fffu + ((u >> 16) & 1u);
    const uint16_t h = (uint16_t)(u >> 16);
    const float r = v - __uint_as_float((uint32_t)h << 16);
    uint32_t w = __float_as_uint(r); w += 0x7fffu + ((w >> 16) & 1u);
    reinterpret_cast<uint16_t*>(base)[i] = h;
    reinterpret_cast<uint16_t*>(base)[i + plane] = (uint16_t)(w >> 16);
}
__global__ void softmax_fwd_kernel(void* __restrict__ S, void* __restrict__ Pd, int dt, const int* __restrict__ lens, int B, int Tp,
                                   int64_t Ts, float p_drop, uint64_t seed, uint32_t stream_id, int64_t pair_plane) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (row >= (int64_t)B * Tp) return;
    int b = (int)(row / Tp);
    int len = lens[b];
    const int64_t base = row * Ts;
    float v[SM_MAXPL];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < SM_MAXPL; ++i) {
        int j = lane + 64 * i;
        v[i] = (j >= 1 && j <= len) ? a_ld(S, base + j, dt) : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    m = xva_wave_max(m);
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXPL; ++i) {
        v[i] = (v[i] == -INFINITY) ? 0.f : expf(v[i] - m);
        sum += v[i];
    }
    sum = xva_wave_sum(sum);
    float inv = sum > 0.f ? 1.f / sum : 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXPL; ++i) {
        int j = lane + 64 * i;
        if (j < Ts) {
            float pv = v[i] * inv;
            a_st(S, base + j, dt, pv);
            if (Pd) {
                const float pd = pv * xva_dropout_scale(p_drop, seed, stream_id, (uint64_t)row * Tp + j);
                if (pair_plane) sm_store_pair(Pd, base + j, pair_plane, pd); else a_st(Pd, base + j, dt, pd);
            }
        }
    }
}
// dS = scale * P * (dP - sum_k dP_k P_k) in place over dP, where dP = dPd * m (attention-dropout mask regenerated).
// dS_pair != nullptr (fp32 P / dP): dS goes there as a split-bf16 pair instead of in place
__global__ void softmax_bwd_kernel(const void* __restrict__ P, void* __restrict__ dP, int dt, int B, int Tp, int64_t Ts, float scale,
                                   float p_drop, uint64_t seed, uint32_t stream_id, void* __restrict__ dS_pair, int64_t pair_plane) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (row >= (int64_t)B * Tp) return;
    const int64_t base = row * Ts;
    float pv[SM_MAXPL], dv[SM_MAXPL];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < SM_MAXPL; ++i) {
        int j = lane + 64 * i;
        pv[i] = 0.f; dv[i] = 0.f;
        if (j < Tp) {
            pv[i] = a_ld(P, base + j, dt);
            dv[i] = a_ld(dP, base + j, dt) * xva_dropout_scale(p_drop, seed, stream_id, (uint64_t)row * Tp + j);
            dot += pv[i] * dv[i];
        }
    }
    dot = xva_wave_sum(dot);
#pragma unroll
    for (int i = 0; i < SM_MAXPL; ++i) {
        int j = lane + 64 * i;
        if (j < Ts) {
            const float ds = (j < Tp) ? scale * pv[i] * (dv[i] - dot) : 0.f;
            if (dS_pair) sm_store_pair(dS_pair, base + j, pair_plane, ds); else a_st(dP, base + j, dt, ds);
        }
    }
}

extern "C" int xva_fp_softmax_fwd(void* S, void* Pd, int dt, const int32_t* lens, int B, int Tp, int64_t Ts, float p_drop, uint64_t seed,
                                  uint32_t stream_id, void* stream) {
    XVA_CHECK_ARG(S && lens && Ts >= Tp && Ts <= 64 * SM_MAXPL, "softmax_fwd: row length %ld unsupported (max %d)", (long)Ts,
                  64 * SM_MAXPL);
    XVA_CHECK_ARG(p_drop == 0.f || Pd, "softmax_fwd: attention dropout needs the dropped-copy buffer");
    int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(xva_cdiv(rows, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                       (hipStream_t)stream, S, p_drop > 0.f ? Pd : (void*)nullptr, dt, lens, B, Tp, Ts, p_drop, seed, stream_id, (int64_t)0);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_softmax_fwd_pairs(void* S, void* Pd_pair, int64_t pair_plane, const int32_t* lens, int B, int Tp, int64_t Ts, float p_drop, uint64_t seed,
                                        uint32_t stream_id, void* stream) {
    XVA_CHECK_ARG(S && Pd_pair && lens && pair_plane >= 0 && Ts >= Tp && Ts <= 64 * SM_MAXPL, "softmax_fwd_pairs: bad args");
    int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(softmax_fwd_kernel, dim3(xva_cdiv(rows, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                       (hipStream_t)stream, S, Pd_pair, XVA_F32, lens, B, Tp, Ts, p_drop, seed, stream_id, pair_plane ? pair_plane : (int64_t)-1);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_softmax_bwd(const void* P, void* dP, int dt, int B, int Tp, int64_t Ts, float scale, float p_drop,
                                  uint64_t seed, uint32_t stream_id, void* stream) {
    XVA_CHECK_ARG(P && dP && Ts >= Tp && Ts <= 64 * SM_MAXPL, "softmax_bwd: bad args");
    int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(xva_cdiv(rows, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                       (hipStream_t)stream, P, dP, dt, B, Tp, Ts, scale, p_drop, seed, stream_id, (void*)nullptr, (int64_t)0);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_softmax_bwd_pairs(const void* P, const void* dP, void* dS_pair, int64_t pair_plane, int B, int Tp, int64_t Ts, float scale, float p_drop,
                                        uint64_t seed, uint32_t stream_id, void* stream) {
    XVA_CHECK_ARG(P && dP && dS_pair && pair_plane >= 0 && Ts >= Tp && Ts <= 64 * SM_MAXPL, "softmax_bwd_pairs: bad args");
    int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3(xva_cdiv(rows, WAVES_PER_BLOCK)), dim3(64 * WAVES_PER_BLOCK), 0,
                       (hipStream_t)stream, P, const_cast<void*>(dP), XVA_F32, B, Tp, Ts, scale, p_drop, seed, stream_id, dS_pair, pair_plane ? pair_plane : (int64_t)-1);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// two adjacent values as a split-bf16 pair: hi = bf16(v) at base[i], lo = bf16(v - hi) at base[i + plane] (the operand form of xva_gemm `planes`)
// plane == 0: ONE IEEE-half tensor instead (the fp16-operand mode: same schedule, single-pass products on v_mfma_f32_16x16x32_f16)
__device__ __forceinline__ void st2_pair(uint16_t* base, int64_t i, int64_t plane, float x, float y) {
    if (plane == 0) { *reinterpret_cast<uint32_t*>(base + i) = pk_f16(x, y); return; }
    uint32_t hi, lo;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(hi) : "v"(x), "v"(y));
    const float r0 = x - __uint_as_float(hi << 16), r1 = y - __uint_as_float(hi & 0xffff0000u);
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(lo) : "v"(r0), "v"(r1));
    *reinterpret_cast<uint32_t*>(base + i) = hi;
    *reinterpret_cast<uint32_t*>(base + i + plane) = lo;
}
// =====================================================================================
// LayerNorm over channels, one wave64 per row  (transformer.py:75,146 post-LN; common/layers.py:96)
// Y = (LN(X) * gamma + beta) * rowmask [* dropout]; saves mean / rstd.  C = 64 * CPL.
// The optional dropout on the OUTPUT is ConvReLUNorm's (common/layers.py:97); transformer LayerNorms pass p_drop = 0.
// =====================================================================================
template <int CPL>
__global__ void layernorm_fwd_kernel(const void* __restrict__ X, const float* __restrict__ gamma,
                                     const float* __restrict__ beta, void* __restrict__ Y, int dt, float* __restrict__ mean,
                                     float* __restrict__ rstd, int64_t rows, int mask_mode, const int* __restrict__ lens,
                                     int Tp, float eps, float p_drop, uint64_t seed, uint32_t stream_id, uint16_t* __restrict__ Ypair, int64_t pair_plane) {
    constexpr int C = 64 * CPL;
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wave;
    if (row >= rows) return;
    // lane owns the column pairs c = 2 * lane + 128 * h + {0, 1}, h < CPL / 2 (paired loads / stores)
    float v[CPL];
    float s = 0.f;
#pragma unroll
    for (int h = 0; h < CPL / 2; ++h) { a_ld2(X, row * C + 2 * lane + 128 * h, dt, v[2 * h], v[2 * h + 1]); s += v[2 * h] + v[2 * h + 1]; }
    float mu = xva_wave_sum(s) * (1.f / C);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < CPL; ++i) { float d = v[i] - mu; q += d * d; }
    float var = xva_wave_sum(q) * (1.f / C);
    float rs = rsqrtf(var + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    bool live = xva_row_live(mask_mode, lens, Tp, row);
#pragma unroll
    for (int h = 0; h < CPL / 2; ++h) {
        const int c = 2 * lane + 128 * h;
        float y[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            y[e] = live ? (v[2 * h + e] - mu) * rs * gamma[c + e] + beta[c + e] : 0.f;
            if (p_drop > 0.f) y[e] *= xva_dropout_scale(p_drop, seed, stream_id, (uint64_t)row * C + c + e);
        }
        a_st2(Y, row * C + c, dt, y[0], y[1]);
        if (Ypair) st2_pair(Ypair, row * C + c, pair_plane, y[0], y[1]);
    }
}

// dX = rstd * (dxhat - mean(dxhat) - xhat * mean(dxhat * xhat)),  dxhat = dY' * gamma, dY' = dY * m_in (dropout applied on
// the LN output in forward); dead rows -> 0.  relu_gate: zero dX where X <= 0 (X is a post-ReLU activation).
// Second output dXm = dX * m_out: the gradient entering a dropout-ed branch (residual + dropout(branch): the residual path
// takes dX, the branch takes dXm); dXm may be null.  dgamma += sum_r dY' * xhat, dbeta += sum_r dY'.
// 16 waves per workgroup and at most ~2 workgroups per CU: each gamma / beta address receives a few hundred atomics per launch
// (one per workgroup) instead of one per 16 rows — same-address L2 atomics serialise.
#define LNB_WAVES 16
#ifndef XVA_LNB_ROWS
#define XVA_LNB_ROWS 4        // rows in flight per wave in the FAST bf16 form (3: 128 VGPRs with 3 spills, 23.2 us; 4: 113 VGPRs, none)
#endif
// BF: bf16 operands (the throughput mode's transformer LayerNorms): the two rows in flight stay PACKED in registers (3 + 3 words per lane
// instead of 6 + 6 floats), which is what lets a second prefetched row fit under 128 VGPRs without spilling
// FAST: the transformer layers' common case fixed at compile time (no dropout on the incoming gradient, no ReLU gate, no pair output, no rank-1 dY): the generic
// instantiation spilled 17 registers at its 128-VGPR budget (16 waves per workgroup); without the dead branches' live ranges the same loop keeps three rows in flight
template <int CPL, bool BF, bool FAST>
__global__ __launch_bounds__(64 * LNB_WAVES) void layernorm_bwd_kernel(const void* __restrict__ dY, const void* __restrict__ X, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const float* __restrict__ gamma, void* __restrict__ dX,
                                     void* __restrict__ dXm, int dt, float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows,
                                     int rows_per_block, int mask_mode, const int* __restrict__ lens, int Tp, int relu_gate, float p_in,
                                     uint64_t seed_in, uint32_t stream_in, float p_out, uint64_t seed_out, uint32_t stream_out,
                                     const float* __restrict__ outer_d, const float* __restrict__ outer_w, uint16_t* __restrict__ dXpair, int64_t pair_plane) {
    constexpr int C = 64 * CPL;
    __shared__ float sh_g[LNB_WAVES][C];
    __shared__ float sh_b[LNB_WAVES][C];
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float ag[CPL], ab[CPL], gm[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) { ag[i] = 0.f; ab[i] = 0.f; gm[i] = gamma[2 * lane + 128 * (i >> 1) + (i & 1)]; }   // column pairs
    // Each wave walks its rows with the NEXT row's operands already in flight: a row is a load -> two wave reductions -> store
    // chain of a few microseconds, and 7 of them back to back per wave made this kernel latency-bound (30 us for 64 MB).
    struct RowInF { float xr[CPL], gr[CPL], mu, rs; bool live; };
    struct RowInB { uint32_t xw[CPL / 2], gw[CPL / 2]; float mu, rs; bool live; };
    using RowIn = typename std::conditional<BF, RowInB, RowInF>::type;
    auto fetch = [&](int64_t row, RowIn& in) {
        in.live = row < r1 && xva_row_live(mask_mode, lens, Tp, row);
        if (!in.live) return;
        in.mu = mean[row]; in.rs = rstd[row];
        if constexpr (BF) {
#pragma unroll
            for (int h = 0; h < CPL / 2; ++h) {
                const int64_t e = row * C + 2 * lane + 128 * h;
                in.xw[h] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(X) + e);
                in.gw[h] = *reinterpret_cast<const uint32_t*>(reinterpret_cast<const uint16_t*>(dY) + e);
            }
        } else {
            const float od = outer_d ? outer_d[row] : 0.f;
#pragma unroll
            for (int h = 0; h < CPL / 2; ++h) {
                const int c = 2 * lane + 128 * h;
                a_ld2(X, row * C + c, dt, in.xr[2 * h], in.xr[2 * h + 1]);
                if (outer_d) { in.gr[2 * h] = od * outer_w[c]; in.gr[2 * h + 1] = od * outer_w[c + 1]; }   // rank-1 dY of a 1-output Linear, kept fp32
                else a_ld2(dY, row * C + c, dt, in.gr[2 * h], in.gr[2 * h + 1]);
            }
        }
    };
    auto xval = [&](const RowIn& in, int i) -> float {
        if constexpr (BF) return (i & 1) ? __uint_as_float(in.xw[i >> 1] & 0xffff0000u) : __uint_as_float(in.xw[i >> 1] << 16);
        else return in.xr[i];
    };
    auto gval = [&](const RowIn& in, int i) -> float {
        if constexpr (BF) return (i & 1) ? __uint_as_float(in.gw[i >> 1] & 0xffff0000u) : __uint_as_float(in.gw[i >> 1] << 16);
        else return in.gr[i];
    };
    // bf16: two rows in flight behind the one being processed (buffers a / b alternate; each is refilled as soon as its row is done): with
    // one, every iteration waited out a memory round trip (measured 37.3 -> 32.2 us with the dropout-masked second output, 31.7 -> 26.6 without).
    auto process = [&](const RowIn& cur, int64_t row) {
        if (!cur.live) {
#pragma unroll
            for (int h = 0; h < CPL / 2; ++h) {
                a_st2(dX, row * C + 2 * lane + 128 * h, dt, 0.f, 0.f);
                if (dXm) a_st2(dXm, row * C + 2 * lane + 128 * h, dt, 0.f, 0.f);
                if (!FAST && dXpair) st2_pair(dXpair, row * C + 2 * lane + 128 * h, pair_plane, 0.f, 0.f);
            }
            return;
        }
        const float rs = cur.rs, mu = cur.mu;
        float xh[CPL], dh[CPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) {
            const int c = 2 * lane + 128 * (i >> 1) + (i & 1);
            float g = gval(cur, i);
            if (!FAST && p_in > 0.f) g *= xva_dropout_scale(p_in, seed_in, stream_in, (uint64_t)row * C + c);
            xh[i] = (xval(cur, i) - mu) * rs;
            dh[i] = g * gm[i];
            s1 += dh[i];
            s2 += dh[i] * xh[i];
            ag[i] += g * xh[i];
            ab[i] += g;
        }
        s1 = xva_wave_sum(s1) * (1.f / C);
        s2 = xva_wave_sum(s2) * (1.f / C);
#pragma unroll
        for (int h = 0; h < CPL / 2; ++h) {
            const int c = 2 * lane + 128 * h;
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                v[e] = rs * (dh[2 * h + e] - s1 - xh[2 * h + e] * s2);
                if (!FAST && relu_gate && !(xval(cur, 2 * h + e) > 0.f)) v[e] = 0.f;
            }
            a_st2(dX, row * C + c, dt, v[0], v[1]);
            if (dXm || (!FAST && dXpair && p_out > 0.f)) {
                v[0] *= xva_dropout_scale(p_out, seed_out, stream_out, (uint64_t)row * C + c);
                v[1] *= xva_dropout_scale(p_out, seed_out, stream_out, (uint64_t)row * C + c + 1);
            }
            if (dXm) a_st2(dXm, row * C + c, dt, v[0], v[1]);
            if (!FAST && dXpair) st2_pair(dXpair, row * C + c, pair_plane, v[0], v[1]);     // the gradient entering the dropout-ed branch (= dX without dropout)
        }
    };
    if constexpr (BF && FAST) {
#if XVA_LNB_ROWS == 4
        RowIn ra, rb, rc, rd;
        fetch(r0 + wave, ra);
        fetch(r0 + wave + LNB_WAVES, rb);
        fetch(r0 + wave + 2 * LNB_WAVES, rc);
        fetch(r0 + wave + 3 * LNB_WAVES, rd);
        for (int64_t row = r0 + wave; row < r1; row += 4 * LNB_WAVES) {
            process(ra, row);
            fetch(row + 4 * LNB_WAVES, ra);
            if (row + LNB_WAVES < r1) process(rb, row + LNB_WAVES);
            fetch(row + 5 * LNB_WAVES, rb);
            if (row + 2 * LNB_WAVES < r1) process(rc, row + 2 * LNB_WAVES);
            fetch(row + 6 * LNB_WAVES, rc);
            if (row + 3 * LNB_WAVES < r1) process(rd, row + 3 * LNB_WAVES);
            fetch(row + 7 * LNB_WAVES, rd);
        }
#else
        RowIn ra, rb, rc;
        fetch(r0 + wave, ra);
        fetch(r0 + wave + LNB_WAVES, rb);
        fetch(r0 + wave + 2 * LNB_WAVES, rc);
        for (int64_t row = r0 + wave; row < r1; row += 3 * LNB_WAVES) {
            process(ra, row);
            fetch(row + 3 * LNB_WAVES, ra);
            if (row + LNB_WAVES < r1) process(rb, row + LNB_WAVES);
            fetch(row + 4 * LNB_WAVES, rb);
            if (row + 2 * LNB_WAVES < r1) process(rc, row + 2 * LNB_WAVES);
            fetch(row + 5 * LNB_WAVES, rc);
        }
#endif
    } else if constexpr (BF) {
        RowIn ra, rb;
        fetch(r0 + wave, ra);
        fetch(r0 + wave + LNB_WAVES, rb);
        for (int64_t row = r0 + wave; row < r1; row += 2 * LNB_WAVES) {
            process(ra, row);
            fetch(row + 2 * LNB_WAVES, ra);
            if (row + LNB_WAVES < r1) process(rb, row + LNB_WAVES);
            fetch(row + 3 * LNB_WAVES, rb);
        }
    } else {   // fp32 rows (parity mode, the predictors): one row ahead — two unpacked rows do not fit the 128 registers of a 16-wave workgroup
        RowIn cur, nxt;
        fetch(r0 + wave, cur);
        for (int64_t row = r0 + wave; row < r1; row += LNB_WAVES) {
            fetch(row + LNB_WAVES, nxt);
            process(cur, row);
            cur = nxt;
        }
    }
    if (dgamma) {
#pragma unroll
        for (int i = 0; i < CPL; ++i) { const int c = 2 * lane + 128 * (i >> 1) + (i & 1); sh_g[wave][c] = ag[i]; sh_b[wave][c] = ab[i]; }
        __syncthreads();
        // (publishing the workgroups' column sums and letting the last of every 8 neighbours issue the atomics — 1 / 8 of them — was built and measured: 22.8 -> 21.7 us
        // with agent-scope stores + counted waits, 114 us with a release fence, which writes the whole L2 back: not kept for 1 us)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float g = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < LNB_WAVES; ++w) { g += sh_g[w][c]; b += sh_b[w][c]; }
            atomicAdd(dgamma + c, g);
            atomicAdd(dbeta + c, b);
        }
    }
}

// ---- round 5: bf16 rows of 384 channels, FOUR rows per wave ------------------------------------------------------------------------------
// A 384-channel bf16 row is 768 bytes: with one row per wave a lane moves 3 x 4 bytes and a wave instruction 256 bytes — 27 584 rows were 27 584
// waves of {3 narrow loads -> two wave reductions -> 3 narrow stores}, 13 us for 42 MB (3.2 TB/s).  Four consecutive rows are 3 072 contiguous bytes =
// 64 lanes x 3 x 16 bytes: lane l's chunk j holds elements (512 j + 8 l) ... + 7 of the group, i.e. row (512 j + 8 l) / 384 and columns
// (512 j + 8 l) % 384 ... — the same columns for every group, so gamma / beta (and the backward's dgamma / dbeta sums) live in registers.  The four
// row sums are masked wave reductions (the same two reductions per row as before); every load / store is a full 1 KB wave instruction.
constexpr int LN4_C = 384;
__device__ __forceinline__ void ln4_unpack(const uint4& r, float (&o)[8]) {
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = __uint_as_float(w[e] << 16); o[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
}
__device__ __forceinline__ uint32_t ln4_pack2(float a, float b) {
    uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    x += 0x7fffu + ((x >> 16) & 1u); y += 0x7fffu + ((y >> 16) & 1u);
    return (x >> 16) | (y & 0xffff0000u);
}
__device__ __forceinline__ uint4 ln4_pack(const float (&v)[8]) { return make_uint4(ln4_pack2(v[0], v[1]), ln4_pack2(v[2], v[3]), ln4_pack2(v[4], v[5]), ln4_pack2(v[6], v[7])); }
// a[r] for a per-lane r in 0 .. 3 without indexing the register array dynamically (that would move it to scratch)
__device__ __forceinline__ float ln4_sel(const float (&a)[4], int r) { return r == 0 ? a[0] : (r == 1 ? a[1] : (r == 2 ? a[2] : a[3])); }
__device__ __forceinline__ bool ln4_selb(const bool (&a)[4], int r) { return r == 0 ? a[0] : (r == 1 ? a[1] : (r == 2 ? a[2] : a[3])); }
// sums of a per-(chunk, lane) value over the lanes of each of the group's four rows: out[r]
__device__ __forceinline__ void ln4_rowsums(const float (&part)[3], const int (&rw)[3], float (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) t += rw[j] == r ? part[j] : 0.f;
        out[r] = xva_wave_sum(t);
    }
}
__global__ __launch_bounds__(256) void layernorm_fwd4_kernel(const uint16_t* __restrict__ X, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             uint16_t* __restrict__ Y, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                             int mask_mode, const int* __restrict__ lens, int Tp, float eps) {
    constexpr int C = LN4_C;
    const int lane = threadIdx.x & 63;
    const int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t row0 = grp * 4;
    if (row0 >= rows) return;
    int rw[3], cl[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int e = 512 * j + 8 * lane; rw[j] = e / C; cl[j] = e - rw[j] * C; }
    float v[3][8];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        uint4 raw = make_uint4(0, 0, 0, 0);
        if (row0 + rw[j] < rows) raw = *reinterpret_cast<const uint4*>(X + row0 * C + 512 * j + 8 * lane);
        ln4_unpack(raw, v[j]);
    }
    float part[3], mu4[4], var4[4];
#pragma unroll
    for (int j = 0; j < 3; ++j) { part[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) part[j] += v[j][e]; }
    ln4_rowsums(part, rw, mu4);
#pragma unroll
    for (int r = 0; r < 4; ++r) mu4[r] *= (1.f / C);
#pragma unroll
    for (int j = 0; j < 3; ++j) { const float mu = ln4_sel(mu4, rw[j]); part[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[j][e] - mu; part[j] += d * d; } }
    ln4_rowsums(part, rw, var4);
    float rs4[4]; bool live4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { rs4[r] = rsqrtf(var4[r] * (1.f / C) + eps); live4[r] = row0 + r < rows && xva_row_live(mask_mode, lens, Tp, row0 + r); }
    if (lane < 4 && row0 + lane < rows) { mean[row0 + lane] = mu4[lane]; rstd[row0 + lane] = rs4[lane]; }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (row0 + rw[j] >= rows) continue;
        const float mu = ln4_sel(mu4, rw[j]), rs = ln4_sel(rs4, rw[j]);
        const bool live = ln4_selb(live4, rw[j]);
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + cl[j]), g1 = *reinterpret_cast<const float4*>(gamma + cl[j] + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + cl[j]), b1 = *reinterpret_cast<const float4*>(beta + cl[j] + 4);
        const float gm[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w}, bt[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) y[e] = live ? (v[j][e] - mu) * rs * gm[e] + bt[e] : 0.f;
        *reinterpret_cast<uint4*>(Y + row0 * C + 512 * j + 8 * lane) = ln4_pack(y);
    }
}

// The BACKWARD stays on the one-row-per-wave kernel below (16 waves per workgroup, two packed rows in flight per wave: 28 us for the decoder rows).  Two
// wider forms were built and measured in round 5 and are not kept: four rows per wave like the forward (24 + 24 dgamma / dbeta column sums per lane next to
// the loads in flight: 250 VGPRs, two waves per SIMD — 58 us) and two rows per wave, one per half-wave with 12 fixed columns per lane and 8-byte accesses
// (168 VGPRs, 12 waves per workgroup — 44 us).  The kernel is bound by rows in flight per CU, and the per-lane column sums are what limits those.
// 1 (default): bf16 rows of 384 channels take the four-rows-per-wave forward kernel above; 0: one row per wave.  env XVA_FP_LN4 (A/B, tests)
static int g_ln4 = [] { const char* e = getenv("XVA_FP_LN4"); return e ? atoi(e) : 1; }();
extern "C" int xva_fp_set_ln4(int mode) { int old = g_ln4; g_ln4 = mode; return old; }

static int layernorm_fwd_impl(const void* X, const float* gamma, const float* beta, void* Y, int dt, float* mean, float* rstd,
                                    int64_t rows, int C, int mask_mode, const int32_t* lens, int Tp, float p_drop, uint64_t seed,
                                    uint32_t stream_id, void* y_pair, int64_t pair_plane, void* stream) {
    XVA_CHECK_ARG(X && gamma && beta && Y && mean && rstd, "layernorm_fwd: null");
    XVA_CHECK_ARG(C == 384 || C == 256, "layernorm: C must be 384 or 256 (got %d)", C);
    XVA_CHECK_ARG(!y_pair || (dt == XVA_F32 && pair_plane >= 0 && pair_plane % 2 == 0 && ((uintptr_t)y_pair % 4) == 0), "layernorm_fwd: pair output wants fp32 rows");
    uint16_t* yp = reinterpret_cast<uint16_t*>(y_pair);
    if (g_ln4 && !y_pair && C == 384 && dt == XVA_BF16 && p_drop == 0.f && ((uintptr_t)X % 16) == 0 && ((uintptr_t)Y % 16) == 0 && ((uintptr_t)gamma % 16) == 0 &&
        ((uintptr_t)beta % 16) == 0) {
        hipLaunchKernelGGL(layernorm_fwd4_kernel, dim3((unsigned)xva_cdiv(xva_cdiv(rows, 4), 4)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint16_t*>(X),
                           gamma, beta, reinterpret_cast<uint16_t*>(Y), mean, rstd, rows, mask_mode, lens, Tp, 1e-5f);
        XVA_LAUNCH_CHECK();
        return XVA_OK;
    }
    dim3 grid(xva_cdiv(rows, WAVES_PER_BLOCK)), block(64 * WAVES_PER_BLOCK);
    if (C == 384)
        hipLaunchKernelGGL((layernorm_fwd_kernel<6>), grid, block, 0, (hipStream_t)stream, X, gamma, beta, Y, dt, mean, rstd, rows,
                           mask_mode, lens, Tp, 1e-5f, p_drop, seed, stream_id, yp, pair_plane);
    else
        hipLaunchKernelGGL((layernorm_fwd_kernel<4>), grid, block, 0, (hipStream_t)stream, X, gamma, beta, Y, dt, mean, rstd, rows,
                           mask_mode, lens, Tp, 1e-5f, p_drop, seed, stream_id, yp, pair_plane);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_layernorm_fwd(const void* X, const float* gamma, const float* beta, void* Y, int dt, float* mean, float* rstd,
                                    int64_t rows, int C, int mask_mode, const int32_t* lens, int Tp, float p_drop, uint64_t seed,
                                    uint32_t stream_id, void* stream) {
    return layernorm_fwd_impl(X, gamma, beta, Y, dt, mean, rstd, rows, C, mask_mode, lens, Tp, p_drop, seed, stream_id, nullptr, 0, stream);
}
// fp32 rows; ALSO writes Y as a split-bf16 pair (hi plane at y_pair, lo plane pair_plane elements after it) — the next product's operand without a split launch
extern "C" int xva_fp_layernorm_fwd_pair(const void* X, const float* gamma, const float* beta, void* Y, void* y_pair, int64_t pair_plane, float* mean, float* rstd,
                                         int64_t rows, int C, int mask_mode, const int32_t* lens, int Tp, void* stream) {
    XVA_CHECK_ARG(y_pair, "layernorm_fwd_pair: null");
    return layernorm_fwd_impl(X, gamma, beta, Y, XVA_F32, mean, rstd, rows, C, mask_mode, lens, Tp, 0.f, 0, 0, y_pair, pair_plane, stream);
}
static int layernorm_bwd_impl(const void* dY, const void* X, const float* mean, const float* rstd, const float* gamma,
                                    void* dX, void* dXm, int dt, float* dgamma, float* dbeta, int64_t rows, int C, int mask_mode,
                                    const int32_t* lens, int Tp, int relu_gate, float p_in, uint64_t seed_in, uint32_t stream_in,
                                    float p_out, uint64_t seed_out, uint32_t stream_out, const float* outer_d, const float* outer_w,
                                    void* dx_pair, int64_t pair_plane, void* stream) {
    XVA_CHECK_ARG((dY || (outer_d && outer_w)) && X && mean && rstd && gamma && dX, "layernorm_bwd: null");
    XVA_CHECK_ARG(!dx_pair || (dt == XVA_F32 && pair_plane >= 0 && pair_plane % 2 == 0 && ((uintptr_t)dx_pair % 4) == 0), "layernorm_bwd: pair output wants fp32 rows");
    uint16_t* dxp = reinterpret_cast<uint16_t*>(dx_pair);
    XVA_CHECK_ARG(C == 384 || C == 256, "layernorm: C must be 384 or 256 (got %d)", C);
    XVA_CHECK_ARG((dgamma == nullptr) == (dbeta == nullptr), "layernorm_bwd: dgamma/dbeta must both be given or both null");
    int rpb = (int)xva_cdiv(rows, 256);          // <= 256 workgroups (one per CU)
    rpb = (rpb + LNB_WAVES - 1) / LNB_WAVES * LNB_WAVES;
    dim3 grid(xva_cdiv(rows, rpb)), block(64 * LNB_WAVES);
    const bool bf = dt == XVA_BF16 && dY != nullptr && !outer_d;
#define XVA_LNB(CPL, BFV) hipLaunchKernelGGL((layernorm_bwd_kernel<CPL, BFV, false>), grid, block, 0, (hipStream_t)stream, dY, X, mean, rstd, gamma, dX, dXm, dt, dgamma, \
                                             dbeta, rows, rpb, mask_mode, lens, Tp, relu_gate, p_in, seed_in, stream_in, p_out, seed_out, stream_out, outer_d, outer_w, \
                                             dxp, pair_plane)
    static const int lnb_fast = 1;
    if (lnb_fast && C == 384 && bf && p_in <= 0.f && !relu_gate && !dxp) {
        hipLaunchKernelGGL((layernorm_bwd_kernel<6, true, true>), grid, block, 0, (hipStream_t)stream, dY, X, mean, rstd, gamma, dX, dXm, dt, dgamma, dbeta, rows, rpb, mask_mode, lens,
                           Tp, 0, 0.f, (uint64_t)0, 0u, p_out, seed_out, stream_out, (const float*)nullptr, (const float*)nullptr, (uint16_t*)nullptr, (int64_t)0);
    } else if (C == 384) { if (bf) XVA_LNB(6, true); else XVA_LNB(6, false); }
    else { if (bf) XVA_LNB(4, true); else XVA_LNB(4, false); }
#undef XVA_LNB
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_layernorm_bwd(const void* dY, const void* X, const float* mean, const float* rstd, const float* gamma,
                                    void* dX, void* dXm, int dt, float* dgamma, float* dbeta, int64_t rows, int C, int mask_mode,
                                    const int32_t* lens, int Tp, int relu_gate, float p_in, uint64_t seed_in, uint32_t stream_in,
                                    float p_out, uint64_t seed_out, uint32_t stream_out, const float* outer_d, const float* outer_w,
                                    void* stream) {
    return layernorm_bwd_impl(dY, X, mean, rstd, gamma, dX, dXm, dt, dgamma, dbeta, rows, C, mask_mode, lens, Tp, relu_gate, p_in, seed_in, stream_in, p_out,
                              seed_out, stream_out, outer_d, outer_w, nullptr, 0, stream);
}
// fp32 rows; ALSO writes the gradient that enters the dropout-ed branch (dX * m_out; dX itself when p_out == 0) as a split-bf16 pair
extern "C" int xva_fp_layernorm_bwd_pair(const void* dY, const void* X, const float* mean, const float* rstd, const float* gamma, void* dX, void* dXm, void* dx_pair,
                                         int64_t pair_plane, float* dgamma, float* dbeta, int64_t rows, int C, int mask_mode, const int32_t* lens, int Tp, float p_out,
                                         uint64_t seed_out, uint32_t stream_out, void* stream) {
    XVA_CHECK_ARG(dx_pair, "layernorm_bwd_pair: null");
    return layernorm_bwd_impl(dY, X, mean, rstd, gamma, dX, dXm, XVA_F32, dgamma, dbeta, rows, C, mask_mode, lens, Tp, 0, 0.f, 0, 0, p_out, seed_out, stream_out,
                              nullptr, nullptr, dx_pair, pair_plane, stream);
}

// =====================================================================================
// Column sum (bias gradients): out[c] += sum_r X[r][c].  Block = 64 columns x 4 row lanes.
// =====================================================================================
__global__ void colsum_kernel(const void* __restrict__ X, int dt, float* __restrict__ out, int64_t rows, int C, int64_t ld,
                              int rows_per_block) {
    __shared__ float sh[4][64];
    int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    int c = blockIdx.x * 64 + cl;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float acc = 0.f;
    if (c < C)
        for (int64_t r = r0 + rl; r < r1; r += 4) acc += a_ld(X, r * ld + c, dt);
    sh[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C) atomicAdd(out + c, sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]);
}
// paired-column version (C and ld even): a wave reads 128 adjacent columns of a row per load
__global__ void colsum2_kernel(const void* __restrict__ X, int dt, float* __restrict__ out, int64_t rows, int C, int64_t ld,
                               int rows_per_block) {
    __shared__ float sh[4][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 128 + 2 * lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    if ((dt == XVA_BF16 || dt == XVA_F16) && C % 8 == 0 && ld % 8 == 0 && ((uintptr_t)X % 16) == 0) {
        // 16-byte loads: a lane owns 8 columns, 16 lanes the 128-column tile, a wave instruction 4 rows, 8 instructions in flight
        const int c8 = blockIdx.x * 128 + (lane & 15) * 8;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c8 < C) {
            const uint16_t* xp = reinterpret_cast<const uint16_t*>(X) + c8;
            auto add = [&](const uint4& q) {
                const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (dt == XVA_F16) { const xva_h2 h = __builtin_bit_cast(xva_h2, u[e]); a[2 * e] += (float)h[0]; a[2 * e + 1] += (float)h[1]; }
                    else { a[2 * e] += __uint_as_float(u[e] << 16); a[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u); }
                }
            };
            int64_t r = r0 + w * 4 + (lane >> 4);
            for (; r + 112 < r1; r += 128) {
                uint4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const uint4*>(xp + (r + 16 * u) * ld);
#pragma unroll
                for (int u = 0; u < 8; ++u) add(q[u]);
            }
            for (; r < r1; r += 16) add(*reinterpret_cast<const uint4*>(xp + r * ld));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] += __shfl_xor(a[e], 16); a[e] += __shfl_xor(a[e], 32); }
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[w][lane * 8 + e] = a[e];
        }
    } else {
        float a0 = 0.f, a1 = 0.f;
        if (c < C) {
            int64_t r = r0 + w;
            for (; r + 4 < r1; r += 8) {       // two rows in flight
                float x0, y0, x1, y1;
                a_ld2(X, r * ld + c, dt, x0, y0);
                a_ld2(X, (r + 4) * ld + c, dt, x1, y1);
                a0 += x0 + x1; a1 += y0 + y1;
            }
            for (; r < r1; r += 4) { float x, y; a_ld2(X, r * ld + c, dt, x, y); a0 += x; a1 += y; }
        }
        sh[w][2 * lane] = a0; sh[w][2 * lane + 1] = a1;
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int cc = blockIdx.x * 128 + threadIdx.x;
        if (cc < C) atomicAdd(out + cc, sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]);
    }
}
extern "C" int xva_fp_colsum(const void* X, int dt, float* out, int64_t rows, int C, int64_t ld, void* stream) {
    XVA_CHECK_ARG(X && out && C > 0, "colsum: bad args");
    if (rows <= 0) return XVA_OK;
    if (C % 2 == 0 && ld % 2 == 0 && ((uintptr_t)X % 8) == 0) {
        const int rpb2 = 128;
        hipLaunchKernelGGL(colsum2_kernel, dim3(xva_cdiv(C, 128), xva_cdiv(rows, rpb2)), dim3(256), 0, (hipStream_t)stream, X, dt, out,
                           rows, C, ld, rpb2);
        XVA_LAUNCH_CHECK();
        return XVA_OK;
    }
    const int rpb = 256;
    hipLaunchKernelGGL(colsum_kernel, dim3(xva_cdiv(C, 64), xva_cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream, X, dt, out, rows,
                       C, ld, rpb);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// =====================================================================================
// average_pitch (model.py:82-100) and the frame->token map of regulate_len (model.py:59-79).
// One block per batch item; thread j owns token j.
//  avg_out: (B, Tt+2) padded 1-channel sequence: mean of the NON-ZERO frames of token j's span (0 if none),
//           optionally log(1 + .)   (energy: model.py:413-414)
// =====================================================================================
__global__ void avg_pitch_kernel(const float* __restrict__ dense, const int* __restrict__ durs, float* __restrict__ avg_out,
                                 int Tt, int Tm, int log1p_) {
    int b = blockIdx.x;
    const int* d = durs + b * Tt;
    const float* x = dense + (int64_t)b * Tm;
    float* o = avg_out + (int64_t)b * (Tt + 2);
    if (threadIdx.x == 0) { o[0] = 0.f; o[Tt + 1] = 0.f; }
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) {
        int start = 0;
        for (int i = 0; i < j; ++i) start += d[i];
        int end = start + d[j];
        if (start > Tm) start = Tm;   // torch.gather would raise; synthetic/real batches never exceed Tm
        if (end > Tm) end = Tm;
        float s = 0.f; int n = 0;
        for (int t = start; t < end; ++t) { float v = x[t]; if (v != 0.f) { s += v; ++n; } }
        float a = n > 0 ? s / (float)n : 0.f;
        if (log1p_) a = logf(1.0f + a);
        o[j + 1] = a;
    }
}
// tok[b][t] = token owning frame t (or -1), tstart[b][j] = first frame of token j, dec_lens[b] = min(sum reps, Tm)
__global__ void lenreg_map_kernel(const int* __restrict__ durs, int* __restrict__ tok, int* __restrict__ tstart,
                                  int* __restrict__ dec_lens, int Tt, int Tm, float pace) {
    int b = blockIdx.x;
    const int* d = durs + b * Tt;
    int* tk = tok + (int64_t)b * Tm;
    for (int t = threadIdx.x; t < Tm; t += blockDim.x) tk[t] = -1;
    __syncthreads();
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) {
        int start = 0;
        for (int i = 0; i < j; ++i) start += (int)((float)d[i] * pace + 0.5f);
        int rep = (int)((float)d[j] * pace + 0.5f);
        tstart[b * (Tt + 1) + j] = start;
        for (int t = start; t < start + rep && t < Tm; ++t) tk[t] = j;
        if (j == Tt - 1) {
            int tot = start + rep;
            tstart[b * (Tt + 1) + Tt] = tot;
            dec_lens[b] = tot < Tm ? tot : Tm;
        }
    }
}
extern "C" int xva_fp_avg_pitch(const float* dense, const int32_t* durs, float* avg_out, int B, int Tt, int Tm, int log1p_,
                                void* stream) {
    XVA_CHECK_ARG(dense && durs && avg_out, "avg_pitch: null");
    hipLaunchKernelGGL(avg_pitch_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dense, durs, avg_out, Tt, Tm, log1p_);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_lenreg_map(const int32_t* durs, int32_t* tok, int32_t* tstart, int32_t* dec_lens, int B, int Tt, int Tm,
                                 float pace, void* stream) {
    XVA_CHECK_ARG(durs && tok && tstart && dec_lens, "lenreg_map: null");
    hipLaunchKernelGGL(lenreg_map_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, durs, tok, tstart, dec_lens, Tt, Tm, pace);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// =====================================================================================
// Conditioning add: out = (in + Conv1d(1 -> C, k=3, pad=1)(s) + bias) * lenmask    (model.py:234-237,403,418)
// s: (B, Tp) padded 1-channel sequence; w: (C, 3) ; rows live iff 1 <= t' <= lens[b].
// =====================================================================================
__global__ void cond_add_fwd_kernel(const void* __restrict__ in, const float* __restrict__ s, const float* __restrict__ w,
                                    const float* __restrict__ bias, void* __restrict__ out, int dt, const int* __restrict__ lens, int Tp,
                                    int C) {
    int64_t r = blockIdx.x;
    int b = (int)(r / Tp), tp = (int)(r % Tp);
    bool live = tp >= 1 && tp <= lens[b];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (live) { const float* sp = s + (int64_t)b * Tp + tp; s0 = sp[-1]; s1 = sp[0]; s2 = sp[1]; }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = 0.f;
        if (live) v = a_ld(in, r * C + c, dt) + w[c * 3 + 0] * s0 + w[c * 3 + 1] * s1 + w[c * 3 + 2] * s2 + bias[c];
        a_st(out, r * C + c, dt, v);
    }
}
// dw[c][k] += sum_r dOut[r][c] * s[b][t'-1+k], db[c] += sum_r dOut[r][c]   (dOut is zero on dead rows)
__global__ void cond_add_bwd_kernel(const void* __restrict__ dOut, int dt, const float* __restrict__ s, float* __restrict__ dw,
                                    float* __restrict__ db, const int* __restrict__ lens, int64_t rows, int Tp, int C,
                                    int rows_per_block) {
    __shared__ float sh[4][4][64];
    int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    int c = blockIdx.x * 64 + cl;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, ab = 0.f;
    if (c < C) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            int b = (int)(r / Tp), tp = (int)(r % Tp);
            if (tp < 1 || tp > lens[b]) continue;
            float g = a_ld(dOut, r * C + c, dt);
            const float* sp = s + (int64_t)b * Tp + tp;
            a0 += g * sp[-1]; a1 += g * sp[0]; a2 += g * sp[1]; ab += g;
        }
    }
    sh[0][rl][cl] = a0; sh[1][rl][cl] = a1; sh[2][rl][cl] = a2; sh[3][rl][cl] = ab;
    __syncthreads();
    if (rl == 0 && c < C) {
        float t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = sh[k][0][cl] + sh[k][1][cl] + sh[k][2][cl] + sh[k][3][cl];
        if (dw) { atomicAdd(dw + c * 3 + 0, t[0]); atomicAdd(dw + c * 3 + 1, t[1]); atomicAdd(dw + c * 3 + 2, t[2]); }
        if (db) atomicAdd(db + c, t[3]);
    }
}
extern "C" int xva_fp_cond_add_fwd(const void* in, const float* s, const float* w, const float* bias, void* out, int dt,
                                   const int32_t* lens, int B, int Tp, int C, void* stream) {
    XVA_CHECK_ARG(in && s && w && bias && out && lens, "cond_add_fwd: null");
    hipLaunchKernelGGL(cond_add_fwd_kernel, dim3(B * Tp), dim3(128), 0, (hipStream_t)stream, in, s, w, bias, out, dt, lens, Tp, C);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_cond_add_bwd(const void* dOut, int dt, const float* s, float* dw, float* db, const int32_t* lens, int B, int Tp,
                                   int C, void* stream) {
    XVA_CHECK_ARG(dOut && s && lens, "cond_add_bwd: null");
    int64_t rows = (int64_t)B * Tp;
    const int rpb = 32;   // 8 dependent row iterations per thread (was 32: 25 us for 3.7 MB, latency-bound); ~150 atomics per address
    hipLaunchKernelGGL(cond_add_bwd_kernel, dim3(xva_cdiv(C, 64), xva_cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream, dOut, dt, s,
                       dw, db, lens, rows, Tp, C, rpb);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// =====================================================================================
// regulate_len (model.py:59-79) as a gather + the decoder's positional embedding (transformer.py:224-227).
//   out[b, t', :] = enc[b, tok[b][t'-1] + 1, :] + pos[t'-1]   for 1 <= t' <= dec_lens[b], else 0
// backward = segmented sum (no atomics): dEnc[b, j+1, :] (+)= sum_{t in span j} dOut[b, t+1, :]
// =====================================================================================
__global__ void lenreg_fwd_kernel(const void* __restrict__ enc, const int* __restrict__ tok, const int* __restrict__ dec_lens,
                                  const float* __restrict__ pos, void* __restrict__ out, int dt, int Tt, int Tm, int C) {
    int Tmp = Tm + 2, Ttp = Tt + 2;
    int64_t r = blockIdx.x;
    int b = (int)(r / Tmp), tp = (int)(r % Tmp);
    int j = -1;
    if (tp >= 1 && tp <= dec_lens[b]) j = tok[(int64_t)b * Tm + tp - 1];
    if (j < 0) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) a_st(out, r * C + c, dt, 0.f);
        return;
    }
    const int64_t eb = ((int64_t)b * Ttp + j + 1) * C;
    const float* p = pos + (int64_t)(tp - 1) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) a_st(out, r * C + c, dt, a_ld(enc, eb + c, dt) + p[c]);
}
__global__ void lenreg_bwd_kernel(const void* __restrict__ dOut, const int* __restrict__ tstart, const int* __restrict__ dec_lens,
                                  void* __restrict__ dEnc, int dt, int Tt, int Tm, int C, int accumulate) {
    int Tmp = Tm + 2, Ttp = Tt + 2;
    int64_t r = blockIdx.x;  // row of dEnc
    int b = (int)(r / Ttp), tp = (int)(r % Ttp);
    int t0 = 0, t1 = 0;
    if (tp >= 1 && tp <= Tt) {
        t0 = tstart[b * (Tt + 1) + tp - 1];
        t1 = tstart[b * (Tt + 1) + tp];
        int dl = dec_lens[b];
        if (t0 > dl) t0 = dl;
        if (t1 > dl) t1 = dl;
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f;
        for (int t = t0; t < t1; ++t) a += a_ld(dOut, ((int64_t)b * Tmp + t + 1) * C + c, dt);
        if (accumulate) a += a_ld(dEnc, r * C + c, dt);
        a_st(dEnc, r * C + c, dt, a);
    }
}
extern "C" int xva_fp_lenreg_fwd(const void* enc, const int32_t* tok, const int32_t* dec_lens, const float* pos, void* out, int dt,
                                 int B, int Tt, int Tm, int C, void* stream) {
    XVA_CHECK_ARG(enc && tok && dec_lens && pos && out && C % 4 == 0, "lenreg_fwd: bad args");
    hipLaunchKernelGGL(lenreg_fwd_kernel, dim3(B * (Tm + 2)), dim3(128), 0, (hipStream_t)stream, enc, tok, dec_lens, pos, out, dt, Tt,
                       Tm, C);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_lenreg_bwd(const void* dOut, const int32_t* tstart, const int32_t* dec_lens, void* dEnc, int dt, int B, int Tt,
                                 int Tm, int C, int accumulate, void* stream) {
    XVA_CHECK_ARG(dOut && tstart && dec_lens && dEnc && C % 4 == 0, "lenreg_bwd: bad args");
    hipLaunchKernelGGL(lenreg_bwd_kernel, dim3(B * (Tt + 2)), dim3(128), 0, (hipStream_t)stream, dOut, tstart, dec_lens, dEnc, dt, Tt,
                       Tm, C, accumulate);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// =====================================================================================
// FastPitchLoss (loss_function.py:63-154).  Two phases so that data-parallel ranks can all-reduce the
// numerators / denominators in between (global normalisation = the reference's DataParallel semantics):
//   partials:  acc[0..7] += {mel_num, mel_den, pitch_num, tok_den, energy_num, dur_num, 0, 0}
//   grads:     d_pred = grad_scale * w * 2 * (pred - tgt) * mask / den
// mel_out: (B, Tm+2, 80) padded token-major ; mel_tgt: (B, 80, Tm) reference layout ; mask = (tgt != 0).
// =====================================================================================
// One workgroup = one item x 64 frames x all NM mel channels: the target (B, NM, Tm) is read along t, the prediction (B, Tm + 2, NM) along c,
// through a [64][NM + 1] LDS tile (read in the target's order, the prediction was fetched 2 bytes per 160-byte row: 65 us for 13 MB).
__global__ __launch_bounds__(256) void mel_loss_partial_kernel(const void* __restrict__ mel_out, int dt, const float* __restrict__ mel_tgt,
                                                               float* __restrict__ acc, int B, int Tm, int NM) {
    __shared__ float sh[16];
    __shared__ float tile[64][81];
    const int tb = blockIdx.x * 64, b = blockIdx.y;
    const int nt = min(64, Tm - tb);
    for (int i = threadIdx.x; i < nt * NM; i += 256) {                 // prediction: consecutive threads = consecutive channels of a frame
        const int t = i / NM, c = i - t * NM;
        tile[t][c] = a_ld(mel_out, ((int64_t)b * (Tm + 2) + tb + t + 1) * NM + c, dt);
    }
    __syncthreads();
    float num = 0.f, den = 0.f;
    for (int i = threadIdx.x; i < NM * 64; i += 256) {                 // target: consecutive threads = consecutive frames of a channel
        const int c = i >> 6, t = i & 63;
        if (t < nt) {
            const float tg = mel_tgt[((int64_t)b * NM + c) * Tm + tb + t];
            if (tg != 0.f) { const float d = tile[t][c] - tg; num += d * d; den += 1.f; }
        }
    }
    num = xva_block_sum(num, sh);
    den = xva_block_sum(den, sh);
    if (threadIdx.x == 0 && den > 0.f) { atomicAdd(acc + 0, num); atomicAdd(acc + 1, den); }
}
__global__ void mel_loss_grad_kernel(const void* __restrict__ mel_out, int dt, const float* __restrict__ mel_tgt,
                                     const float* __restrict__ acc, void* __restrict__ d_mel, int B, int Tm, int NM,
                                     float grad_scale) {
    int Tmp = Tm + 2;
    int64_t total = (int64_t)B * Tmp * NM;
    float den = acc[1];
    float k = den > 0.f ? 2.f * grad_scale / den : 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int c = (int)(i % NM);
        int tp = (int)((i / NM) % Tmp);
        int b = (int)(i / ((int64_t)NM * Tmp));
        float g = 0.f;
        if (tp >= 1 && tp <= Tm) {
            float tg = mel_tgt[((int64_t)b * NM + c) * Tm + tp - 1];
            if (tg != 0.f) g = k * (a_ld(mel_out, i, dt) - tg);
        }
        a_st(d_mel, i, dt, g);
    }
}
// token-level masked MSE between pred (B, Tp) padded sequence [ld 1] and tgt (B, Tp) padded sequence.
// mode 0: tgt as is ; mode 1: tgt = log(dur + 1) from integer durations (B, Tt) (stage 2, loss_function.py:87-90)
__global__ void tok_loss_partial_kernel(const float* __restrict__ pred, const float* __restrict__ tgt,
                                        const int* __restrict__ durs, const int* __restrict__ lens, float* __restrict__ acc,
                                        int slot_num, int slot_den, int B, int Tt, int mode) {
    __shared__ float sh[16];
    int Tp = Tt + 2;
    float num = 0.f, den = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * Tp; i += gridDim.x * blockDim.x) {
        int b = i / Tp, tp = i % Tp;
        if (tp >= 1 && tp <= lens[b]) {
            float tg = mode == 1 ? logf((float)durs[b * Tt + tp - 1] + 1.0f) : tgt[i];
            float d = pred[i] - tg;
            num += d * d;
            den += 1.f;
        }
    }
    num = xva_block_sum(num, sh);
    den = xva_block_sum(den, sh);
    if (threadIdx.x == 0) { atomicAdd(acc + slot_num, num); if (slot_den >= 0) atomicAdd(acc + slot_den, den); }
}
__global__ void tok_loss_grad_kernel(const float* __restrict__ pred, const float* __restrict__ tgt, const int* __restrict__ durs,
                                     const int* __restrict__ lens, const float* __restrict__ acc, int slot_den,
                                     float* __restrict__ d_pred, int B, int Tt, int mode, float weight) {
    int Tp = Tt + 2;
    float den = acc[slot_den];
    float k = den > 0.f ? 2.f * weight / den : 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * Tp; i += gridDim.x * blockDim.x) {
        int b = i / Tp, tp = i % Tp;
        float g = 0.f;
        if (tp >= 1 && tp <= lens[b]) {
            float tg = mode == 1 ? logf((float)durs[b * Tt + tp - 1] + 1.0f) : tgt[i];
            g = k * (pred[i] - tg);
        }
        d_pred[i] = g;
    }
}
// out[0] = total, out[1] = mel, out[2] = dur, out[3] = pitch, out[4] = energy  (unscaled component means)
__global__ void loss_finalize_kernel(const float* __restrict__ acc, float* __restrict__ out, int stage, float dur_w, float pitch_w,
                                     float energy_w) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float mel = (stage >= 3 && acc[1] > 0.f) ? acc[0] / acc[1] : 0.f;
    float tden = acc[3];
    float pitch = (stage == 3 && tden > 0.f) ? acc[2] / tden : 0.f;
    float energy = (stage == 3 && tden > 0.f) ? acc[4] / tden : 0.f;
    float dur = (stage == 2 && tden > 0.f) ? acc[5] / tden : 0.f;
    out[0] = mel + dur_w * dur + pitch_w * pitch + energy_w * energy;
    out[1] = mel; out[2] = dur; out[3] = pitch; out[4] = energy;
}

extern "C" int xva_fp_loss_partials(int stage, int dt, const void* mel_out, const float* mel_tgt, const float* pitch_pred,
                                    const float* pitch_tgt, const float* energy_pred, const float* energy_tgt,
                                    const float* log_dur_pred, const int32_t* durs, const int32_t* in_lens, float* acc, int B,
                                    int Tt, int Tm, void* stream) {
    XVA_CHECK_ARG(acc && in_lens, "loss_partials: null");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(acc, 0, 8 * sizeof(float), st) != hipSuccess) { xva_set_error("loss_partials: memset failed"); return XVA_ERR_HIP; }
    int gtok = xva_cdiv((int64_t)B * (Tt + 2), 256);
    if (stage == 3 || stage == 4) {
        XVA_CHECK_ARG(mel_out && mel_tgt, "loss_partials: null mel");
        hipLaunchKernelGGL(mel_loss_partial_kernel, dim3(xva_cdiv(Tm, 64), B), dim3(256), 0, st, mel_out, dt, mel_tgt, acc, B, Tm, 80);
    }
    if (stage == 3) {
        XVA_CHECK_ARG(pitch_pred && pitch_tgt && energy_pred && energy_tgt, "loss_partials: null pitch/energy");
        hipLaunchKernelGGL(tok_loss_partial_kernel, dim3(gtok), dim3(256), 0, st, pitch_pred, pitch_tgt, (const int*)nullptr, in_lens,
                           acc, 2, 3, B, Tt, 0);
        hipLaunchKernelGGL(tok_loss_partial_kernel, dim3(gtok), dim3(256), 0, st, energy_pred, energy_tgt, (const int*)nullptr,
                           in_lens, acc, 4, -1, B, Tt, 0);
    }
    if (stage == 2) {
        XVA_CHECK_ARG(log_dur_pred && durs, "loss_partials: null durations");
        hipLaunchKernelGGL(tok_loss_partial_kernel, dim3(gtok), dim3(256), 0, st, log_dur_pred, (const float*)nullptr, durs, in_lens,
                           acc, 5, 3, B, Tt, 1);
    }
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// The two DENOMINATORS of the masked means from the targets alone — den[0] = #(mel_tgt != 0), den[1] = sum of the token lengths — so that
// data-parallel ranks can all-reduce them while the forward pass runs: the gradients need only the global denominators (the numerators are
// reporting), which takes the 8-float exchange of xva_fp_loss_partials off the forward -> backward critical path.
__global__ __launch_bounds__(256) void loss_den_kernel(const float* __restrict__ mel_tgt, const int* __restrict__ lens, float* __restrict__ den, int64_t nmel,
                                                       int B, int Tt) {
    __shared__ float sh[16];
    float c = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nmel; i += (int64_t)gridDim.x * blockDim.x) c += mel_tgt[i] != 0.f ? 1.f : 0.f;
    c = xva_block_sum(c, sh);
    if (threadIdx.x == 0 && c > 0.f) atomicAdd(den + 0, c);
    if (blockIdx.x == 0) {
        float t = 0.f;
        for (int b = threadIdx.x; b < B; b += blockDim.x) t += (float)min(max(lens[b], 0), Tt + 1);
        t = xva_block_sum(t, sh);
        if (threadIdx.x == 0) atomicAdd(den + 1, t);
    }
}
extern "C" int xva_fp_loss_denominators(int stage, const float* mel_tgt, const int32_t* in_lens, float* den2, int B, int Tt, int Tm, void* stream) {
    XVA_CHECK_ARG(den2 && in_lens && (mel_tgt || stage == 2), "loss_denominators: null");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(den2, 0, 2 * sizeof(float), st) != hipSuccess) { xva_set_error("loss_denominators: memset failed"); return XVA_ERR_HIP; }
    const int64_t nmel = (stage == 3 || stage == 4) ? (int64_t)B * 80 * Tm : 0;
    int grid = (int)((nmel + 256 * 16 - 1) / (256 * 16)); if (grid < 1) grid = 1; if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(loss_den_kernel, dim3(grid), dim3(256), 0, st, mel_tgt, in_lens, den2, nmel, B, Tt);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

extern "C" int xva_fp_loss_grads(int stage, int dt, const void* mel_out, const float* mel_tgt, const float* pitch_pred,
                                 const float* pitch_tgt, const float* energy_pred, const float* energy_tgt,
                                 const float* log_dur_pred, const int32_t* durs, const int32_t* in_lens, const float* acc,
                                 float* losses_out, void* d_mel, float* d_pitch, float* d_energy, float* d_logdur, int B, int Tt,
                                 int Tm, float grad_scale, float dur_w, float pitch_w, float energy_w, void* stream) {
    XVA_CHECK_ARG(acc && losses_out && in_lens, "loss_grads: null");
    hipStream_t st = (hipStream_t)stream;
    int gtok = xva_cdiv((int64_t)B * (Tt + 2), 256);
    if (stage == 3 || stage == 4) {
        XVA_CHECK_ARG(mel_out && mel_tgt && d_mel, "loss_grads: null mel");
        int64_t total = (int64_t)B * (Tm + 2) * 80;
        int grid = (int)((total + 255) / 256); if (grid > 4096) grid = 4096;
        hipLaunchKernelGGL(mel_loss_grad_kernel, dim3(grid), dim3(256), 0, st, mel_out, dt, mel_tgt, acc, d_mel, B, Tm, 80, grad_scale);
    }
    if (stage == 3) {
        XVA_CHECK_ARG(d_pitch && d_energy, "loss_grads: null pitch/energy grads");
        hipLaunchKernelGGL(tok_loss_grad_kernel, dim3(gtok), dim3(256), 0, st, pitch_pred, pitch_tgt, (const int*)nullptr, in_lens, acc,
                           3, d_pitch, B, Tt, 0, grad_scale * pitch_w);
        hipLaunchKernelGGL(tok_loss_grad_kernel, dim3(gtok), dim3(256), 0, st, energy_pred, energy_tgt, (const int*)nullptr, in_lens,
                           acc, 3, d_energy, B, Tt, 0, grad_scale * energy_w);
    }
    if (stage == 2) {
        XVA_CHECK_ARG(d_logdur, "loss_grads: null duration grad");
        hipLaunchKernelGGL(tok_loss_grad_kernel, dim3(gtok), dim3(256), 0, st, log_dur_pred, (const float*)nullptr, durs, in_lens, acc,
                           3, d_logdur, B, Tt, 1, grad_scale * dur_w);
    }
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, acc, losses_out, stage, dur_w, pitch_w, energy_w);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// dur_pred = clamp(exp(log_dur_pred) - 1, 0, max_duration)   (model.py:370)
__global__ void dur_from_log_kernel(const float* __restrict__ logd, float* __restrict__ out, int n, float max_dur) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = fminf(fmaxf(expf(logd[i]) - 1.f, 0.f), max_dur);
}
extern "C" int xva_fp_dur_from_log(const float* logd, float* out, int n, float max_dur, void* stream) {
    XVA_CHECK_ARG(logd && out, "dur_from_log: null");
    hipLaunchKernelGGL(dur_from_log_kernel, dim3(xva_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, logd, out, n, max_dur);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// =====================================================================================
// Final 256 -> 1 projection of a TemporalPredictor, backward (model.py:116,121): the GEMM's 16-byte
// operand alignment cannot hold for a 1-wide operand, so its two backward products are vector kernels.
//   outer:          out[r][c]  = s[r] * w[c]
//   rowscale_colsum out[c]    += sum_r s[r] * X[r][c]
// =====================================================================================
__global__ void outer_kernel(const float* __restrict__ s, const float* __restrict__ w, void* __restrict__ out, int dt, int64_t rows, int C) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    a_st(out, i, dt, s[i / C] * w[i % C]);
}
__global__ void rowscale_colsum_kernel(const void* __restrict__ X, int dt, const float* __restrict__ s, float* __restrict__ out,
                                       int64_t rows, int C, int rows_per_block) {
    __shared__ float sh[4][64];
    int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    int c = blockIdx.x * 64 + cl;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float acc = 0.f;
    if (c < C)
        for (int64_t r = r0 + rl; r < r1; r += 4) acc += s[r] * a_ld(X, r * C + c, dt);
    sh[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C) atomicAdd(out + c, sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]);
}
extern "C" int xva_fp_outer(const float* s, const float* w, void* out, int dt, int64_t rows, int C, void* stream) {
    XVA_CHECK_ARG(s && w && out, "outer: null");
    hipLaunchKernelGGL(outer_kernel, dim3(xva_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, s, w, out, dt, rows, C);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_rowscale_colsum(const void* X, int dt, const float* s, float* out, int64_t rows, int C, void* stream) {
    XVA_CHECK_ARG(X && s && out, "rowscale_colsum: null");
    const int rpb = 256;
    hipLaunchKernelGGL(rowscale_colsum_kernel, dim3(xva_cdiv(C, 64), xva_cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream, X, dt, s,
                       out, rows, C, rpb);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// fp32 master parameters -> activation-dtype shadow used as GEMM operands (bf16 mode), once per step
__global__ void cast_kernel(const float* __restrict__ src, void* __restrict__ dst, int dt, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) a_st(dst, i, dt, src[i]);
}
// fp32 -> bf16, 8 elements per thread and iteration: two 16-byte loads, one 16-byte store (the scalar form stored 2 bytes per lane: 122 us
// for the 46 M parameters of the FastPitch shadow = 2.3 TB/s)
__global__ void cast_bf16x8_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = src[2 * i], b = src[2 * i + 1];
        uint4 o;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.x) : "v"(a.x), "v"(a.y));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.y) : "v"(a.z), "v"(a.w));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.z) : "v"(b.x), "v"(b.y));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.w) : "v"(b.z), "v"(b.w));
        dst[i] = o;
    }
}
__global__ void cast_f16x8_kernel(const float4* __restrict__ src, uint4* __restrict__ dst, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = src[2 * i], b = src[2 * i + 1];
        dst[i] = make_uint4(pk_f16(a.x, a.y), pk_f16(a.z, a.w), pk_f16(b.x, b.y), pk_f16(b.z, b.w));
    }
}
// dst += src (activation dtype, element pairs; n even)
__global__ void add_act_kernel(void* __restrict__ dst, const void* __restrict__ src, int dt, int64_t npairs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npairs; i += (int64_t)gridDim.x * blockDim.x) {
        float a, b, x, y;
        a_ld2(dst, 2 * i, dt, a, b); a_ld2(src, 2 * i, dt, x, y);
        a_st2(dst, 2 * i, dt, a + x, b + y);
    }
}
extern "C" int xva_fp_add_act(void* dst, const void* src, int dt, int64_t n, void* stream) {
    XVA_CHECK_ARG(dst && src && n % 2 == 0, "add_act: null or odd length");
    if (n == 0) return XVA_OK;
    int g = (int)((n / 2 + 255) / 256); if (g > 4096) g = 4096;
    hipLaunchKernelGGL(add_act_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, dst, src, dt, n / 2);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_cast_f32(const float* src, void* dst, int dt, int64_t n, void* stream) {
    XVA_CHECK_ARG(src && dst, "cast: null");
    if ((dt == XVA_BF16 || dt == XVA_F16) && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0 && n >= 8) {
        const int64_t n8 = n / 8;
        int g8 = (int)((n8 + 255) / 256); if (g8 > 8192) g8 = 8192;
        hipLaunchKernelGGL(dt == XVA_F16 ? cast_f16x8_kernel : cast_bf16x8_kernel, dim3(g8), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(src), reinterpret_cast<uint4*>(dst), n8);
        if (n % 8 == 0) { XVA_LAUNCH_CHECK(); return XVA_OK; }
        src += n8 * 8; dst = reinterpret_cast<uint16_t*>(dst) + n8 * 8; n -= n8 * 8;
    }
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(cast_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, dt, n);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// Zero up to 40 small byte spans (multiples of 4, 4-byte aligned) in ONE launch: the guard rows of the split-bf16 pairs that live in fp32 slots (a memset per
// span was 72 stream operations per FastPitch step).  One workgroup per span.
struct xva_spans { void* p[40]; int32_t words[40]; };
__global__ void zero_spans_kernel(xva_spans sp) {
    uint32_t* d = reinterpret_cast<uint32_t*>(sp.p[blockIdx.x]);
    for (int i = threadIdx.x; i < sp.words[blockIdx.x]; i += blockDim.x) d[i] = 0u;
}
extern "C" int xva_zero_spans(void* const* ptrs, const int64_t* bytes, int n, void* stream) {
    XVA_CHECK_ARG(n >= 0 && (n == 0 || (ptrs && bytes)), "zero_spans: bad args");
    for (int i0 = 0; i0 < n; i0 += 40) {
        xva_spans sp;
        const int m = n - i0 < 40 ? n - i0 : 40;
        for (int i = 0; i < m; ++i) {
            XVA_CHECK_ARG(ptrs[i0 + i] && bytes[i0 + i] % 4 == 0 && bytes[i0 + i] < ((int64_t)1 << 32) && ((uintptr_t)ptrs[i0 + i] % 4) == 0, "zero_spans: spans are 4-byte aligned multiples of 4");
            sp.p[i] = ptrs[i0 + i]; sp.words[i] = (int32_t)(bytes[i0 + i] / 4);
        }
        hipLaunchKernelGGL(zero_spans_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, sp);
    }
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// fp32 tensor -> split-bf16 pair (hi = bf16(x) at dst[i], lo = bf16(x - hi) at dst[plane + i]): the operand format of xva_gemm's `planes` products
// (include/xva_gemm.h).  8 elements per thread and iteration.
__global__ void split_bf16x8_kernel(const float4* __restrict__ src, uint4* __restrict__ hi, uint4* __restrict__ lo, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 a = src[2 * i], b = src[2 * i + 1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        uint32_t h[4], l[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(h[e]) : "v"(v[2 * e]), "v"(v[2 * e + 1]));
            const float r0 = v[2 * e] - __uint_as_float(h[e] << 16), r1 = v[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(l[e]) : "v"(r0), "v"(r1));
        }
        hi[i] = make_uint4(h[0], h[1], h[2], h[3]);
        lo[i] = make_uint4(l[0], l[1], l[2], l[3]);
    }
}
extern "C" int xva_split_bf16(const float* src, void* dst, int64_t plane, int64_t n, void* stream) {
    if (plane == 0) return xva_cast_f32(src, dst, XVA_F16, n, stream);      // one IEEE-half tensor instead of a pair
    XVA_CHECK_ARG(src && dst && n >= 0 && n % 8 == 0 && plane % 8 == 0 && ((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "split_bf16: n and the plane offset must be multiples of 8, pointers 16-byte aligned");
    if (n == 0) return XVA_OK;
    const int64_t n8 = n / 8;
    int g8 = (int)((n8 + 255) / 256); if (g8 > 8192) g8 = 8192;
    hipLaunchKernelGGL(split_bf16x8_kernel, dim3(g8), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(src), reinterpret_cast<uint4*>(dst),
                       reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(dst) + plane), n8);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// Transposed bf16 shadow of tap-major convolution weights for the backward-data products (round 5): src fp32 [Cout][3][Cin] -> dst bf16
// [Cin][3][Cout] with the taps reversed, dst[n][m][co] = src[co][2 - m][n] — the weight of the k = 3 convolution that maps d(output) to d(input),
// so that backward-data runs the NT main loop (k-contiguous weight rows straight into LDS) instead of the NN one (transposing LDS reads of
// row segments; measured 172 us against 138 us for FastPitch's conv2 backward-data, and 1.75x the algorithmic fetch).  One launch for up to 16
// tensors; a 32 x 32 tile is transposed through LDS (coalesced 128-byte reads along Cin, 64-byte writes along Cout).
struct xva_wt_batch { int64_t src[16]; int64_t dst[16]; int n; };
__global__ __launch_bounds__(256) void wt_transpose3_kernel(const float* __restrict__ params, uint16_t* __restrict__ out, xva_wt_batch bt, int Cout, int Cin, int64_t plane, int dt16) {
    __shared__ float tile[32][33];
    const int which = blockIdx.z / 3, tap = blockIdx.z % 3;
    const float* src = params + bt.src[which];
    uint16_t* dst = out + bt.dst[which];
    const int n0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int co = c0 + r, n = n0 + tx;
        tile[r][tx] = (co < Cout && n < Cin) ? src[((int64_t)co * 3 + tap) * Cin + n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
        const int n = n0 + r, co = c0 + tx;
        if (n < Cin && co < Cout) {
            const int64_t o = ((int64_t)n * 3 + (2 - tap)) * Cout + co;
            const float v = tile[tx][r];
            a_st(dst, o, dt16, v);
            if (plane) a_st(dst, o + plane, XVA_BF16, v - a_ld(dst, o, XVA_BF16));          // the lo plane of a split-bf16 pair
        }
    }
}
static int wt_transpose3(const float* params, void* out, const int64_t* src_off, const int64_t* dst_off, int n, int Cout, int Cin, int64_t plane, void* stream, int dt16 = XVA_BF16) {
    XVA_CHECK_ARG(params && out && src_off && dst_off && n >= 1 && n <= 16, "wt_transpose3: bad arguments");
    xva_wt_batch bt;
    bt.n = n;
    for (int i = 0; i < n; ++i) { bt.src[i] = src_off[i]; bt.dst[i] = dst_off[i]; }
    hipLaunchKernelGGL(wt_transpose3_kernel, dim3(xva_cdiv(Cin, 32), xva_cdiv(Cout, 32), 3 * n), dim3(256), 0, (hipStream_t)stream, params,
                       reinterpret_cast<uint16_t*>(out), bt, Cout, Cin, plane, dt16);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_fp_wt_transpose3(const float* params, void* out, const int64_t* src_off, const int64_t* dst_off, int n, int Cout, int Cin, void* stream) {
    return wt_transpose3(params, out, src_off, dst_off, n, Cout, Cin, 0, stream);
}
// the same as a split-bf16 pair: the lo plane `plane` elements after the hi plane
extern "C" int xva_fp_wt_transpose3_planes(const float* params, void* out, const int64_t* src_off, const int64_t* dst_off, int n, int Cout, int Cin, int64_t plane, void* stream) {
    XVA_CHECK_ARG(plane >= 0, "wt_transpose3_planes: plane offset");
    if (plane == 0) return wt_transpose3(params, out, src_off, dst_off, n, Cout, Cin, 0, stream, XVA_F16);      // one IEEE-half tensor instead of a pair
    return wt_transpose3(params, out, src_off, dst_off, n, Cout, Cin, plane, stream);
}

// activation-dtype tensor -> fp32 copy (the temporal predictors run on fp32-stored tensors: their gradients are sums of
// near-cancelling terms and bf16 storage noise is amplified ~10x there; they are < 2 % of the step's bytes)
__global__ void cast_to_f32_kernel(const void* __restrict__ src, int dt, float* __restrict__ dst, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] = a_ld(src, i, dt);
}
extern "C" int xva_cast_to_f32(const void* src, int dt, float* dst, int64_t n, void* stream) {
    XVA_CHECK_ARG(src && dst, "cast: null");
    int grid = (int)((n + 255) / 256); if (grid > 4096) grid = 4096; if (grid < 1) grid = 1;
    hipLaunchKernelGGL(cast_to_f32_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dt, dst, n);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// =====================================================================================
// Inference helpers (FastPitch.infer, model.py:426-481)
// infer_finish: per item, reps[t] = (long)(dur_pred[t] * pace + 0.5) (regulate_len, model.py:62-64; 0 on padding tokens),
// dec_lens = sum reps, and the token-level predictions unpadded to (B, Tt).
// =====================================================================================
__global__ void infer_finish_kernel(const float* __restrict__ dur_pad, const float* __restrict__ pitch_pad, const float* __restrict__ energy_pad,
                                    const int* __restrict__ lens, float pace, int Tt, int* __restrict__ durs, int* __restrict__ dec_lens,
                                    float* __restrict__ dur_out, float* __restrict__ pitch_out, float* __restrict__ energy_out) {
    __shared__ float sh[16];
    const int b = blockIdx.x, Tp = Tt + 2;
    float tot = 0.f;
    for (int t = threadIdx.x; t < Tt; t += blockDim.x) {
        const bool live = t < lens[b];
        const float d = live ? dur_pad[b * Tp + t + 1] : 0.f;
        const int rep = (int)(d * pace + 0.5f);
        durs[b * Tt + t] = rep;
        tot += (float)rep;
        dur_out[b * Tt + t] = d;
        pitch_out[b * Tt + t] = live ? pitch_pad[b * Tp + t + 1] : 0.f;
        energy_out[b * Tt + t] = live ? energy_pad[b * Tp + t + 1] : 0.f;
    }
    tot = xva_block_sum(tot, sh);
    if (threadIdx.x == 0) dec_lens[b] = (int)(tot + 0.5f);
}
extern "C" int xva_fp_infer_finish(const float* dur_pad, const float* pitch_pad, const float* energy_pad, const int32_t* lens, float pace, int B,
                                   int Tt, int32_t* durs, int32_t* dec_lens, float* dur_out, float* pitch_out, float* energy_out, void* stream) {
    XVA_CHECK_ARG(dur_pad && pitch_pad && energy_pad && lens && durs && dec_lens && dur_out && pitch_out && energy_out, "infer_finish: null");
    hipLaunchKernelGGL(infer_finish_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, dur_pad, pitch_pad, energy_pad, lens, pace, Tt, durs,
                       dec_lens, dur_out, pitch_out, energy_out);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
// padded time-major mel (B, Tm + 2, C) in the activation dtype -> (B, C, Tm) fp32 (the layout inference hands to the vocoder)
__global__ void mel_tm_to_bct_kernel(const void* __restrict__ in, int dt, float* __restrict__ out, int B, int Tm, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * C * Tm) return;
    const int t = (int)(i % Tm), c = (int)((i / Tm) % C), b = (int)(i / ((int64_t)Tm * C));
    out[i] = a_ld(in, ((int64_t)b * (Tm + 2) + t + 1) * C + c, dt);
}
extern "C" int xva_fp_mel_to_bct(const void* in, int dt, float* out, int B, int Tm, int C, void* stream) {
    XVA_CHECK_ARG(in && out, "mel_to_bct: null");
    hipLaunchKernelGGL(mel_tm_to_bct_kernel, dim3((unsigned)xva_cdiv((int64_t)B * C * Tm, 256)), dim3(256), 0, (hipStream_t)stream, in, dt, out, B, Tm, C);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
