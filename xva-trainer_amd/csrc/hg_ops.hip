// hg_ops.hip — the non-GEMM kernels of the HiFi-GAN hot loop (gfx950): single-channel convolutions at the waveform
// boundary, period folding, pooling, reparametrisations (weight_norm / spectral_norm), GAN / feature / L1 losses.
// Reference: python/hifigan/models.py:17-294, python/hifigan/xva_train.py:479-515.
//
// Activations are "time-major sequences": nseq items, each Hp = padF + T + padB rows of C channels, element type
// fp32 or bf16 (dt = XVA_F32 / XVA_BF16), pad rows structurally zero.  All kernels here are HBM-bound single passes.
#include "xva_common.h"
#include "../../include/xva_gemm.h"
#include "../../include/xva_hip.h"
#include "hg_wn.h"

__device__ __forceinline__ float hg_ld(const void* p, int64_t i, int dt) {
    return dt == XVA_BF16 ? __uint_as_float(((uint32_t) reinterpret_cast<const uint16_t*>(p)[i]) << 16)
                          : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void hg_st(void* p, int64_t i, int dt, float v) {
    if (dt == XVA_BF16) {
        uint32_t u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);   // round to nearest even
        reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)(u >> 16);
    } else {
        reinterpret_cast<float*>(p)[i] = v;
    }
}
__device__ __forceinline__ float hg_lrelu(float v, float s) { return v > 0.f ? v : v * s; }
// 8 consecutive elements (16-byte aligned for bf16, 32-byte for fp32)
__device__ __forceinline__ void hg_ld8(const void* p, int64_t i, int dt, float (&v)[8]) {
    if (dt == XVA_BF16) {
        const uint4 u = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p) + i);
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(w[e] << 16); v[2 * e + 1] = __uint_as_float(w[e] & 0xffff0000u); }
    } else {
        const float4 a = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i);
        const float4 b = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p) + i + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
}
__device__ __forceinline__ uint32_t hg_pack2(float a, float b) {
    uint32_t x = __float_as_uint(a), y = __float_as_uint(b);
    x += 0x7fffu + ((x >> 16) & 1u); y += 0x7fffu + ((y >> 16) & 1u);
    return (x >> 16) | (y & 0xffff0000u);
}
__device__ __forceinline__ void hg_st8(void* p, int64_t i, int dt, const float (&v)[8]) {
    if (dt == XVA_BF16) {
        *reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p) + i) =
            make_uint4(hg_pack2(v[0], v[1]), hg_pack2(v[2], v[3]), hg_pack2(v[4], v[5]), hg_pack2(v[6], v[7]));
    } else {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(p) + i + 4) = make_float4(v[4], v[5], v[6], v[7]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// mel (B, C, T) fp32  ->  time-major sequence rows (B, Hp, C) at valid rows          (Generator input, models.py:110)
__global__ void hg_mel_to_tm_kernel(const float* __restrict__ mel, void* __restrict__ out, int dt, int B, int C, int T, int Hp,
                                    int padF) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T * C) return;
    int c = (int)(i % C);
    int t = (int)((i / C) % T);
    int b = (int)(i / ((int64_t)C * T));
    hg_st(out, ((int64_t)b * Hp + padF + t) * C + c, dt, mel[((int64_t)b * C + c) * T + t]);
}
extern "C" int xva_hg_mel_to_tm(const float* mel, void* out, int dt, int B, int C, int T, int Hp, int padF, void* stream) {
    XVA_CHECK_ARG(mel && out, "mel_to_tm: null");
    int64_t n = (int64_t)B * T * C;
    hipLaunchKernelGGL(hg_mel_to_tm_kernel, dim3(xva_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, mel, out, dt, B, C, T, Hp, padF);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Single-input-channel convolution at the waveform boundary (DiscriminatorS conv0: Conv1d(1,128,15,1,pad 7) models.py:207;
// DiscriminatorP conv0: Conv2d(1,32,(5,1),(3,1),pad (2,0)) on the wave folded to (T/p, p) with right reflect padding,
// models.py:154-163).  Sequence (b, w) of a period-p discriminator holds samples wav[b, h * p + w]; p = 1 is the plain case.
//   out[(b,w)][padF + h'][co] = lrelu(bias[co] + sum_j W[co][j] * x(b, w, s*h' + j - P))
// wav: (nb, Tw) fp32 ; reflect: samples Tw <= i < Hfold * p map to wav[2*(Tw-1) - i]; x = 0 outside [0, Hfold).
struct Cin1Geom { int nb, Tw, p, Hfold, k, s, P, Cout, Tout, Hp, padF; };

__device__ __forceinline__ float cin1_sample(const float* __restrict__ wav, const Cin1Geom& g, int b, int w, int h) {
    if (h < 0 || h >= g.Hfold) return 0.f;
    int i = h * g.p + w;
    if (i >= g.Tw) i = 2 * (g.Tw - 1) - i;
    return wav[(int64_t)b * g.Tw + i];
}
__global__ void hg_cin1_fwd_kernel(const float* __restrict__ wav, const float* __restrict__ W, const float* __restrict__ bias,
                                   void* __restrict__ out, int dt, Cin1Geom g, float slope) {
    // one block per (sequence, chunk of 64 output rows); threads over (row, co)
    const int seq = blockIdx.y, b = seq / g.p, w = seq % g.p;
    const int h0 = blockIdx.x * 64;
    for (int idx = threadIdx.x; idx < 64 * g.Cout; idx += blockDim.x) {
        int hl = idx / g.Cout, co = idx % g.Cout;
        int h = h0 + hl;
        if (h >= g.Tout) continue;
        float acc = bias[co];
        for (int j = 0; j < g.k; ++j) acc += W[co * g.k + j] * cin1_sample(wav, g, b, w, g.s * h + j - g.P);
        hg_st(out, ((int64_t)seq * g.Hp + g.padF + h) * g.Cout + co, dt, hg_lrelu(acc, slope));
    }
}
__device__ __forceinline__ float hg_round_bf16(float v) { uint32_t u = __float_as_uint(v); u += 0x7fffu + ((u >> 16) & 1u); return __uint_as_float(u & 0xffff0000u); }
// bf16 mode, Cout = 128, k <= 16 (DiscriminatorS conv0, models.py:207): the same forward on the matrix pipe WITHOUT an im2col in memory.  (Round 5's first form
// of this layer — a packed-fp32 VALU kernel — gave sporadically wrong even channels on the discriminators' side stream lanes and was bound by VALU issue anyway;
// its cause was never isolated and the kernel is gone from the library, round 6; tests/test_lanes_gpu.py holds every engine's lanes-on results to its one-stream
// results bit for bit instead.)  Here a wave takes 16 rows at a
// time: the lane's 4 consecutive taps of its row come straight from the staged waveform (LDS, already rounded to bf16: the operands of the GEMM form), eight
// v_mfma_f32_16x16x16_bf16 (taps padded 15 -> 16 with a zero weight) give the 128 channels, and the weight rows are PERMUTED over the eight tiles so that a lane ends
// up with 4 x 8 consecutive channels of its row: four 16-byte stores, 64 contiguous bytes per row and instruction.  Products are the GEMM form's (bf16 x bf16 exact in
// fp32), the summation order is the matrix pipe's.
typedef short hg_s4 __attribute__((ext_vector_type(4)));
typedef float hg_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ short hg_bf16_bits(float v) { return (short)(__float_as_uint(hg_round_bf16(v)) >> 16); }
constexpr int CIN1_MFMA_ROWS = 512;      // rows per workgroup (4 waves x 8 blocks of 16 rows)
template <int NT>                        // Cout = 16 NT: 8 tiles (DiscriminatorS, 128 channels) or 2 (DiscriminatorP, 32 channels; k = 5: 11 zero taps — the matrix pipe is idle anyway)
__global__ __launch_bounds__(256) void hg_cin1_fwd_mfma_kernel(const float* __restrict__ wav, const float* __restrict__ W, const float* __restrict__ bias,
                                                               uint16_t* __restrict__ out, Cin1Geom g, float slope) {
    constexpr int COUT = 16 * NT, NQ = NT / 2;                      // NQ 16-byte pieces per lane and row
    extern __shared__ float sx[];                                   // samples s * h0 - P ... of this block's rows (+ 16 of slack: the zero-weight taps)
    const int seq = blockIdx.y, b = seq / g.p, wf = seq % g.p;
    const int h0 = blockIdx.x * CIN1_MFMA_ROWS;
    const int nsmp = g.s * (CIN1_MFMA_ROWS - 1) + 16;
    for (int i = threadIdx.x; i < nsmp; i += 256) sx[i] = hg_round_bf16(cin1_sample(wav, g, b, wf, g.s * h0 - g.P + i));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 15, gq = lane >> 4, k0 = gq * 4;
    // tile j = 2 q + hh holds, as its row m = 4 gg + i, channel q * 32 + gg * 8 + hh * 4 + i: the lane with lane >> 4 == gg then owns channels q * 32 + gg * 8 .. + 7 of its row
    hg_s4 wfr[NT];
    float bs[NT][4];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = j >> 1, hh = j & 1;
        const int co = q * 32 + (m >> 2) * 8 + hh * 4 + (m & 3);
#pragma unroll
        for (int t = 0; t < 4; ++t) wfr[j][t] = (k0 + t < g.k) ? hg_bf16_bits(W[(int64_t)co * g.k + k0 + t]) : (short)0;
#pragma unroll
        for (int i = 0; i < 4; ++i) bs[j][i] = bias[q * 32 + gq * 8 + hh * 4 + i];
    }
    __syncthreads();
    for (int blk = wave; blk < CIN1_MFMA_ROWS / 16; blk += 4) {
        const int n = blk * 16 + m, h = h0 + n;
        if (h0 + blk * 16 >= g.Tout) break;
        const float* xs = sx + g.s * n + k0;
        hg_s4 xf;
#pragma unroll
        for (int t = 0; t < 4; ++t) xf[t] = (short)(__float_as_uint(xs[t]) >> 16);
        hg_f4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wfr[j], xf, (hg_f4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (h < g.Tout) {
            uint16_t* dst = out + ((int64_t)seq * g.Hp + g.padF + h) * COUT + gq * 8;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                uint32_t pk[4];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const int j = 2 * q + hh;
                    const float v0 = hg_lrelu(acc[j][0] + bs[j][0], slope), v1 = hg_lrelu(acc[j][1] + bs[j][1], slope);
                    const float v2 = hg_lrelu(acc[j][2] + bs[j][2], slope), v3 = hg_lrelu(acc[j][3] + bs[j][3], slope);
                    pk[2 * hh] = (__float_as_uint(hg_round_bf16(v0)) >> 16) | (__float_as_uint(hg_round_bf16(v1)) & 0xffff0000u);
                    pk[2 * hh + 1] = (__float_as_uint(hg_round_bf16(v2)) >> 16) | (__float_as_uint(hg_round_bf16(v3)) & 0xffff0000u);
                }
                *reinterpret_cast<uint4*>(dst + q * 32) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
            }
        }
    }
}
// dW[co][j] += sum dY[seq][h][co] * x(seq, s*h + j - P) ; db[co] += sum dY   (dY = gradient w.r.t. the pre-activation)
__global__ void hg_cin1_bwd_weight_kernel(const float* __restrict__ wav, const void* __restrict__ dY, int dt, float* __restrict__ dW,
                                          float* __restrict__ db, Cin1Geom g) {
    const int seq = blockIdx.y, b = seq / g.p, w = seq % g.p;
    const int h0 = blockIdx.x * 256, h1 = min(h0 + 256, g.Tout);
    for (int idx = threadIdx.x; idx < g.Cout * (g.k + 1); idx += blockDim.x) {
        int co = idx / (g.k + 1), j = idx % (g.k + 1);
        float acc = 0.f;
        for (int h = h0; h < h1; ++h) {
            float d = hg_ld(dY, ((int64_t)seq * g.Hp + g.padF + h) * g.Cout + co, dt);
            acc += (j == g.k) ? d : d * cin1_sample(wav, g, b, w, g.s * h + j - g.P);
        }
        if (j == g.k) atomicAdd(db + co, acc); else atomicAdd(dW + co * g.k + j, acc);
    }
}
// d_wav[b][i] (+)= sum over folded positions aliasing sample i of sum_{j, co} dY[(b,w)][h'][co] W[co][j], s*h' + j - P = h
__global__ void hg_cin1_bwd_data_kernel(const void* __restrict__ dY, int dt, const float* __restrict__ W, float* __restrict__ dwav,
                                        Cin1Geom g, int accumulate) {
    extern __shared__ float sW[];   // Cout * k
    for (int i = threadIdx.x; i < g.Cout * g.k; i += blockDim.x) sW[i] = W[i];
    __syncthreads();
    int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= (int64_t)g.nb * g.Tw) return;
    int b = (int)(gi / g.Tw), i = (int)(gi % g.Tw);
    float total = 0.f;
    for (int alias = 0; alias < 2; ++alias) {
        int ii = i;
        if (alias == 1) { ii = 2 * (g.Tw - 1) - i; if (ii < g.Tw || ii >= g.Hfold * g.p) break; }
        int h = ii / g.p, w = ii % g.p;
        int seq = b * g.p + w;
        for (int j = 0; j < g.k; ++j) {
            int num = h + g.P - j;
            if (num < 0 || num % g.s != 0) continue;
            int hp = num / g.s;
            if (hp >= g.Tout) continue;
            int64_t base = ((int64_t)seq * g.Hp + g.padF + hp) * g.Cout;
            float acc = 0.f;
            for (int co = 0; co < g.Cout; ++co) acc += hg_ld(dY, base + co, dt) * sW[co * g.k + j];
            total += acc;
        }
    }
    if (accumulate) dwav[gi] += total; else dwav[gi] = total;
}
static int cin1_geom(Cin1Geom* g, int nb, int Tw, int p, int k, int s, int P, int Cout, int Hp, int padF) {
    XVA_CHECK_ARG(nb > 0 && Tw > 1 && p >= 1 && k >= 1 && s >= 1 && Cout >= 1, "cin1: bad geometry");
    g->nb = nb; g->Tw = Tw; g->p = p; g->k = k; g->s = s; g->P = P; g->Cout = Cout; g->Hp = Hp; g->padF = padF;
    g->Hfold = (Tw + p - 1) / p;
    g->Tout = (g->Hfold + 2 * P - k) / s + 1;
    XVA_CHECK_ARG(padF + g->Tout <= Hp, "cin1: output rows do not fit Hp");
    return XVA_OK;
}
extern "C" int xva_hg_cin1_out_len(int Tw, int p, int k, int s, int P) { return ((Tw + p - 1) / p + 2 * P - k) / s + 1; }
extern "C" int xva_hg_cin1_fwd(const float* wav, const float* W, const float* bias, void* out, int dt, int nb, int Tw, int p, int k,
                               int s, int P, int Cout, int Hp, int padF, float slope, void* stream) {
    Cin1Geom g;
    XVA_TRY(cin1_geom(&g, nb, Tw, p, k, s, P, Cout, Hp, padF));
    XVA_CHECK_ARG(wav && W && bias && out, "cin1_fwd: null");
    if (dt == XVA_BF16 && (Cout == 128 || Cout == 32) && k <= 16 && (((uintptr_t)out) % 16) == 0 && ((int64_t)Hp * Cout) % 8 == 0) {
        const size_t lds = (size_t)(s * (CIN1_MFMA_ROWS - 1) + 16) * sizeof(float);
        const dim3 grid(xva_cdiv(g.Tout, CIN1_MFMA_ROWS), nb * p);
        if (Cout == 128) hipLaunchKernelGGL(hg_cin1_fwd_mfma_kernel<8>, grid, dim3(256), lds, (hipStream_t)stream, wav, W, bias, (uint16_t*)out, g, slope);
        else hipLaunchKernelGGL(hg_cin1_fwd_mfma_kernel<2>, grid, dim3(256), lds, (hipStream_t)stream, wav, W, bias, (uint16_t*)out, g, slope);
        XVA_LAUNCH_CHECK();
        return XVA_OK;
    }
    hipLaunchKernelGGL(hg_cin1_fwd_kernel, dim3(xva_cdiv(g.Tout, 64), nb * p), dim3(256), 0, (hipStream_t)stream, wav, W, bias, out, dt, g, slope);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_cin1_bwd_weight(const float* wav, const void* dY, int dt, float* dW, float* db, int nb, int Tw, int p, int k, int s,
                                      int P, int Cout, int Hp, int padF, void* stream) {
    Cin1Geom g;
    XVA_TRY(cin1_geom(&g, nb, Tw, p, k, s, P, Cout, Hp, padF));
    XVA_CHECK_ARG(wav && dY && dW && db, "cin1_bwd_weight: null");
    hipLaunchKernelGGL(hg_cin1_bwd_weight_kernel, dim3(xva_cdiv(g.Tout, 256), nb * p), dim3(256), 0, (hipStream_t)stream, wav, dY, dt, dW, db, g);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_cin1_bwd_data(const void* dY, int dt, const float* W, float* dwav, int nb, int Tw, int p, int k, int s, int P,
                                    int Cout, int Hp, int padF, int accumulate, void* stream) {
    Cin1Geom g;
    XVA_TRY(cin1_geom(&g, nb, Tw, p, k, s, P, Cout, Hp, padF));
    XVA_CHECK_ARG(dY && W && dwav, "cin1_bwd_data: null");
    int64_t n = (int64_t)nb * Tw;
    hipLaunchKernelGGL(hg_cin1_bwd_data_kernel, dim3(xva_cdiv(n, 256)), dim3(256), Cout * k * sizeof(float), (hipStream_t)stream, dY, dt, W,
                       dwav, g, accumulate);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Single-OUTPUT-channel convolution backward (conv_post layers: models.py:108,152,216).  Forward runs on the GEMM (N = 1).
// Rows are the merged row index of the input sequence tensor x (rows = nseq * Hp, C channels); d is the gradient of the
// 1-channel output on the same row grid.
//   dX[r][c] = gate(x[r][c]) * sum_j d[r + P - j*dil] * w[j*C + c]      (valid rows only, pads -> 0)
//   dw[j*C + c] += sum_r d[r] * act(x[r + j*dil - P][c]) ; db += sum_r d[r]
__global__ void hg_cout1_bwd_data_kernel(const void* __restrict__ d, const float* __restrict__ w, const void* __restrict__ x,
                                         void* __restrict__ dX, int dt, int64_t rows, int C, int k, int dil, int P, int Hp, int padF,
                                         int T, int gate, float slope) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C) return;
    int64_t r = i / C;
    int c = (int)(i % C);
    int t = (int)(r % Hp);
    float v = 0.f;
    if (t >= padF && t < padF + T) {
        for (int j = 0; j < k; ++j) {
            int64_t rr = r + P - (int64_t)j * dil;
            if (rr < 0 || rr >= rows) continue;
            v += hg_ld(d, rr, dt) * w[j * C + c];
        }
        if (gate && !(hg_ld(x, i, dt) > 0.f)) v *= slope;
    }
    hg_st(dX, i, dt, v);
}
// bf16 tensors, C % 8 == 0: a thread owns 8 consecutive channels of a row — one 16-byte load of x (the gate), one 16-byte store of dX, the
// taps' weights as float4 pairs (the element-per-thread form moved 2 bytes per lane: 46 us average over the conv_post layers of a step)
__global__ void hg_cout1_bwd_data_bf16x8_kernel(const uint16_t* __restrict__ d, const float* __restrict__ w, const uint16_t* __restrict__ x,
                                                uint16_t* __restrict__ dX, int64_t rows, int C, int k, int dil, int P, int Hp, int padF, int T, int gate,
                                                float slope) {
    const int C8 = C >> 3;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * C8) return;
    const int64_t r = i / C8;
    const int c = (int)(i - r * C8) << 3;
    const int t = (int)(r % Hp);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (t >= padF && t < padF + T) {
        for (int j = 0; j < k; ++j) {
            const int64_t rr = r + P - (int64_t)j * dil;
            if (rr < 0 || rr >= rows) continue;
            const float dv = __uint_as_float((uint32_t)d[rr] << 16);
            const float4 w0 = *reinterpret_cast<const float4*>(w + (int64_t)j * C + c), w1 = *reinterpret_cast<const float4*>(w + (int64_t)j * C + c + 4);
            v[0] += dv * w0.x; v[1] += dv * w0.y; v[2] += dv * w0.z; v[3] += dv * w0.w;
            v[4] += dv * w1.x; v[5] += dv * w1.y; v[6] += dv * w1.z; v[7] += dv * w1.w;
        }
        if (gate) {
            const uint4 xr = *reinterpret_cast<const uint4*>(x + r * C + c);
            const uint32_t xw[4] = {xr.x, xr.y, xr.z, xr.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!(__uint_as_float(xw[e] << 16) > 0.f)) v[2 * e] *= slope;
                if (!(__uint_as_float(xw[e] & 0xffff0000u) > 0.f)) v[2 * e + 1] *= slope;
            }
        }
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.x) : "v"(v[0]), "v"(v[1]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.y) : "v"(v[2]), "v"(v[3]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.z) : "v"(v[4]), "v"(v[5]));
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o.w) : "v"(v[6]), "v"(v[7]));
    }
    *reinterpret_cast<uint4*>(dX + r * C + c) = o;
}
#define COUT1_MAXK 8
__global__ void hg_cout1_bwd_weight_kernel(const void* __restrict__ d, const void* __restrict__ x, float* __restrict__ dw,
                                           float* __restrict__ db, int dt, int64_t rows, int C, int k, int dil, int P, int act,
                                           float slope, int rows_per_block, int cpb) {
    // dw[j*C + c] = sum_r x[r][c] * d[r - j*dil + P].  A workgroup covers cpb (power of two <= 256) channels x 256 / cpb row lanes:
    // thread (c, rl) reads x[r][c] ONCE per row r = r0 + rl + i * (256 / cpb) and the k shifted d values from LDS; the row lanes
    // are combined through LDS before one global atomic per (tap, channel).
    __shared__ float sd[1024 + 64];
    __shared__ float sacc[COUT1_MAXK][256];
    const int cl = threadIdx.x & (cpb - 1), rl = threadIdx.x / cpb, nrl = blockDim.x / cpb;
    const int c = blockIdx.x * cpb + cl;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const int span = (k - 1) * dil;
    const int64_t dlo = r0 - span + P;
    const int nd = (int)(r1 - r0) + span;
    for (int i = threadIdx.x; i < nd; i += blockDim.x) {
        int64_t rr = dlo + i;
        sd[i] = (rr >= 0 && rr < rows) ? hg_ld(d, rr, dt) : 0.f;
    }
    __syncthreads();
    float acc[COUT1_MAXK];
#pragma unroll
    for (int j = 0; j < COUT1_MAXK; ++j) acc[j] = 0.f;
    if (c < C) {
        auto step = [&](int64_t r, float xv) {
            if (act) xv = hg_lrelu(xv, slope);
            const int base = (int)(r - r0) + span;   // index of d[r + P] in sd
#pragma unroll
            for (int j = 0; j < COUT1_MAXK; ++j)
                if (j < k) acc[j] += xv * sd[base - j * dil];
        };
        int64_t r = r0 + rl;
        for (; r + 3 * nrl < r1; r += 4 * nrl) {                  // four rows' loads in flight (one per iteration was latency-bound: 50 us)
            const float x0 = hg_ld(x, r * C + c, dt), x1 = hg_ld(x, (r + nrl) * C + c, dt), x2 = hg_ld(x, (r + 2 * nrl) * C + c, dt),
                        x3 = hg_ld(x, (r + 3 * nrl) * C + c, dt);
            step(r, x0); step(r + nrl, x1); step(r + 2 * nrl, x2); step(r + 3 * nrl, x3);
        }
        for (; r < r1; r += nrl) step(r, hg_ld(x, r * C + c, dt));
    }
#pragma unroll
    for (int j = 0; j < COUT1_MAXK; ++j) sacc[j][threadIdx.x] = acc[j];
    __syncthreads();
    for (int idx = threadIdx.x; idx < k * cpb; idx += blockDim.x) {
        const int j = idx / cpb, cc = idx - j * cpb;
        float v = 0.f;
        for (int q = 0; q < nrl; ++q) v += sacc[j][q * cpb + cc];
        if (blockIdx.x * cpb + cc < C && v != 0.f) atomicAdd(dw + j * C + blockIdx.x * cpb + cc, v);
    }
    if (blockIdx.x == 0) {   // db += sum_r d[r] ; d[r] = sd[(r - r0) + span - P]
        __shared__ float shb[16];
        float accb = 0.f;
        for (int i = threadIdx.x; i < (int)(r1 - r0); i += blockDim.x) accb += sd[i + span - P];
        accb = xva_block_sum(accb, shb);
        if (threadIdx.x == 0 && accb != 0.f) atomicAdd(db, accb);
    }
}
extern "C" int xva_hg_cout1_bwd_data(const void* d, const float* w, const void* x, void* dX, int dt, int64_t rows, int C, int k, int dil,
                                     int P, int Hp, int padF, int T, int gate, float slope, void* stream) {
    XVA_CHECK_ARG(d && w && x && dX, "cout1_bwd_data: null");
    if (dt == XVA_BF16 && C % 8 == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)dX % 16) == 0 && ((uintptr_t)w % 16) == 0) {
        hipLaunchKernelGGL(hg_cout1_bwd_data_bf16x8_kernel, dim3(xva_cdiv(rows * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)d, w,
                           (const uint16_t*)x, (uint16_t*)dX, rows, C, k, dil, P, Hp, padF, T, gate, slope);
        XVA_LAUNCH_CHECK();
        return XVA_OK;
    }
    hipLaunchKernelGGL(hg_cout1_bwd_data_kernel, dim3(xva_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, d, w, x, dX, dt, rows, C, k,
                       dil, P, Hp, padF, T, gate, slope);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_cout1_bwd_weight(const void* d, const void* x, float* dw, float* db, int dt, int64_t rows, int C, int k, int dil, int P,
                                       int act, float slope, void* stream) {
    XVA_CHECK_ARG(d && x && dw && db, "cout1_bwd_weight: null");
    XVA_CHECK_ARG(k <= COUT1_MAXK && (k - 1) * dil <= 64, "cout1_bwd_weight: kernel size unsupported");
    int cpb = 256;
    while (cpb > 1 && cpb / 2 >= C) cpb /= 2;
    // ~1024 workgroups whatever the tensor shape (rows may be a few thousand with C = 1024, or half a million with C = 32)
    int64_t rpb64 = xva_cdiv(rows, xva_cdiv(1024, xva_cdiv(C, cpb)));
    int rpb = rpb64 < 32 ? 32 : (rpb64 > 1024 ? 1024 : (int)rpb64);
    hipLaunchKernelGGL(hg_cout1_bwd_weight_kernel, dim3(xva_cdiv(C, cpb), xva_cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream, d, x, dw, db, dt,
                       rows, C, k, dil, P, act, slope, rpb, cpb);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// AvgPool1d(4, 2, padding=2) on waveforms (count_include_pad) and its backward   (MultiScaleDiscriminator, models.py:240-249)
__global__ void hg_avgpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int nb, int T, int To) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nb * To) return;
    int b = (int)(i / To), t = (int)(i % To);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) { int k = 2 * t - 2 + j; if (k >= 0 && k < T) s += x[(int64_t)b * T + k]; }
    y[i] = 0.25f * s;
}
__global__ void hg_avgpool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int nb, int T, int To, int accumulate) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nb * T) return;
    int b = (int)(i / T), k = (int)(i % T);
    float s = 0.f;
    for (int j = 0; j < 4; ++j) { int num = k + 2 - j; if (num >= 0 && num % 2 == 0 && num / 2 < To) s += dy[(int64_t)b * To + num / 2]; }
    s *= 0.25f;
    if (accumulate) dx[i] += s; else dx[i] = s;
}
extern "C" int xva_hg_avgpool_fwd(const float* x, float* y, int nb, int T, void* stream) {
    int To = T / 2 + 1;
    hipLaunchKernelGGL(hg_avgpool_fwd_kernel, dim3(xva_cdiv((int64_t)nb * To, 256)), dim3(256), 0, (hipStream_t)stream, x, y, nb, T, To);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_avgpool_bwd(const float* dy, float* dx, int nb, int T, int accumulate, void* stream) {
    int To = T / 2 + 1;
    hipLaunchKernelGGL(hg_avgpool_bwd_kernel, dim3(xva_cdiv((int64_t)nb * T, 256)), dim3(256), 0, (hipStream_t)stream, dy, dx, nb, T, To, accumulate);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Reductions over the valid region of sequence tensors (losses, models.py:263-294; xva_train.py:504).
// mode 0: sum |a - b|   mode 1: sum (1 - a)^2   mode 2: sum a^2        (b unused for 1, 2)
// The valid region of sequence s is ONE contiguous span of T * C elements starting at (s * Hp + padF) * C: blockIdx.y walks the
// sequences, blockIdx.x the span in 8-element vectors (VEC) or single elements — no per-element index arithmetic.
__device__ __forceinline__ float hg_red_term(float av, float bv, int mode) {
    if (mode == 0) return fabsf(av - bv);
    if (mode == 1) return (1.f - av) * (1.f - av);
    return av * av;
}
template <bool VEC>
__device__ __forceinline__ void hg_reduce_body(const void* __restrict__ a, const void* __restrict__ b, int dt, int Hp, int padF, int T, int C, int mode,
                                               float scale, float* __restrict__ out, int bx, int by, int gx) {
    __shared__ float sh[16];
    const int64_t base = ((int64_t)by * Hp + padF) * C;
    const int64_t L = (int64_t)T * C;
    float acc = 0.f;
    if (VEC) {
        // four 16-byte chunks per operand in flight per thread: with one, the ~130 K threads a big feature map gets cover 4 MB of loads,
        // half of what HBM needs outstanding (measured 1.7 TB/s)
        const int64_t step = (int64_t)gx * blockDim.x * 8;
        int64_t i = ((int64_t)bx * blockDim.x + threadIdx.x) * 8;
        for (; i + 3 * step < L; i += 4 * step) {
            float av[4][8], bv[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) { hg_ld8(a, base + i + u * step, dt, av[u]); if (mode == 0) hg_ld8(b, base + i + u * step, dt, bv[u]); }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += hg_red_term(av[u][e], mode == 0 ? bv[u][e] : 0.f, mode);
        }
        for (; i < L; i += step) {
            float av[8], bv[8];
            hg_ld8(a, base + i, dt, av);
            if (mode == 0) hg_ld8(b, base + i, dt, bv);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += hg_red_term(av[e], mode == 0 ? bv[e] : 0.f, mode);
        }
    } else {
        for (int64_t i = (int64_t)bx * blockDim.x + threadIdx.x; i < L; i += (int64_t)gx * blockDim.x)
            acc += hg_red_term(hg_ld(a, base + i, dt), mode == 0 ? hg_ld(b, base + i, dt) : 0.f, mode);
    }
    acc = xva_block_sum(acc, sh);
    if (threadIdx.x == 0 && acc != 0.f) atomicAdd(out, acc * scale);
}
template <bool VEC>
__global__ void hg_reduce_kernel(const void* __restrict__ a, const void* __restrict__ b, int dt, int Hp, int padF, int T, int C, int mode,
                                 float scale, float* __restrict__ out) {
    hg_reduce_body<VEC>(a, b, dt, Hp, padF, T, C, mode, scale, out, blockIdx.x, blockIdx.y, gridDim.x);
}
// many tensors, one launch: workgroup id -> (tensor, x, sequence)
__global__ void hg_reduce_batch_kernel(xva_red_batch bt) {
    int l = 0;
    while (l + 1 < bt.n && (int)blockIdx.x >= bt.d[l + 1].block0) ++l;
    const xva_red_desc& d = bt.d[l];
    const int local = (int)blockIdx.x - d.block0;
    const int bx = local % d.gx, by = local / d.gx;
    if (d.vec) hg_reduce_body<true>(d.a, d.b, d.dt, d.Hp, d.padF, d.T, d.C, d.mode, d.scale, d.out, bx, by, d.gx);
    else hg_reduce_body<false>(d.a, d.b, d.dt, d.Hp, d.padF, d.T, d.C, d.mode, d.scale, d.out, bx, by, d.gx);
}
static inline bool hg_vec8_ok(const void* p, int dt, int C, int T, int Hp, int padF) {
    const int es = dt == XVA_BF16 ? 2 : 4;
    return p == nullptr || (((uintptr_t)p % (8 * es)) == 0 && ((int64_t)T * C) % 8 == 0 && ((int64_t)Hp * C) % 8 == 0 && ((int64_t)padF * C) % 8 == 0);
}
extern "C" int xva_hg_reduce(const void* a, const void* b, int dt, int nseq, int Hp, int padF, int T, int C, int mode, float scale, float* out, void* stream) {
    XVA_CHECK_ARG(a && out && (mode != 0 || b), "hg_reduce: null");
    XVA_CHECK_ARG(nseq <= 65535, "hg_reduce: too many sequences");
    if (nseq <= 0 || T <= 0) return XVA_OK;
    const int64_t L = (int64_t)T * C;
    const bool vec = hg_vec8_ok(a, dt, C, T, Hp, padF) && hg_vec8_ok(b, dt, C, T, Hp, padF);
    int gx = (int)((L / (vec ? 8 : 1) + 255) / 256);
    const int cap = 512 / nseq > 1 ? 512 / nseq : 1;    // every workgroup ends in ONE atomic on the same address: keep them few
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    if (vec) hipLaunchKernelGGL((hg_reduce_kernel<true>), dim3(gx, nseq), dim3(256), 0, (hipStream_t)stream, a, b, dt, Hp, padF, T, C, mode, scale, out);
    else hipLaunchKernelGGL((hg_reduce_kernel<false>), dim3(gx, nseq), dim3(256), 0, (hipStream_t)stream, a, b, dt, Hp, padF, T, C, mode, scale, out);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_reduce_batch(const xva_red_desc* descs, int n, void* stream) {
    XVA_CHECK_ARG(descs || n == 0, "hg_reduce_batch: null");
    for (int i0 = 0; i0 < n; i0 += XVA_RED_BATCH) {
        xva_red_batch bt;
        bt.n = n - i0 < XVA_RED_BATCH ? n - i0 : XVA_RED_BATCH;
        int blocks = 0;
        for (int i = 0; i < bt.n; ++i) {
            xva_red_desc& d = bt.d[i];
            d = descs[i0 + i];
            XVA_CHECK_ARG(d.a && d.out && (d.mode != 0 || d.b) && d.nseq > 0 && d.T > 0, "hg_reduce_batch: bad descriptor %d", i0 + i);
            const int64_t L = (int64_t)d.T * d.C;
            d.vec = hg_vec8_ok(d.a, d.dt, d.C, d.T, d.Hp, d.padF) && hg_vec8_ok(d.b, d.dt, d.C, d.T, d.Hp, d.padF);
            int gx = (int)((L / (d.vec ? 8 : 1) + 255) / 256);
            const int cap = 512 / d.nseq > 1 ? 512 / d.nseq : 1;
            if (gx > cap) gx = cap;
            if (gx < 1) gx = 1;
            d.gx = gx; d.block0 = blocks;
            blocks += gx * d.nseq;
        }
        hipLaunchKernelGGL(hg_reduce_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bt);
    }
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
// Gradient seeds / additions on the valid region of the FAKE half of a discriminator tensor:
//   dY (+)= c_fm * sign(g - r) [* lrelu'(g) if gated] + (mode 1: c_gan * 2 (g - 1) ; mode 2: c_gan * 2 g ; mode 3 (real): c_gan * 2 (r - 1))
// r, g: real / fake tensors (same geometry); dY: gradient tensor of the same geometry.  init: 0 -> accumulate, 1 -> overwrite.
__device__ __forceinline__ float hg_seed_term(float rv, float gv, float old, float c_fm, float c_gan, int gan_mode, int gated, float slope, int init) {
    float v = 0.f;
    if (c_fm != 0.f) { float df = gv - rv; v += c_fm * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f)); }
    if (gan_mode == 1) v += c_gan * 2.f * (gv - 1.f);
    else if (gan_mode == 2) v += c_gan * 2.f * gv;
    else if (gan_mode == 3) v += c_gan * 2.f * (rv - 1.f);
    if (!init) v += old;
    if (gated) { float ref = (gan_mode == 3) ? rv : gv; if (!(ref > 0.f)) v *= slope; }
    return v;
}
template <bool VEC>
__global__ void hg_seed_grad_kernel(const void* __restrict__ r, const void* __restrict__ g, void* __restrict__ dY, int dt, int Hp, int padF,
                                    int T, int C, float c_fm, float c_gan, int gan_mode, int gated, float slope, int init) {
    const int64_t base = ((int64_t)blockIdx.y * Hp + padF) * C;
    const int64_t L = (int64_t)T * C;
    if (VEC) {
        for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < L; i += (int64_t)gridDim.x * blockDim.x * 8) {
            float gv[8], rv[8], ov[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { gv[e] = 0.f; rv[e] = 0.f; ov[e] = 0.f; }
            if (g) hg_ld8(g, base + i, dt, gv);
            if (r) hg_ld8(r, base + i, dt, rv);
            if (!init) hg_ld8(dY, base + i, dt, ov);
#pragma unroll
            for (int e = 0; e < 8; ++e) ov[e] = hg_seed_term(rv[e], gv[e], ov[e], c_fm, c_gan, gan_mode, gated, slope, init);
            hg_st8(dY, base + i, dt, ov);
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < L; i += (int64_t)gridDim.x * blockDim.x) {
            const int64_t idx = base + i;
            const float gv = g ? hg_ld(g, idx, dt) : 0.f, rv = r ? hg_ld(r, idx, dt) : 0.f;
            hg_st(dY, idx, dt, hg_seed_term(rv, gv, init ? 0.f : hg_ld(dY, idx, dt), c_fm, c_gan, gan_mode, gated, slope, init));
        }
    }
}
extern "C" int xva_hg_seed_grad(const void* r, const void* g, void* dY, int dt, int nseq, int Hp, int padF, int T, int C, float c_fm,
                                float c_gan, int gan_mode, int gated, float slope, int init, void* stream) {
    XVA_CHECK_ARG(dY, "hg_seed_grad: null");
    XVA_CHECK_ARG(nseq <= 65535, "hg_seed_grad: too many sequences");
    if (nseq <= 0 || T <= 0) return XVA_OK;
    const int64_t L = (int64_t)T * C;
    const bool vec = hg_vec8_ok(r, dt, C, T, Hp, padF) && hg_vec8_ok(g, dt, C, T, Hp, padF) && hg_vec8_ok(dY, dt, C, T, Hp, padF);
    int gx = (int)((L / (vec ? 8 : 1) + 255) / 256);
    const int cap = 8192 / nseq > 1 ? 8192 / nseq : 1;
    if (gx > cap) gx = cap;
    if (gx < 1) gx = 1;
    if (vec) hipLaunchKernelGGL((hg_seed_grad_kernel<true>), dim3(gx, nseq), dim3(256), 0, (hipStream_t)stream, r, g, dY, dt, Hp, padF, T, C, c_fm, c_gan, gan_mode, gated, slope, init);
    else hipLaunchKernelGGL((hg_seed_grad_kernel<false>), dim3(gx, nseq), dim3(256), 0, (hipStream_t)stream, r, g, dY, dt, Hp, padF, T, C, c_fm, c_gan, gan_mode, gated, slope, init);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// wav (nb, T) fp32 <-> 1-channel sequence tensor ; tanh backward
__global__ void hg_seq1_to_wav_kernel(const void* __restrict__ s, int dt, float* __restrict__ wav, int nb, int T, int Hp, int padF) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nb * T) return;
    int b = (int)(i / T), t = (int)(i % T);
    wav[i] = hg_ld(s, (int64_t)b * Hp + padF + t, dt);
}
// d_pre[seq rows] = d_wav * (1 - y^2), y = tanh output (1-channel sequence tensor)
__global__ void hg_tanh_bwd_kernel(const float* __restrict__ dwav, const void* __restrict__ y, void* __restrict__ dpre, int dt, int nb, int T,
                                   int Hp, int padF) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)nb * T) return;
    int b = (int)(i / T), t = (int)(i % T);
    int64_t idx = (int64_t)b * Hp + padF + t;
    float yv = hg_ld(y, idx, dt);
    hg_st(dpre, idx, dt, dwav[i] * (1.f - yv * yv));
}
extern "C" int xva_hg_seq1_to_wav(const void* s, int dt, float* wav, int nb, int T, int Hp, int padF, void* stream) {
    hipLaunchKernelGGL(hg_seq1_to_wav_kernel, dim3(xva_cdiv((int64_t)nb * T, 256)), dim3(256), 0, (hipStream_t)stream, s, dt, wav, nb, T, Hp, padF);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_tanh_bwd(const float* dwav, const void* y, void* dpre, int dt, int nb, int T, int Hp, int padF, void* stream) {
    hipLaunchKernelGGL(hg_tanh_bwd_kernel, dim3(xva_cdiv((int64_t)nb * T, 256)), dim3(256), 0, (hipStream_t)stream, dwav, y, dpre, dt, nb, T, Hp, padF);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// dst = lrelu(src, slope) over a whole sequence tensor (n % 8 == 0): the activated copy of a residual-stream tensor where the producing
// GEMM cannot store it itself (exact-fp32 parity mode; the direct-to-LDS kernels write it from their epilogue, xva_gemm_params.C2)
__global__ void hg_lrelu_copy_kernel(const void* __restrict__ src, void* __restrict__ dst, int dt, int64_t n, float slope) {
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8; i < n; i += (int64_t)gridDim.x * 2048) {
        float v[8];
        hg_ld8(src, i, dt, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] > 0.f ? v[e] : v[e] * slope;
        hg_st8(dst, i, dt, v);
    }
}
extern "C" int xva_hg_lrelu_copy(const void* src, void* dst, int dt, int64_t n, float slope, void* stream) {
    XVA_CHECK_ARG(n % 8 == 0, "lrelu_copy: n must be a multiple of 8");
    int64_t nb = xva_cdiv(n / 8, 256); if (nb > 4096) nb = 4096; if (nb < 1) nb = 1;
    hipLaunchKernelGGL(hg_lrelu_copy_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, src, dst, dt, n, slope);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// column sums over ALL rows of a sequence tensor (bias gradients; pad rows are zero): out[c] += sum_r X[r][c]
__global__ void hg_colsum_kernel(const void* __restrict__ X, int dt, float* __restrict__ out, int64_t rows, int C, int rows_per_block, float scale) {
    __shared__ float sh[4][64];
    int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    int c = blockIdx.x * 64 + cl;
    int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    float acc = 0.f;
    if (c < C)
        for (int64_t r = r0 + rl; r < r1; r += 4) acc += hg_ld(X, r * C + c, dt);
    sh[rl][cl] = acc;
    __syncthreads();
    if (rl == 0 && c < C) atomicAdd(out + c, scale * (sh[0][cl] + sh[1][cl] + sh[2][cl] + sh[3][cl]));
}
// column pairs: a wave reads 128 adjacent columns (4-byte / 8-byte accesses), two rows in flight
// Narrow tensors are FOLDED by the host: f consecutive rows of Creal channels are read as one row of C = f * Creal columns.
__device__ __forceinline__ void hg_colsum2_body(const void* __restrict__ X, int dt, float* __restrict__ out, int64_t rows, int C, int rows_per_block, float scale, int Creal,
                                                int bx, int by) {
    __shared__ float sh[4][128];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r0 = (int64_t)by * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    if (dt == XVA_BF16 && C % 8 == 0 && ((uintptr_t)X % 16) == 0) {
        // 16-byte loads: a lane owns 8 columns, 16 lanes cover the 128-column tile, a wave instruction covers 4 rows; 4 rows x 4
        // instructions in flight per wave (4-byte loads, two in flight, measured 1.3 TB/s)
        const int c = bx * 128 + (lane & 15) * 8;
        float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c < C) {
            const uint16_t* xp = reinterpret_cast<const uint16_t*>(X) + c;
            auto add = [&](const uint4& q) {
                const uint32_t u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { a[2 * e] += __uint_as_float(u[e] << 16); a[2 * e + 1] += __uint_as_float(u[e] & 0xffff0000u); }
            };
            int64_t r = r0 + w * 4 + (lane >> 4);
            for (; r + 112 < r1; r += 128) {
                uint4 q[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) q[u] = *reinterpret_cast<const uint4*>(xp + (r + 16 * u) * C);
#pragma unroll
                for (int u = 0; u < 8; ++u) add(q[u]);
            }
            for (; r < r1; r += 16) add(*reinterpret_cast<const uint4*>(xp + r * C));
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {   // the 4 row-lanes of a column group
            a[e] += __shfl_xor(a[e], 16);
            a[e] += __shfl_xor(a[e], 32);
        }
        if (lane < 16) {
#pragma unroll
            for (int e = 0; e < 8; ++e) sh[w][lane * 8 + e] = a[e];
        }
    } else {
        const int c = bx * 128 + 2 * lane;
        float a0 = 0.f, a1 = 0.f;
        if (c < C) {
            if (dt == XVA_BF16) {
                const uint16_t* xp = reinterpret_cast<const uint16_t*>(X);
                int64_t r = r0 + w;
                for (; r + 4 < r1; r += 8) {
                    const uint32_t u = *reinterpret_cast<const uint32_t*>(xp + r * C + c), v = *reinterpret_cast<const uint32_t*>(xp + (r + 4) * C + c);
                    a0 += __uint_as_float(u << 16) + __uint_as_float(v << 16);
                    a1 += __uint_as_float(u & 0xffff0000u) + __uint_as_float(v & 0xffff0000u);
                }
                for (; r < r1; r += 4) { const uint32_t u = *reinterpret_cast<const uint32_t*>(xp + r * C + c); a0 += __uint_as_float(u << 16); a1 += __uint_as_float(u & 0xffff0000u); }
            } else {
                const float* xp = reinterpret_cast<const float*>(X);
                for (int64_t r = r0 + w; r < r1; r += 4) { const float2 f = *reinterpret_cast<const float2*>(xp + r * C + c); a0 += f.x; a1 += f.y; }
            }
        }
        sh[w][2 * lane] = a0; sh[w][2 * lane + 1] = a1;
    }
    __syncthreads();
    // folded narrow tensors (Creal < 128 | 128): the 128 / Creal replicas of a column are summed here, ONE atomic per real column and
    // workgroup (same-address atomics serialise in L2)
    const int wtile = C < 128 ? C : 128;
    if (Creal < wtile && wtile % Creal == 0) {
        if ((int)threadIdx.x < Creal) {
            float t = 0.f;
            for (int m = threadIdx.x; m < wtile; m += Creal) t += sh[0][m] + sh[1][m] + sh[2][m] + sh[3][m];
            atomicAdd(out + threadIdx.x, scale * t);
        }
    } else if (threadIdx.x < 128) {
        const int cc = bx * 128 + threadIdx.x;
        if (cc < C) atomicAdd(out + (cc % Creal), scale * (sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x]));
    }
}
__global__ void hg_colsum2_kernel(const void* __restrict__ X, int dt, float* __restrict__ out, int64_t rows, int C, int rows_per_block, float scale, int Creal) {
    hg_colsum2_body(X, dt, out, rows, C, rows_per_block, scale, Creal, blockIdx.x, blockIdx.y);
}
__global__ void hg_colsum2_batch_kernel(xva_cs_batch bt) {
    int l = 0;
    while (l + 1 < bt.n && (int)blockIdx.x >= bt.d[l + 1].block0) ++l;
    const xva_cs_desc& d = bt.d[l];
    const int local = (int)blockIdx.x - d.block0;
    hg_colsum2_body(d.X, d.dt, d.out, d.rows, d.C, d.rpb, d.scale, d.Creal, local % d.cb, local / d.cb);
}
extern "C" int xva_hg_colsum(const void* X, int dt, float* out, int64_t rows, int C, float scale, void* stream) {
    XVA_CHECK_ARG(X && out, "hg_colsum: null");
    if (rows <= 0) return XVA_OK;
    if (C % 2 == 0 && ((uintptr_t)X % 8) == 0) {
        const int Creal = C;
        while (C * 2 <= 128 && rows % 2 == 0) { C *= 2; rows /= 2; }   // fold rows of narrow tensors so every lane of a wave has a column pair
        // enough row blocks to fill the chip even for narrow tensors, few enough to keep the atomics per column low
        const int cb = xva_cdiv(C, 128);
        int rpb2 = (int)xva_cdiv(rows, xva_cdiv(256, cb) > 16 ? xva_cdiv(256, cb) : 16);
        if (rpb2 < 64) rpb2 = 64;
        hipLaunchKernelGGL(hg_colsum2_kernel, dim3(cb, (unsigned)xva_cdiv(rows, rpb2)), dim3(256), 0, (hipStream_t)stream, X, dt, out, rows, C, rpb2, scale, Creal);
        XVA_LAUNCH_CHECK();
        return XVA_OK;
    }
    const int rpb = 512;
    hipLaunchKernelGGL(hg_colsum_kernel, dim3(xva_cdiv(C, 64), xva_cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream, X, dt, out, rows, C, rpb, scale);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// several tensors, one launch (tensors that do not fit the paired-column form go through the single-tensor entry point)
extern "C" int xva_hg_colsum_batch(const xva_cs_desc* descs, int n, void* stream) {
    XVA_CHECK_ARG(descs || n == 0, "hg_colsum_batch: null");
    xva_cs_batch bt;
    bt.n = 0;
    int blocks = 0;
    auto flush = [&]() {
        if (bt.n > 0) hipLaunchKernelGGL(hg_colsum2_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, bt);
        bt.n = 0; blocks = 0;
    };
    for (int i = 0; i < n; ++i) {
        xva_cs_desc d = descs[i];
        XVA_CHECK_ARG(d.X && d.out, "hg_colsum_batch: null tensor %d", i);
        if (d.rows <= 0) continue;
        if (!(d.C % 2 == 0 && ((uintptr_t)d.X % 8) == 0)) { XVA_TRY(xva_hg_colsum(d.X, d.dt, d.out, d.rows, d.C, d.scale, stream)); continue; }
        d.Creal = d.C;
        while (d.C * 2 <= 128 && d.rows % 2 == 0) { d.C *= 2; d.rows /= 2; }
        d.cb = xva_cdiv(d.C, 128);
        int rpb = (int)xva_cdiv(d.rows, xva_cdiv(256, d.cb) > 16 ? xva_cdiv(256, d.cb) : 16);
        if (rpb < 64) rpb = 64;
        d.rpb = rpb;
        d.block0 = blocks;
        blocks += d.cb * (int)xva_cdiv(d.rows, rpb);
        bt.d[bt.n++] = d;
        if (bt.n == XVA_CS_BATCH) flush();
    }
    flush();
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// seq[b][padF + t][c] += vec[b][c] on the T valid rows of each item (pad rows stay zero): the VITS decoder's o + cond_layer(g)
// (python/xvapitch/hifigan.py:247-248).  One thread per 4 channels.
__global__ void hg_add_item_vec_kernel(void* __restrict__ seq, int dt, const float* __restrict__ vec, int Hp, int padF, int T, int C, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int c4 = C >> 2;
    const int64_t row = i / c4;                       // over B * T valid rows
    const int c = (int)(i % c4) * 4;
    const int b = (int)(row / T), t = (int)(row % T);
    const int64_t e = ((int64_t)b * Hp + padF + t) * C + c;
    const float4 v = *reinterpret_cast<const float4*>(vec + (int64_t)b * C + c);
    hg_st(seq, e, dt, hg_ld(seq, e, dt) + v.x); hg_st(seq, e + 1, dt, hg_ld(seq, e + 1, dt) + v.y);
    hg_st(seq, e + 2, dt, hg_ld(seq, e + 2, dt) + v.z); hg_st(seq, e + 3, dt, hg_ld(seq, e + 3, dt) + v.w);
}
extern "C" int xva_hg_add_item_vec(void* seq, int dt, const float* vec, int B, int Hp, int padF, int T, int C, void* stream) {
    XVA_CHECK_ARG(seq && vec && C % 4 == 0, "add_item_vec: null or C %% 4 != 0");
    const int64_t total4 = (int64_t)B * T * (C / 4);
    hipLaunchKernelGGL(hg_add_item_vec_kernel, dim3((unsigned)xva_cdiv(total4, 256)), dim3(256), 0, (hipStream_t)stream, seq, dt, vec, Hp, padF, T, C, total4);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// weight_norm (old API, dim 0): w = g * v / ||v||  (torch.nn.utils.weight_norm; models.py:21-108)
// v: (D0, inner) fp32 in the checkpoint layout; writes the effective weight in a GEMM layout given by an index map:
//   kind 0 (Conv, v = (Cout, Cin_g, k)):       eff[o][j*Cin_g + i]              = w[o][i][j]         (tap-major)
//   kind 2 (grouped Conv with few channels per group, v = (Cout, Cin_g, k)): the groups are taken S / Cin_g at a time as SUPER-GROUPS of S input
//           channels, each a dense product over a block-diagonal weight: eff[o][j*S + ((o / Cout_g) % (S / Cin_g))*Cin_g + i] = w[o][i][j], zero
//           elsewhere (s = S, pconv = Cout_g; S = Cin: one dense product over all groups)
//   kind 1 (ConvTranspose, v = (Cin, Cout, k)): effF[phase][co][m*Cin + ci]      = w[ci][co][j0(phase) + m*s]   (forward, per phase)
//                                               effB[ci][j*Cout + co]            = w[ci][co][j]                  (backward-data conv)
// One block per dim-0 index.  norm[o] saved for the backward.
__device__ __forceinline__ void hg_weight_norm_fwd_body(const float* __restrict__ v, const float* __restrict__ gparam, void* __restrict__ eff,
                                                        void* __restrict__ effB, float* __restrict__ norm, int dt, int kind, int D0, int D1, int k, int s,
                                                        int pconv, int o) {
    __shared__ float sh[16];
    const int inner = D1 * k;
    const float* vo = v + (int64_t)o * inner;
    float acc = 0.f;
    for (int i = threadIdx.x; i < inner; i += blockDim.x) acc += vo[i] * vo[i];
    acc = xva_block_sum(acc, sh);
    float n = sqrtf(acc);
    if (threadIdx.x == 0) norm[o] = n;
    float sc = gparam ? gparam[o] / n : 1.f;          // gparam == null: a plain (not reparametrised) weight, re-laid out only
    if (kind == 0 && dt == XVA_BF16 && (D1 & 7) == 0 && ((uintptr_t)eff & 15) == 0) {
        // Conv weights in WRITE order: a thread owns 8 consecutive input channels of one tap = one 16-byte store; its 8 reads walk the row with stride k (the row
        // was read for the norm a moment ago: L1 / L2).  The read-order loop below scatters 2-byte stores D1 elements apart: 0.8 TB/s on the 1024-channel layers.
        uint16_t* eo = reinterpret_cast<uint16_t*>(eff) + (int64_t)o * inner;
        for (int c = threadIdx.x; c < inner / 8; c += blockDim.x) {
            const int t = c * 8, j = t / D1, i1 = t - j * D1;
            uint32_t pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = hg_pack2(vo[(int64_t)(i1 + 2 * e) * k + j] * sc, vo[(int64_t)(i1 + 2 * e + 1) * k + j] * sc);
            *reinterpret_cast<uint4*>(eo + t) = make_uint4(pk[0], pk[1], pk[2], pk[3]);
        }
        return;
    }
    for (int idx = threadIdx.x; idx < inner; idx += blockDim.x) {
        int i1 = idx / k, j = idx % k;
        float w = vo[idx] * sc;
        if (kind == 0) {
            hg_st(eff, (int64_t)o * inner + (int64_t)j * D1 + i1, dt, w);
        } else if (kind == 2) {   // block-diagonal weight of a super-group of s input channels, pconv = Cout per group; the rest stays zero
            hg_st(eff, ((int64_t)o * k + j) * s + (int64_t)((o / pconv) % (s / D1)) * D1 + i1, dt, w);
        } else {
            // o = ci, i1 = co ; forward phases: t_out = s*q + phi uses taps j = j0 + m*s with j0 = (phi + pconv) % s
            int ntap = k / s;
            int Cin = D0, Cout = D1;
            int j0 = j % s, m = j / s;
            int phi = ((j0 - pconv) % s + s) % s;
            hg_st(eff, (((int64_t)phi * Cout + i1) * ntap + m) * Cin + o, dt, w);
            hg_st(effB, (int64_t)o * k * Cout + (int64_t)j * Cout + i1, dt, w);
        }
    }
}
__global__ void hg_weight_norm_fwd_kernel(const float* __restrict__ v, const float* __restrict__ gparam, void* __restrict__ eff,
                                          void* __restrict__ effB, float* __restrict__ norm, int dt, int kind, int D0, int D1, int k, int s,
                                          int pconv) {
    hg_weight_norm_fwd_body(v, gparam, eff, effB, norm, dt, kind, D0, D1, k, s, pconv, blockIdx.x);
}
// Batched form: up to XVA_WN_BATCH layers per launch, one workgroup per (layer, dim-0 index); the descriptors travel as kernel arguments.
__global__ void hg_weight_norm_fwd_batch_kernel(xva_wn_batch b) {
    int l = 0;
    while (l + 1 < b.n && (int)blockIdx.x >= b.d[l + 1].block0) ++l;
    const xva_wn_desc& d = b.d[l];
    hg_weight_norm_fwd_body(d.v, d.g, d.eff, d.effB, d.norm, d.dt, d.kind, d.D0, d.D1, d.k, d.s, d.pconv, (int)blockIdx.x - d.block0);
}
// dW: fp32 gradient of the effective weight in the layout: kind 0 tap-major [o][j*D1 + i] ; kind 1 [ci][j*Cout + co].
// dg[o] = sum dW * v / ||v|| ; dv = (g/||v||) * (dW - (dg/||v||) * v)
__device__ __forceinline__ void hg_weight_norm_bwd_body(const float* __restrict__ dW, const float* __restrict__ v, const float* __restrict__ gparam,
                                                        const float* __restrict__ norm, float* __restrict__ dv, float* __restrict__ dg, int kind, int D0,
                                                        int D1, int k, int o, int s = 0, int pconv = 1) {
    __shared__ float sh[16];
    const int inner = D1 * k;
    const float* vo = v + (int64_t)o * inner;
    // kind 2: dW is the gradient of the block-diagonal super-group weight; only this row's own block is read (tap pitch s)
    const int tp = kind == 2 ? s : D1;
    const float* dwo = kind == 2 ? dW + (int64_t)o * k * s + (int64_t)((o / pconv) % (s / D1)) * D1 : dW + (int64_t)o * inner;
    if (!gparam) {                                    // plain weight: dv += dW in the checkpoint layout
        for (int idx = threadIdx.x; idx < inner; idx += blockDim.x) dv[(int64_t)o * inner + idx] += dwo[(int64_t)(idx % k) * tp + idx / k];
        return;
    }
    float acc = 0.f;
    for (int idx = threadIdx.x; idx < inner; idx += blockDim.x) {
        int i1 = idx / k, j = idx % k;
        acc += dwo[(int64_t)j * tp + i1] * vo[idx];
    }
    acc = xva_block_sum(acc, sh);
    float n = norm[o], g = gparam[o];
    float dgo = acc / n;
    if (threadIdx.x == 0) dg[o] += dgo;
    for (int idx = threadIdx.x; idx < inner; idx += blockDim.x) {
        int i1 = idx / k, j = idx % k;
        dv[(int64_t)o * inner + idx] += (g / n) * (dwo[(int64_t)j * tp + i1] - (dgo / n) * vo[idx]);
    }
}
__global__ void hg_weight_norm_bwd_kernel(const float* __restrict__ dW, const float* __restrict__ v, const float* __restrict__ gparam,
                                          const float* __restrict__ norm, float* __restrict__ dv, float* __restrict__ dg, int kind, int D0,
                                          int D1, int k) {
    hg_weight_norm_bwd_body(dW, v, gparam, norm, dv, dg, kind, D0, D1, k, blockIdx.x);
}
__global__ void hg_weight_norm_bwd_batch_kernel(xva_wn_batch b) {
    int l = 0;
    while (l + 1 < b.n && (int)blockIdx.x >= b.d[l + 1].block0) ++l;
    const xva_wn_desc& d = b.d[l];
    hg_weight_norm_bwd_body(d.dW, d.v, d.g, d.norm, d.dv, d.dg, d.kind, d.D0, d.D1, d.k, (int)blockIdx.x - d.block0, d.s, d.pconv);
}
// launches: `n` layers in chunks of XVA_WN_BATCH
extern "C" int xva_hg_weight_norm_batch(const xva_wn_desc* descs, int n, int backward, void* stream) {
    XVA_CHECK_ARG(descs || n == 0, "weight_norm_batch: null");
    for (int i0 = 0; i0 < n; i0 += XVA_WN_BATCH) {
        xva_wn_batch b;
        b.n = n - i0 < XVA_WN_BATCH ? n - i0 : XVA_WN_BATCH;
        int blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            b.d[i] = descs[i0 + i];
            XVA_CHECK_ARG(b.d[i].v && b.d[i].norm && (b.d[i].g || b.d[i].kind != 1) &&
                          (backward ? (b.d[i].dW && b.d[i].dv && (b.d[i].dg || !b.d[i].g)) : (b.d[i].eff && (b.d[i].kind != 1 || b.d[i].effB))),
                          "weight_norm_batch: null tensor in layer %d", i0 + i);
            XVA_CHECK_ARG(backward || b.d[i].kind != 1 || (b.d[i].k % b.d[i].s == 0), "weight_norm_batch: transposed conv needs k %% s == 0");
            b.d[i].block0 = blocks;
            blocks += b.d[i].D0;
        }
        if (blocks == 0) continue;
        if (backward) hipLaunchKernelGGL(hg_weight_norm_bwd_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
        else hipLaunchKernelGGL(hg_weight_norm_fwd_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, b);
    }
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_weight_norm_fwd(const float* v, const float* g, void* eff, void* effB, float* norm, int dt, int kind, int D0, int D1, int k,
                                      int s, int pconv, void* stream) {
    XVA_CHECK_ARG(v && g && eff && norm && (kind == 0 || effB), "weight_norm_fwd: null");
    XVA_CHECK_ARG(kind == 0 || (k % s == 0), "weight_norm_fwd: transposed conv needs k %% s == 0");
    hipLaunchKernelGGL(hg_weight_norm_fwd_kernel, dim3(D0), dim3(256), 0, (hipStream_t)stream, v, g, eff, effB, norm, dt, kind, D0, D1, k, s, pconv);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_weight_norm_bwd(const float* dW, const float* v, const float* g, const float* norm, float* dv, float* dg, int kind, int D0,
                                      int D1, int k, void* stream) {
    XVA_CHECK_ARG(dW && v && g && norm && dv && dg, "weight_norm_bwd: null");
    hipLaunchKernelGGL(hg_weight_norm_bwd_kernel, dim3(D0), dim3(256), 0, (hipStream_t)stream, dW, v, g, norm, dv, dg, kind, D0, D1, k);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// spectral_norm (torch.nn.utils.spectral_norm, 1 power iteration per training forward, eps 1e-12; MSD discriminator 0,
// models.py:205,233).  W = weight_orig viewed (D0, inner = D1*k):
//   v <- normalize(W^T u) ; u <- normalize(W v) ; sigma = u . (W v) ; eff (tap-major) = W / sigma
// tmp: inner + D0 + 2 floats of scratch.
__global__ void sn_wt_u_kernel(const float* __restrict__ W, const float* __restrict__ u, float* __restrict__ t, int D0, int inner) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= inner) return;
    const int o0 = blockIdx.y * 64, o1 = min(o0 + 64, D0);
    float a = 0.f;
    for (int o = o0; o < o1; ++o) a += W[(int64_t)o * inner + i] * u[o];
    atomicAdd(t + i, a);
}
// out = in / max(||in||, eps); optionally dot_out = out . in  (single block)
__global__ void sn_normalize_kernel(const float* __restrict__ in, float* __restrict__ out, int n, float* __restrict__ dot_out) {
    __shared__ float sh[16];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += in[i] * in[i];
    a = xva_block_sum(a, sh);
    float inv = 1.f / fmaxf(sqrtf(a), 1e-12f);
    float d = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { float o = in[i] * inv; out[i] = o; d += o * in[i]; }
    if (dot_out) { d = xva_block_sum(d, sh); if (threadIdx.x == 0) dot_out[0] = d; }
}
__global__ void sn_w_v_kernel(const float* __restrict__ W, const float* __restrict__ v, float* __restrict__ sres, int D0, int inner) {
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int o = blockIdx.x * 4 + wave;
    if (o >= D0) return;
    float a = 0.f;
    for (int i = lane; i < inner; i += 64) a += W[(int64_t)o * inner + i] * v[i];
    a = xva_wave_sum(a);
    if (lane == 0) sres[o] = a;
}
__global__ void sn_scale_kernel(const float* __restrict__ W, const float* __restrict__ sigma, void* __restrict__ eff, int dt, int D0, int D1, int k) {
    const int inner = D1 * k;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)D0 * inner) return;
    int o = (int)(idx / inner), rem = (int)(idx % inner);
    int i1 = rem / k, j = rem % k;
    hg_st(eff, (int64_t)o * inner + (int64_t)j * D1 + i1, dt, W[idx] / sigma[0]);
}
// dW_orig += dW_eff / sigma - (sum dW_eff * W_orig) / sigma^2 * u v^T    (u, v = the buffers AFTER this pass's power iteration)
__global__ void sn_bwd_dot_kernel(const float* __restrict__ dWeff, const float* __restrict__ W, float* __restrict__ dot, int D0, int D1, int k) {
    __shared__ float sh[16];
    const int inner = D1 * k;
    float acc = 0.f;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)D0 * inner; idx += (int64_t)gridDim.x * blockDim.x) {
        int o = (int)(idx / inner), rem = (int)(idx % inner);
        int i1 = rem / k, j = rem % k;
        acc += dWeff[(int64_t)o * inner + (int64_t)j * D1 + i1] * W[idx];
    }
    acc = xva_block_sum(acc, sh);
    if (threadIdx.x == 0) atomicAdd(dot, acc);
}
__global__ void sn_bwd_apply_kernel(const float* __restrict__ dWeff, const float* __restrict__ u, const float* __restrict__ vv,
                                    const float* __restrict__ sigma, const float* __restrict__ dot, float* __restrict__ dWorig, int D0, int D1,
                                    int k) {
    const int inner = D1 * k;
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)D0 * inner) return;
    int o = (int)(idx / inner), rem = (int)(idx % inner);
    int i1 = rem / k, j = rem % k;
    float sg = sigma[0];
    dWorig[idx] += dWeff[(int64_t)o * inner + (int64_t)j * D1 + i1] / sg - (dot[0] / (sg * sg)) * u[o] * vv[rem];
}
extern "C" int xva_hg_spectral_norm_fwd(const float* W, float* u, float* v, void* eff, float* sigma, int dt, int D0, int D1, int k, float* tmp,
                                        void* stream) {
    XVA_CHECK_ARG(W && u && v && eff && sigma && tmp, "spectral_norm_fwd: null");
    hipStream_t st = (hipStream_t)stream;
    const int inner = D1 * k;
    if (hipMemsetAsync(tmp, 0, inner * sizeof(float), st) != hipSuccess) { xva_set_error("spectral_norm_fwd: memset failed"); return XVA_ERR_HIP; }
    hipLaunchKernelGGL(sn_wt_u_kernel, dim3(xva_cdiv(inner, 256), xva_cdiv(D0, 64)), dim3(256), 0, st, W, u, tmp, D0, inner);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, tmp, v, inner, (float*)nullptr);
    hipLaunchKernelGGL(sn_w_v_kernel, dim3(xva_cdiv(D0, 4)), dim3(256), 0, st, W, v, tmp + inner, D0, inner);
    hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, st, tmp + inner, u, D0, sigma);
    hipLaunchKernelGGL(sn_scale_kernel, dim3(xva_cdiv((int64_t)D0 * inner, 256)), dim3(256), 0, st, W, sigma, eff, dt, D0, D1, k);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
// ---- the same, every phase one launch over up to XVA_SN_BATCH layers (descriptors travel as kernel arguments) ----
template <int PHASE>
__device__ __forceinline__ int sn_layer_of(const xva_sn_batch& bt, int blk) {
    int l = 0;
    for (int i = 1; i < bt.n; ++i) {
        const int b0 = PHASE == 0 ? bt.d[i].b_wtu : (PHASE == 1 ? bt.d[i].b_wv : bt.d[i].b_scale);
        if (blk >= b0) l = i;
    }
    return l;
}
// phase 0: t[i] = sum_o W[o][i] u[o]: a thread per column i and 64-row chunk (coalesced across i), partial sums by atomics
__global__ void sn_b_wtu_kernel(xva_sn_batch bt) {
    const int l = sn_layer_of<0>(bt, blockIdx.x);
    const xva_sn_desc& d = bt.d[l];
    const int inner = d.D1 * d.k;
    const int cb = (inner + 255) / 256;                       // column blocks of this layer; the rest of its blocks split D0 in chunks of 64 rows
    const int local = blockIdx.x - d.b_wtu;
    const int i = (local % cb) * 256 + threadIdx.x;
    if (i >= inner) return;
    const int o0 = (local / cb) * 64, o1 = min(o0 + 64, d.D0);
    float a = 0.f;
    for (int o = o0; o < o1; ++o) a += d.W[(int64_t)o * inner + i] * d.u[o];
    atomicAdd(d.tmp + i, a);                                  // tmp[0 .. inner) is zeroed by the host before the launch
}
// phases 1 / 3: one workgroup per layer: out = in / max(||in||, eps) (+ saved copy, + sigma = out . in)
__global__ void sn_b_normalize_kernel(xva_sn_batch bt, int second) {
    __shared__ float sh[16];
    const xva_sn_desc& d = bt.d[blockIdx.x];
    const int inner = d.D1 * d.k;
    const float* in = second ? d.tmp + inner : d.tmp;
    float* out = second ? d.u : d.v;
    float* keep = second ? d.su : d.sv;
    const int n = second ? d.D0 : inner;
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) a += in[i] * in[i];
    a = xva_block_sum(a, sh);
    const float inv = 1.f / fmaxf(sqrtf(a), 1e-12f);
    float dt = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const float o = in[i] * inv; out[i] = o; if (keep) keep[i] = o; dt += o * in[i]; }
    if (second) { dt = xva_block_sum(dt, sh); if (threadIdx.x == 0) d.sigma[0] = dt; }
}
// phase 2: s[o] = W[o] . v, one wave per row
__global__ void sn_b_wv_kernel(xva_sn_batch bt) {
    const int l = sn_layer_of<1>(bt, blockIdx.x);
    const xva_sn_desc& d = bt.d[l];
    const int inner = d.D1 * d.k;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int o = (blockIdx.x - d.b_wv) * 4 + wave;
    if (o >= d.D0) return;
    float a = 0.f;
    for (int i = lane; i < inner; i += 64) a += d.W[(int64_t)o * inner + i] * d.v[i];
    a = xva_wave_sum(a);
    if (lane == 0) d.tmp[inner + o] = a;
}
// phase 4: eff (tap-major, dt) = W / sigma (+ fp32 copy)
__global__ void sn_b_scale_kernel(xva_sn_batch bt) {
    const int l = sn_layer_of<2>(bt, blockIdx.x);
    const xva_sn_desc& d = bt.d[l];
    const int inner = d.D1 * d.k;
    const int64_t idx = (int64_t)(blockIdx.x - d.b_scale) * 256 + threadIdx.x;
    if (idx >= (int64_t)d.D0 * inner) return;
    const int o = (int)(idx / inner), rem = (int)(idx % inner);
    const int i1 = rem / d.k, j = rem % d.k;
    const float w = d.W[idx] / d.sigma[0];
    const int64_t at = (int64_t)o * inner + (int64_t)j * d.D1 + i1;
    hg_st(d.eff, at, d.dt, w);
    if (d.eff2) reinterpret_cast<float*>(d.eff2)[at] = w;
}
extern "C" int xva_hg_spectral_norm_fwd_batch(xva_sn_desc* descs, int n, void* stream) {
    XVA_CHECK_ARG(descs && n > 0 && n <= XVA_SN_BATCH, "spectral_norm_fwd_batch: 1..%d layers", XVA_SN_BATCH);
    hipStream_t st = (hipStream_t)stream;
    xva_sn_batch bt;
    bt.n = n; bt.nb_wtu = bt.nb_wv = bt.nb_scale = 0;
    for (int i = 0; i < n; ++i) {
        xva_sn_desc d = descs[i];
        XVA_CHECK_ARG(d.W && d.u && d.v && d.eff && d.sigma && d.tmp, "spectral_norm_fwd_batch: null in layer %d", i);
        const int inner = d.D1 * d.k;
        d.b_wtu = bt.nb_wtu; bt.nb_wtu += xva_cdiv(inner, 256) * xva_cdiv(d.D0, 64);
        if (hipMemsetAsync(d.tmp, 0, inner * sizeof(float), st) != hipSuccess) { xva_set_error("spectral_norm_fwd_batch: memset failed"); return XVA_ERR_HIP; }
        d.b_wv = bt.nb_wv; bt.nb_wv += xva_cdiv(d.D0, 4);
        d.b_scale = bt.nb_scale; bt.nb_scale += xva_cdiv((int64_t)d.D0 * inner, 256);
        bt.d[i] = d;
    }
    hipLaunchKernelGGL(sn_b_wtu_kernel, dim3(bt.nb_wtu), dim3(256), 0, st, bt);
    hipLaunchKernelGGL(sn_b_normalize_kernel, dim3(n), dim3(1024), 0, st, bt, 0);
    hipLaunchKernelGGL(sn_b_wv_kernel, dim3(bt.nb_wv), dim3(256), 0, st, bt);
    hipLaunchKernelGGL(sn_b_normalize_kernel, dim3(n), dim3(1024), 0, st, bt, 1);
    hipLaunchKernelGGL(sn_b_scale_kernel, dim3(bt.nb_scale), dim3(256), 0, st, bt);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
// eff = W / sigma in another dtype (fp32 copy for the 1-channel direct kernels), no power iteration
extern "C" int xva_hg_sn_scale(const float* W, const float* sigma, void* eff, int dt, int D0, int D1, int k, void* stream) {
    XVA_CHECK_ARG(W && sigma && eff, "sn_scale: null");
    hipLaunchKernelGGL(sn_scale_kernel, dim3(xva_cdiv((int64_t)D0 * D1 * k, 256)), dim3(256), 0, (hipStream_t)stream, W, sigma, eff, dt, D0, D1, k);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
// tmp: 1 float of scratch (zeroed here)
extern "C" int xva_hg_spectral_norm_bwd(const float* dWeff, const float* W, const float* u, const float* v, const float* sigma, float* dWorig, int D0,
                                        int D1, int k, float* tmp, void* stream) {
    XVA_CHECK_ARG(dWeff && W && u && v && sigma && dWorig && tmp, "spectral_norm_bwd: null");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(tmp, 0, sizeof(float), st) != hipSuccess) { xva_set_error("spectral_norm_bwd: memset failed"); return XVA_ERR_HIP; }
    int64_t n = (int64_t)D0 * D1 * k;
    int grid = (int)((n + 255) / 256); if (grid > 512) grid = 512;
    hipLaunchKernelGGL(sn_bwd_dot_kernel, dim3(grid), dim3(256), 0, st, dWeff, W, tmp, D0, D1, k);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(xva_cdiv(n, 256)), dim3(256), 0, st, dWeff, u, v, sigma, tmp, dWorig, D0, D1, k);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Waveform-boundary convs as GEMMs: explicit im2col of the 1-channel input (k taps padded to kp columns) so that the
// first discriminator layer's forward / backward-weight / backward-data all run on the MFMA GEMM.
//   xcol[(b,w)][padF + h'][j] = x(b, w, s*h' + j - P)  (j < k), 0 for k <= j < kp
__global__ void hg_im2col1_kernel(const float* __restrict__ wav, void* __restrict__ xcol, int dt, Cin1Geom g, int kp) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)g.nb * g.p * g.Tout * kp;
    if (i >= total) return;
    int j = (int)(i % kp);
    int h = (int)((i / kp) % g.Tout);
    int seq = (int)(i / ((int64_t)kp * g.Tout));
    int b = seq / g.p, w = seq % g.p;
    float v = j < g.k ? cin1_sample(wav, g, b, w, g.s * h + j - g.P) : 0.f;
    hg_st(xcol, ((int64_t)seq * g.Hp + g.padF + h) * kp + j, dt, v);
}
// d_wav[b][i] (+)= sum over folded positions aliasing sample i, sum_j [ (h + P - j) % s == 0 ] dxcol[(b,w)][(h + P - j)/s][j]
__global__ void hg_col2im1_kernel(const void* __restrict__ dxcol, int dt, float* __restrict__ dwav, Cin1Geom g, int kp, int accumulate) {
    int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= (int64_t)g.nb * g.Tw) return;
    int b = (int)(gi / g.Tw), i = (int)(gi % g.Tw);
    float total = 0.f;
    for (int alias = 0; alias < 2; ++alias) {
        int ii = i;
        if (alias == 1) { ii = 2 * (g.Tw - 1) - i; if (ii < g.Tw || ii >= g.Hfold * g.p) break; }
        int h = ii / g.p, w = ii % g.p;
        int seq = b * g.p + w;
        for (int j = 0; j < g.k; ++j) {
            int num = h + g.P - j;
            if (num < 0 || num % g.s != 0) continue;
            int hp = num / g.s;
            if (hp >= g.Tout) continue;
            total += hg_ld(dxcol, ((int64_t)seq * g.Hp + g.padF + hp) * kp + j, dt);
        }
    }
    if (accumulate) dwav[gi] += total; else dwav[gi] = total;
}
// zero-padded copy of a (rows, k) fp32 matrix into (rows, kp) of dtype dt ; and the reverse accumulation (fp32 -> fp32)
__global__ void hg_pad_cols_kernel(const float* __restrict__ src, void* __restrict__ dst, int dt, int rows, int k, int kp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * kp) return;
    int r = i / kp, j = i % kp;
    hg_st(dst, i, dt, j < k ? src[r * k + j] : 0.f);
}
__global__ void hg_unpad_cols_add_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int k, int kp) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * k) return;
    int r = i / k, j = i % k;
    dst[i] += src[r * kp + j];
}
extern "C" int xva_hg_im2col1(const float* wav, void* xcol, int dt, int nb, int Tw, int p, int k, int s, int P, int kp, int Hp, int padF, void* stream) {
    Cin1Geom g;
    XVA_TRY(cin1_geom(&g, nb, Tw, p, k, s, P, 1, Hp, padF));
    XVA_CHECK_ARG(wav && xcol && kp >= k, "im2col1: bad args");
    int64_t total = (int64_t)nb * p * g.Tout * kp;
    hipLaunchKernelGGL(hg_im2col1_kernel, dim3(xva_cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, wav, xcol, dt, g, kp);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_col2im1(const void* dxcol, int dt, float* dwav, int nb, int Tw, int p, int k, int s, int P, int kp, int Hp, int padF,
                              int accumulate, void* stream) {
    Cin1Geom g;
    XVA_TRY(cin1_geom(&g, nb, Tw, p, k, s, P, 1, Hp, padF));
    XVA_CHECK_ARG(dxcol && dwav, "col2im1: null");
    hipLaunchKernelGGL(hg_col2im1_kernel, dim3(xva_cdiv((int64_t)nb * Tw, 256)), dim3(256), 0, (hipStream_t)stream, dxcol, dt, dwav, g, kp, accumulate);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_pad_cols(const float* src, void* dst, int dt, int rows, int k, int kp, void* stream) {
    hipLaunchKernelGGL(hg_pad_cols_kernel, dim3(xva_cdiv(rows * kp, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, dt, rows, k, kp);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
__global__ void hg_add_f32_kernel(float* __restrict__ dst, const float* __restrict__ src, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
extern "C" int xva_hg_add_f32(float* dst, const float* src, int64_t n, void* stream) {
    if (n <= 0) return XVA_OK;
    int g = (int)((n + 255) / 256); if (g > 2048) g = 2048;
    hipLaunchKernelGGL(hg_add_f32_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, dst, src, n);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_hg_unpad_cols_add(const float* src, float* dst, int rows, int k, int kp, void* stream) {
    hipLaunchKernelGGL(hg_unpad_cols_add_kernel, dim3(xva_cdiv(rows * k, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, rows, k, kp);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
