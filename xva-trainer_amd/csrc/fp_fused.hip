// fp_fused.hip — fused blocks of the FastPitch transformer layer (gfx950, bf16 throughput mode).
//
// xva_fp_onet_ln_fwd: the tail of MultiHeadAttn.forward (python/fastpitch1_1/fastpitch/transformer.py:132-147) as ONE kernel:
//     sum1 = x + dropout(AV Wo^T) ; y1 = LayerNorm(sum1) * rowmask
// Before (rounds 1 - 4): a 64 x 64-tile GEMM launch (K = 64: a prologue and an epilogue with nothing between them, ~14 us for the decoder's 27 584 rows) that
// stored sum1, then a LayerNorm launch (~12 us) that read it back.  Here a wave owns 16 rows x all 384 columns: 48 MFMAs (2 k-steps x 24 column tiles) from
// A fragments loaded straight from HBM and W fragments from a 48 KB LDS image filled once per workgroup; the row statistics come out of the accumulators
// (two half-wave shuffles), sum1 is stored once (LayerNorm backward needs it) and never re-read.  Same operation order as the two kernels it replaces:
// v = acc ; dropout ; + x ; round to bf16 (the stored sum1) ; statistics and normalisation of the ROUNDED values.
#include "xva_common.h"
#include "gemm_core.h"
#include "../../include/xva_hip.h"

namespace {
using xva_gemm_impl::bf16x8;
using xva_gemm_impl::f32x4;
using xva_gemm_impl::pack_bf2;
#define XVA_LDS __attribute__((address_space(3)))

constexpr int DM = 384, DH = 64, NJ = DM / 16, OW = 8;          // OW waves per workgroup

__global__ __launch_bounds__(64 * OW) void onet_ln_fwd_kernel(const uint16_t* __restrict__ AV, const uint16_t* __restrict__ W, const uint16_t* __restrict__ X,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta, uint16_t* __restrict__ SUM1,
                                                              uint16_t* __restrict__ Y1, float* __restrict__ mean, float* __restrict__ rstd, int64_t rows,
                                                              int mask_mode, const int* __restrict__ lens, int Tp, float eps, float p_drop, uint64_t seed,
                                                              uint32_t stream_id) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* wl = (XVA_LDS uint8_t*)smem_raw;                 // [384 n][64 k] bf16, 128-byte rows, 16-byte chunk c of row n at position c ^ (n & 7)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int idx = threadIdx.x; idx < DM * 8; idx += 64 * OW) {
        const int n = idx >> 3, c = idx & 7;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(W + (int64_t)n * DH + c * 8);
        *reinterpret_cast<XVA_LDS bf16x8*>(wl + n * 128 + ((c ^ (n & 7)) << 4)) = v;
    }
    XVA_LDS float* gl = (XVA_LDS float*)(wl + DM * 128);             // gamma | beta
    for (int c = threadIdx.x; c < 2 * DM; c += 64 * OW) gl[c] = c < DM ? gamma[c] : beta[c - DM];
    __syncthreads();
    const int r16 = lane & 15, g = lane >> 4;
    const uint32_t bo0 = r16 * 128 + ((g ^ (r16 & 7)) << 4), bo1 = bo0 ^ 64;          // B fragment offsets inside a 16-row tile of the image, k half 0 / 1
    const int64_t nblk = (rows + 15) / 16;
    for (int64_t blk = (int64_t)blockIdx.x * OW + wave; blk < nblk; blk += (int64_t)gridDim.x * OW) {
        const int64_t row = blk * 16 + r16;
        const bool in = row < rows;
        const int64_t rowc = in ? row : rows - 1;
        // A fragments: row r16 of the block, k = kh * 32 + g * 8 ... + 7
        const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(AV + rowc * DH + g * 8);
        const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(AV + rowc * DH + 32 + g * 8);
        uint2 xr[NJ];                                                  // the residual row pieces: all in flight before the products (one round trip, not six)
#pragma unroll
        for (int j = 0; j < NJ; ++j) xr[j] = *reinterpret_cast<const uint2*>(X + rowc * DM + j * 16 + g * 4);
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bf16x8 b0 = *reinterpret_cast<const XVA_LDS bf16x8*>(wl + j * 16 * 128 + bo0);
            const bf16x8 b1 = *reinterpret_cast<const XVA_LDS bf16x8*>(wl + j * 16 * 128 + bo1);
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0, a0, z, 0, 0, 0);           // operands swapped as in gemm_glds.h: a lane ends with 4 consecutive
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1, a1, z, 0, 0, 0);      // columns (j * 16 + g * 4 ...) of row r16
        }
        // epilogue: dropout, + x, round to bf16 (= the stored sum1), statistics of the rounded values
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = j * 16 + g * 4;
            float v[4] = {acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
            if (p_drop > 0.f) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= xva_dropout_scale(p_drop, seed, stream_id, (uint64_t)row * DM + col + e);
            }
            v[0] += __uint_as_float(xr[j].x << 16); v[1] += __uint_as_float(xr[j].x & 0xffff0000u);
            v[2] += __uint_as_float(xr[j].y << 16); v[3] += __uint_as_float(xr[j].y & 0xffff0000u);
            const uint2 pk = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            if (in) *reinterpret_cast<uint2*>(SUM1 + row * DM + col) = pk;
            acc[j][0] = __uint_as_float(pk.x << 16); acc[j][1] = __uint_as_float(pk.x & 0xffff0000u);
            acc[j][2] = __uint_as_float(pk.y << 16); acc[j][3] = __uint_as_float(pk.y & 0xffff0000u);
            s += (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
            if ((j & 3) == 3) asm volatile("" ::: "memory");       // keep the compiler from forming all 96 dropout hashes at once (spills)
        }
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);                        // the four lanes that share row r16
        const float mu = s * (1.f / DM);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc[j][e] - mu; q += d * d; }
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        const float rs = rsqrtf(q * (1.f / DM) + eps);
        if (in && g == 0) { mean[row] = mu; rstd[row] = rs; }
        const bool live = in && xva_row_live(mask_mode, lens, Tp, row);
        if (in) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = j * 16 + g * 4;
                const f32x4 gmv = *reinterpret_cast<const XVA_LDS f32x4*>(gl + col), btv = *reinterpret_cast<const XVA_LDS f32x4*>(gl + DM + col);
                const float4 gm = make_float4(gmv[0], gmv[1], gmv[2], gmv[3]), bt = make_float4(btv[0], btv[1], btv[2], btv[3]);
                float y[4];
                y[0] = live ? (acc[j][0] - mu) * rs * gm.x + bt.x : 0.f; y[1] = live ? (acc[j][1] - mu) * rs * gm.y + bt.y : 0.f;
                y[2] = live ? (acc[j][2] - mu) * rs * gm.z + bt.z : 0.f; y[3] = live ? (acc[j][3] - mu) * rs * gm.w + bt.w : 0.f;
                *reinterpret_cast<uint2*>(Y1 + row * DM + col) = make_uint2(pack_bf2(y[0], y[1]), pack_bf2(y[2], y[3]));
            }
        }
    }
}
// ---- the fp16-operand mode's flavour (round 6) -----------------------------------------------------------------------------------------------------
// Same block on the fp32 residual stream: AV and Wo are IEEE-half operands (v_mfma_f32_16x16x32_f16), x / sum1 / y1 are fp32 (nothing is rounded before the
// statistics), and y1 is also stored as the half copy conv1 reads (what xva_fp_layernorm_fwd_pair writes at plane distance 0).  Replaces a K = 64 GEMM whose
// epilogue read 42 MB of x and wrote 42 MB of sum1 (37 us on the decoder's 27 584 rows) and the LayerNorm launch that read the sum back (16 us).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_h2(float a, float b) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(uint32_t, v);
}
__global__ __launch_bounds__(64 * OW) void onet_ln_fwd_f16_kernel(const uint16_t* __restrict__ AV, const uint16_t* __restrict__ W, const float* __restrict__ X,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ SUM1,
                                                                  float* __restrict__ Y1, uint16_t* __restrict__ YH, float* __restrict__ mean, float* __restrict__ rstd,
                                                                  int64_t rows, int mask_mode, const int* __restrict__ lens, int Tp, float eps, float p_drop, uint64_t seed,
                                                                  uint32_t stream_id) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* wl = (XVA_LDS uint8_t*)smem_raw;                 // the image of onet_ln_fwd_kernel
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int idx = threadIdx.x; idx < DM * 8; idx += 64 * OW) {
        const int n = idx >> 3, c = idx & 7;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(W + (int64_t)n * DH + c * 8);
        *reinterpret_cast<XVA_LDS bf16x8*>(wl + n * 128 + ((c ^ (n & 7)) << 4)) = v;
    }
    XVA_LDS float* gl = (XVA_LDS float*)(wl + DM * 128);
    for (int c = threadIdx.x; c < 2 * DM; c += 64 * OW) gl[c] = c < DM ? gamma[c] : beta[c - DM];
    __syncthreads();
    const int r16 = lane & 15, g = lane >> 4;
    const uint32_t bo0 = r16 * 128 + ((g ^ (r16 & 7)) << 4), bo1 = bo0 ^ 64;
    const int64_t nblk = (rows + 15) / 16;
    for (int64_t blk = (int64_t)blockIdx.x * OW + wave; blk < nblk; blk += (int64_t)gridDim.x * OW) {
        const int64_t row = blk * 16 + r16;
        const bool in = row < rows;
        const int64_t rowc = in ? row : rows - 1;
        const f16x8 a0 = *reinterpret_cast<const f16x8*>(AV + rowc * DH + g * 8);
        const f16x8 a1 = *reinterpret_cast<const f16x8*>(AV + rowc * DH + 32 + g * 8);
        f32x4 acc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[j] = *reinterpret_cast<const f32x4*>(X + rowc * DM + j * 16 + g * 4);      // the residual row pieces ride in the accumulators
        if (p_drop > 0.f) {          // dropout scales the product alone: products into zeroed accumulators, then + x
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f16x8 b0 = *reinterpret_cast<const XVA_LDS f16x8*>(wl + j * 16 * 128 + bo0);
                const f16x8 b1 = *reinterpret_cast<const XVA_LDS f16x8*>(wl + j * 16 * 128 + bo1);
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, a0, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, a1, z, 0, 0, 0);
                const int col = j * 16 + g * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] += z[e] * xva_dropout_scale(p_drop, seed, stream_id, (uint64_t)row * DM + col + e);
                if ((j & 3) == 3) asm volatile("" ::: "memory");
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const f16x8 b0 = *reinterpret_cast<const XVA_LDS f16x8*>(wl + j * 16 * 128 + bo0);
                const f16x8 b1 = *reinterpret_cast<const XVA_LDS f16x8*>(wl + j * 16 * 128 + bo1);
                f32x4 z = {0.f, 0.f, 0.f, 0.f};
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(b0, a0, z, 0, 0, 0);
                z = __builtin_amdgcn_mfma_f32_16x16x32_f16(b1, a1, z, 0, 0, 0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[j][e] += z[e];
            }
        }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (in) *reinterpret_cast<f32x4*>(SUM1 + row * DM + j * 16 + g * 4) = acc[j];
            s += (acc[j][0] + acc[j][1]) + (acc[j][2] + acc[j][3]);
        }
        s += __shfl_xor(s, 16, 64); s += __shfl_xor(s, 32, 64);
        const float mu = s * (1.f / DM);
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float d = acc[j][e] - mu; q += d * d; }
        q += __shfl_xor(q, 16, 64); q += __shfl_xor(q, 32, 64);
        const float rs = rsqrtf(q * (1.f / DM) + eps);
        if (in && g == 0) { mean[row] = mu; rstd[row] = rs; }
        const bool live = in && xva_row_live(mask_mode, lens, Tp, row);
        if (in) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = j * 16 + g * 4;
                const f32x4 gm = *reinterpret_cast<const XVA_LDS f32x4*>(gl + col), bt = *reinterpret_cast<const XVA_LDS f32x4*>(gl + DM + col);
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = live ? (acc[j][e] - mu) * rs * gm[e] + bt[e] : 0.f;
                *reinterpret_cast<f32x4*>(Y1 + row * DM + col) = y;
                *reinterpret_cast<uint2*>(YH + row * DM + col) = make_uint2(pack_h2(y[0], y[1]), pack_h2(y[2], y[3]));
            }
        }
    }
}
}  // namespace

extern "C" int xva_fp_onet_ln_fwd(const void* av, const void* w_bf16, const void* x, const float* gamma, const float* beta, void* sum1, void* y1, float* mean,
                                  float* rstd, int64_t rows, int mask_mode, const int32_t* lens, int Tp, float p_drop, uint64_t seed, uint32_t stream_id,
                                  void* stream) {
    XVA_CHECK_ARG(av && w_bf16 && x && gamma && beta && sum1 && y1 && mean && rstd && rows > 0, "onet_ln_fwd: null");
    auto al = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
    XVA_CHECK_ARG(al(av) && al(w_bf16) && al(x) && al(sum1) && al(y1) && al(gamma) && al(beta), "onet_ln_fwd: 16-byte alignment");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(onet_ln_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DM * 128 + 2 * DM * 4) != hipSuccess) {
            xva_set_error("onet_ln_fwd: cannot raise the dynamic LDS limit"); return XVA_ERR_HIP;
        }
        attr_set = true;
    }
    const int64_t nblk = (rows + 15) / 16;
    int grid = (int)((nblk + OW - 1) / OW); if (grid > 512) grid = 512;
    hipLaunchKernelGGL(onet_ln_fwd_kernel, dim3(grid), dim3(64 * OW), DM * 128 + 2 * DM * 4, (hipStream_t)stream, reinterpret_cast<const uint16_t*>(av),
                       reinterpret_cast<const uint16_t*>(w_bf16), reinterpret_cast<const uint16_t*>(x), gamma, beta, reinterpret_cast<uint16_t*>(sum1),
                       reinterpret_cast<uint16_t*>(y1), mean, rstd, rows, mask_mode, lens, Tp, 1e-5f, p_drop, seed, stream_id);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

extern "C" int xva_fp_onet_ln_fwd_f16(const void* av_f16, const void* w_f16, const float* x, const float* gamma, const float* beta, float* sum1, float* y1, void* y1_f16,
                                      float* mean, float* rstd, int64_t rows, int mask_mode, const int32_t* lens, int Tp, float p_drop, uint64_t seed, uint32_t stream_id,
                                      void* stream) {
    XVA_CHECK_ARG(av_f16 && w_f16 && x && gamma && beta && sum1 && y1 && y1_f16 && mean && rstd && rows > 0, "onet_ln_fwd_f16: null");
    auto al = [](const void* p) { return ((uintptr_t)p % 16) == 0; };
    XVA_CHECK_ARG(al(av_f16) && al(w_f16) && al(x) && al(sum1) && al(y1) && al(y1_f16) && al(gamma) && al(beta), "onet_ln_fwd_f16: 16-byte alignment");
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(onet_ln_fwd_f16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DM * 128 + 2 * DM * 4) != hipSuccess) {
            xva_set_error("onet_ln_fwd_f16: cannot raise the dynamic LDS limit"); return XVA_ERR_HIP;
        }
        attr_set = true;
    }
    const int64_t nblk = (rows + 15) / 16;
    int grid = (int)((nblk + OW - 1) / OW); if (grid > 512) grid = 512;
    hipLaunchKernelGGL(onet_ln_fwd_f16_kernel, dim3(grid), dim3(64 * OW), DM * 128 + 2 * DM * 4, (hipStream_t)stream, reinterpret_cast<const uint16_t*>(av_f16),
                       reinterpret_cast<const uint16_t*>(w_f16), x, gamma, beta, sum1, y1, reinterpret_cast<uint16_t*>(y1_f16), mean, rstd, rows, mask_mode, lens, Tp, 1e-5f,
                       p_drop, seed, stream_id);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
