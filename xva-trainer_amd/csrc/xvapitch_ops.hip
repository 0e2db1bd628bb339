// xvapitch_ops.hip — kernels of the xVAPitch-only blocks, first set: the WaveNet gate, the residual / skip
// split, monotonic alignment search, segment gather / scatter and the KL term.
//
// Reference:
//   fused_add_tanh_sigmoid_multiply, WN.forward       python/xvapitch/wavenet.py:5-12,92-109
//   maximum_path                                      python/xvapitch/util.py:14-53   (numpy on the CPU, one D2H + H2D round trip per step)
//   rand_segments / segment                           python/xvapitch/util.py:145-178
//   VitsGeneratorLoss.kl_loss                         python/xvapitch/losses.py:87-104
//
// Sequence tensors follow the HiFi-GAN engine's convention: time-major (B, Tp = pad + T + pad, C), rows = time, channels
// contiguous, pad rows structurally zero (they ARE the zero padding of the convolutions, which run on xva_gemm).  Everything
// here is HBM-bound: one coalesced pass, 16-byte vectors where the element type allows.
#include "xva_common.h"
#include "../../include/xva_hip.h"
#include <math.h>

namespace {
__device__ __forceinline__ float ld(const void* p, int64_t i, int dt) {
    return dt == XVA_BF16 ? __uint_as_float((uint32_t)reinterpret_cast<const uint16_t*>(p)[i] << 16) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st(void* p, int64_t i, int dt, float v) {
    if (dt == XVA_BF16) {
        uint32_t u = __float_as_uint(v);
        u += 0x7fffu + ((u >> 16) & 1u);                       // round to nearest even
        reinterpret_cast<uint16_t*>(p)[i] = (uint16_t)(u >> 16);
    } else reinterpret_cast<float*>(p)[i] = v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }
}  // namespace

// ---- WaveNet gate -----------------------------------------------------------------------------------------------------------------
// acts[r][c] = tanh(a[r][c] + g[b][c]) * sigmoid(a[r][H + c] + g[b][H + c])      (wavenet.py:5-12)
// a: (rows, 2H) the dilated conv's output; g: per-item conditioning (B, 2H; row stride g_ld) broadcast over time (the reference's cond_layer(g) has a
// singleton time axis) or NULL; pad rows (outside [pad, pad + len)) produce 0.
__global__ void wn_gate_fwd_kernel(const void* __restrict__ a, const float* __restrict__ g, void* __restrict__ acts, int dt, int64_t rows, int H,
                                   int Tp, int64_t g_ld) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H) return;
    const int64_t r = idx / H;
    const int c = (int)(idx - r * H);
    const int b = (int)(r / Tp);
    float ta = ld(a, r * 2 * H + c, dt), sa = ld(a, r * 2 * H + H + c, dt);
    if (g) { ta += g[(int64_t)b * g_ld + c]; sa += g[(int64_t)b * g_ld + H + c]; }
    st(acts, idx, dt, tanhf(ta) * sigmoidf_(sa));
}
// d_a[r][c] = d * s * (1 - t^2) ; d_a[r][H + c] = d * t * s * (1 - s) ; d_g[b][.] += the same summed over the item's rows (optional)
__global__ void wn_gate_bwd_kernel(const void* __restrict__ a, const float* __restrict__ g, const void* __restrict__ d_acts, void* __restrict__ d_a,
                                   int dt, int64_t rows, int H, int Tp, int64_t g_ld) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H) return;
    const int64_t r = idx / H;
    const int c = (int)(idx - r * H);
    const int b = (int)(r / Tp);
    float ta = ld(a, r * 2 * H + c, dt), sa = ld(a, r * 2 * H + H + c, dt);
    if (g) { ta += g[(int64_t)b * g_ld + c]; sa += g[(int64_t)b * g_ld + H + c]; }
    const float t = tanhf(ta), s = sigmoidf_(sa), d = ld(d_acts, idx, dt);
    st(d_a, r * 2 * H + c, dt, d * s * (1.f - t * t));
    st(d_a, r * 2 * H + H + c, dt, d * t * s * (1.f - s));
}
extern "C" int xva_wn_gate_fwd(const void* a, const float* g, int64_t g_ld, void* acts, int dt, int B, int Tp, int H, void* stream) {
    XVA_CHECK_ARG(a && acts && B > 0 && Tp > 0 && H > 0, "wn_gate_fwd: bad arguments");
    const int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(wn_gate_fwd_kernel, dim3((unsigned)xva_cdiv(rows * H, 256)), dim3(256), 0, (hipStream_t)stream, a, g, acts, dt, rows, H, Tp, g_ld);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_wn_gate_bwd(const void* a, const float* g, int64_t g_ld, const void* d_acts, void* d_a, int dt, int B, int Tp, int H, void* stream) {
    XVA_CHECK_ARG(a && d_acts && d_a && B > 0 && Tp > 0 && H > 0, "wn_gate_bwd: bad arguments");
    const int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(wn_gate_bwd_kernel, dim3((unsigned)xva_cdiv(rows * H, 256)), dim3(256), 0, (hipStream_t)stream, a, g, d_acts, d_a, dt, rows, H, Tp, g_ld);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- residual / skip split (wavenet.py:103-108) -------------------------------------------------------------------------------------
// rs: (rows, 2H) (last layer: (rows, H)).  x_next = (x + rs[:, :H]) * mask (a new tensor: the layer's input is kept for its backward) ;
// out += rs[:, H:]   (last: out += rs).  mask = live rows.
__global__ void wn_res_skip_fwd_kernel(const void* __restrict__ rs, const void* __restrict__ x, void* __restrict__ x_next, void* __restrict__ out, int dt,
                                       int64_t rows, int H, int last, const int32_t* __restrict__ lens, int Tp, int pad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H) return;
    const int64_t r = idx / H;
    const int c = (int)(idx - r * H);
    const int b = (int)(r / Tp), t = (int)(r - (int64_t)b * Tp) - pad;
    const bool live = t >= 0 && t < lens[b];
    if (last) { st(out, idx, dt, ld(out, idx, dt) + (live ? ld(rs, r * H + c, dt) : 0.f)); return; }
    st(x_next, idx, dt, live ? ld(x, idx, dt) + ld(rs, r * 2 * H + c, dt) : 0.f);
    st(out, idx, dt, ld(out, idx, dt) + (live ? ld(rs, r * 2 * H + H + c, dt) : 0.f));
}
// d_rs[:, :H] = d_x * mask ; d_rs[:, H:] = d_out   (last: d_rs = d_out) ; d_x (the residual path) stays d_x * mask and receives the
// in_layer's backward-data on top (done by the caller's GEMM with beta = 1).
__global__ void wn_res_skip_bwd_kernel(const void* __restrict__ d_x, const void* __restrict__ d_out, void* __restrict__ d_rs, int dt, int64_t rows, int H,
                                       int last, const int32_t* __restrict__ lens, int Tp, int pad) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * H) return;
    const int64_t r = idx / H;
    const int c = (int)(idx - r * H);
    const int b = (int)(r / Tp), t = (int)(r - (int64_t)b * Tp) - pad;
    const bool live = t >= 0 && t < lens[b];
    const float dout = live ? ld(d_out, idx, dt) : 0.f;
    if (last) { st(d_rs, r * H + c, dt, dout); return; }
    st(d_rs, r * 2 * H + c, dt, live ? ld(d_x, idx, dt) : 0.f);
    st(d_rs, r * 2 * H + H + c, dt, dout);
}
extern "C" int xva_wn_res_skip_fwd(const void* rs, const void* x, void* x_next, void* out, int dt, int B, int Tp, int pad, int H, int last, const int32_t* lens,
                                   void* stream) {
    XVA_CHECK_ARG(rs && out && lens && (last || (x && x_next)), "wn_res_skip_fwd: null");
    const int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(wn_res_skip_fwd_kernel, dim3((unsigned)xva_cdiv(rows * H, 256)), dim3(256), 0, (hipStream_t)stream, rs, x, x_next, out, dt, rows, H, last,
                       lens, Tp, pad);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_wn_res_skip_bwd(const void* d_x, const void* d_out, void* d_rs, int dt, int B, int Tp, int pad, int H, int last, const int32_t* lens,
                                   void* stream) {
    XVA_CHECK_ARG(d_out && d_rs && lens && (last || d_x), "wn_res_skip_bwd: null");
    const int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(wn_res_skip_bwd_kernel, dim3((unsigned)xva_cdiv(rows * H, 256)), dim3(256), 0, (hipStream_t)stream, d_x, d_out, d_rs, dt, rows, H, last,
                       lens, Tp, pad);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- monotonic alignment search (util.py:14-53) --------------------------------------------------------------------------------------
// value (B, t_x, t_y) fp32, x_lens / y_lens (B): path (B, t_x, t_y) fp32 of 0 / 1 — the reference's numpy loop, all on the device
// (no D2H / H2D round trip).  One workgroup per item, one thread per text position x; per mel frame j:
//   v0[x] = v[x - 1] (-inf at x = 0), direction[x][j] = v[x] >= v0[x], v[x] <- (x <= j) ? max(v[x], v0[x]) + value[x][j] * mask : -inf
// then one thread walks back from (x_len - 1, y_len - 1).  `dirs`: B * t_x * t_y bytes of scratch.  Ties resolve like the reference (>=).
__global__ void maximum_path_kernel(const float* __restrict__ value, const int32_t* __restrict__ x_lens, const int32_t* __restrict__ y_lens,
                                    float* __restrict__ path, uint8_t* __restrict__ dirs, int t_x, int t_y) {
    extern __shared__ float v_sh[];                              // 2 * t_x floats (double buffer)
    const int b = blockIdx.x;
    const int xl = x_lens[b], yl = y_lens[b];
    const float* val = value + (int64_t)b * t_x * t_y;
    uint8_t* dir = dirs + (int64_t)b * t_x * t_y;
    float* pth = path + (int64_t)b * t_x * t_y;
    for (int64_t i = threadIdx.x; i < (int64_t)t_x * t_y; i += blockDim.x) pth[i] = 0.f;
    for (int x = threadIdx.x; x < t_x; x += blockDim.x) v_sh[x] = 0.f;
    __syncthreads();
    int cur = 0;
    for (int j = 0; j < t_y; ++j) {
        float* vin = v_sh + cur * t_x;
        float* vout = v_sh + (cur ^ 1) * t_x;
        for (int x = threadIdx.x; x < t_x; x += blockDim.x) {
            const float v1 = vin[x], v0 = x > 0 ? vin[x - 1] : -INFINITY;
            const bool mm = v1 >= v0;
            const bool m = x < xl && j < yl;
            dir[(int64_t)x * t_y + j] = m ? (mm ? 1 : 0) : 1;     // direction = where(mask, direction, 1)
            const float vm = mm ? v1 : v0;
            vout[x] = x <= j ? vm + (m ? val[(int64_t)x * t_y + j] : 0.f) : -INFINITY;
        }
        cur ^= 1;
        __syncthreads();
    }
    if (threadIdx.x == 0 && xl > 0 && yl > 0) {
        int index = xl - 1;
        for (int j = t_y - 1; j >= 0; --j) {
            if (index >= 0 && index < t_x) {
                if (j < yl) pth[(int64_t)index * t_y + j] = 1.f;                           // path * mask
                index = index + (int)dir[(int64_t)index * t_y + j] - 1;
            }
        }
    }
}
extern "C" int64_t xva_maximum_path_workspace_bytes(int B, int t_x, int t_y) { return (int64_t)B * t_x * t_y; }
extern "C" int xva_maximum_path(const float* value, const int32_t* x_lens, const int32_t* y_lens, float* path, void* workspace, int64_t workspace_bytes,
                                int B, int t_x, int t_y, void* stream) {
    XVA_CHECK_ARG(value && x_lens && y_lens && path && workspace && B > 0 && t_x > 0 && t_y > 0, "maximum_path: bad arguments");
    XVA_CHECK_ARG(workspace_bytes >= (int64_t)B * t_x * t_y, "maximum_path: workspace too small");
    XVA_CHECK_ARG(t_x <= 8192, "maximum_path: t_x %d > 8192", t_x);
    hipLaunchKernelGGL(maximum_path_kernel, dim3(B), dim3(256), 2 * t_x * sizeof(float), (hipStream_t)stream, value, x_lens, y_lens, path,
                       (uint8_t*)workspace, t_x, t_y);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- segment gather / scatter (util.py:145-178) --------------------------------------------------------------------------------------
// x (B, C, T) -> out (B, C, S): out[b, c, s] = x[b, c, idx[b] + s] (0 past T, like the zeros_like initialisation when a slice runs short)
__global__ void segment_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx, float* __restrict__ out, int C, int T, int S) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C * S) return;
    const int c = e / S, s = e - c * S;
    const int64_t t = idx[b] + s;
    out[((int64_t)b * C + c) * S + s] = (t >= 0 && t < T) ? x[((int64_t)b * C + c) * T + t] : 0.f;
}
// d_x (B, C, T) = 0 except d_x[b, c, idx[b] + s] = d_out[b, c, s]
__global__ void segment_bwd_kernel(const float* __restrict__ d_out, const int64_t* __restrict__ idx, float* __restrict__ d_x, int C, int T, int S) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= C * T) return;
    const int c = e / T, t = e - c * T;
    const int64_t s = t - idx[b];
    d_x[((int64_t)b * C + c) * T + t] = (s >= 0 && s < S) ? d_out[((int64_t)b * C + c) * S + s] : 0.f;
}
extern "C" int xva_segment_fwd(const float* x, const int64_t* idx, float* out, int B, int C, int T, int S, void* stream) {
    XVA_CHECK_ARG(x && idx && out && B > 0 && C > 0 && T > 0 && S > 0, "segment_fwd: bad arguments");
    hipLaunchKernelGGL(segment_fwd_kernel, dim3(xva_cdiv((long)C * S, 256), B), dim3(256), 0, (hipStream_t)stream, x, idx, out, C, T, S);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_segment_bwd(const float* d_out, const int64_t* idx, float* d_x, int B, int C, int T, int S, void* stream) {
    XVA_CHECK_ARG(d_out && idx && d_x && B > 0 && C > 0 && T > 0 && S > 0, "segment_bwd: bad arguments");
    hipLaunchKernelGGL(segment_bwd_kernel, dim3(xva_cdiv((long)C * T, 256), B), dim3(256), 0, (hipStream_t)stream, d_out, idx, d_x, C, T, S);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- KL term of VitsGeneratorLoss (losses.py:87-104) -----------------------------------------------------------------------------------
// kl = logs_p - logs_q - 0.5 + 0.5 (z_p - m_p)^2 exp(-2 logs_p) ; loss = sum(kl * mask) / sum(mask).  Tensors (B, H, T) fp32, mask (B, 1, T).
// acc[0] += sum(kl * mask), acc[1] += sum(mask) (acc zeroed by the caller); kl_sample_wise written when non-null.
__global__ void kl_fwd_kernel(const float* __restrict__ z_p, const float* __restrict__ logs_q, const float* __restrict__ m_p, const float* __restrict__ logs_p,
                              const float* __restrict__ mask, float* __restrict__ kl_out, float* __restrict__ acc, int H, int T, int64_t n) {
    __shared__ float sh[16];
    float a = 0.f, ms = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t bt = i / ((int64_t)H * T);
        const int t = (int)(i % T);
        const float m = mask[bt * T + t];
        const float d = z_p[i] - m_p[i];
        const float kl = (logs_p[i] - logs_q[i] - 0.5f + 0.5f * d * d * __expf(-2.f * logs_p[i])) * m;
        if (kl_out) kl_out[i] = kl;
        a += kl;
        if ((i / T) % H == 0) ms += m;                         // sum(z_mask): the mask is (B, 1, T) — counted once per (b, t), not per channel
    }
    a = xva_block_sum(a, sh);
    ms = xva_block_sum(ms, sh);
    if (threadIdx.x == 0) { atomicAdd(acc, a); atomicAdd(acc + 1, ms); }
}
// gradients of loss = acc[0] / acc[1] scaled by `gscale` (the upstream gradient x the loss weight)
__global__ void kl_bwd_kernel(const float* __restrict__ z_p, const float* __restrict__ m_p, const float* __restrict__ logs_p, const float* __restrict__ mask,
                              const float* __restrict__ acc, float gscale, float* __restrict__ d_z_p, float* __restrict__ d_logs_q, float* __restrict__ d_m_p,
                              float* __restrict__ d_logs_p, int H, int T, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t bt = i / ((int64_t)H * T);
    const int t = (int)(i % T);
    const float k = gscale * mask[bt * T + t] / acc[1];
    const float d = z_p[i] - m_p[i], e = __expf(-2.f * logs_p[i]);
    if (d_z_p) d_z_p[i] = k * d * e;
    if (d_m_p) d_m_p[i] = -k * d * e;
    if (d_logs_q) d_logs_q[i] = -k;
    if (d_logs_p) d_logs_p[i] = k * (1.f - d * d * e);
}
extern "C" int xva_kl_loss_fwd(const float* z_p, const float* logs_q, const float* m_p, const float* logs_p, const float* mask, float* kl_sample_wise,
                               float* acc2, int B, int H, int T, void* stream) {
    XVA_CHECK_ARG(z_p && logs_q && m_p && logs_p && mask && acc2, "kl_loss_fwd: null");
    const int64_t n = (int64_t)B * H * T;
    int grid = (int)((n + 255) / 256); if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(kl_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, z_p, logs_q, m_p, logs_p, mask, kl_sample_wise, acc2, H, T, n);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_kl_loss_bwd(const float* z_p, const float* m_p, const float* logs_p, const float* mask, const float* acc2, float gscale, float* d_z_p,
                               float* d_logs_q, float* d_m_p, float* d_logs_p, int B, int H, int T, void* stream) {
    XVA_CHECK_ARG(z_p && m_p && logs_p && mask && acc2, "kl_loss_bwd: null");
    const int64_t n = (int64_t)B * H * T;
    hipLaunchKernelGGL(kl_bwd_kernel, dim3((unsigned)xva_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, z_p, m_p, logs_p, mask, acc2, gscale, d_z_p, d_logs_q,
                       d_m_p, d_logs_p, H, T, n);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- layout changes between the reference's (B, C, T) tensors and the time-major sequences the convolutions run on -----------------------
// seq[b][pad + t][c] = x[b][c][t] * (t < lens[b]) ; pad rows are written as zeros too (the tensor needs no prior clearing)
__global__ void bct_to_seq_kernel(const float* __restrict__ x, void* __restrict__ seq, int dt, int C, int T, int Tp, int pad, const int32_t* __restrict__ lens) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32 - pad, c0 = blockIdx.y * 32;   // tile rows (time incl. pads) x channels
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int len = lens ? lens[b] : T;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        tile[r][tx] = (c < C && t >= 0 && t < T && t < len) ? x[((int64_t)b * C + c) * T + t] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int tp = blockIdx.x * 32 + r, c = c0 + tx;
        if (tp < Tp && c < C) st(seq, ((int64_t)b * Tp + tp) * C + c, dt, tile[tx][r]);
    }
}
__global__ void seq_to_bct_kernel(const void* __restrict__ seq, float* __restrict__ x, int dt, int C, int T, int Tp, int pad, int accumulate) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < T && c < C) ? ld(seq, ((int64_t)b * Tp + pad + t) * C + c, dt) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        if (c < C && t < T) { float* d = x + ((int64_t)b * C + c) * T + t; *d = accumulate ? *d + tile[tx][r] : tile[tx][r]; }
    }
}
extern "C" int xva_bct_to_seq(const float* x, void* seq, int dt, int B, int C, int T, int pad, const int32_t* lens, void* stream) {
    XVA_CHECK_ARG(x && seq && B > 0 && C > 0 && T > 0 && pad >= 0, "bct_to_seq: bad arguments");
    const int Tp = T + 2 * pad;
    hipLaunchKernelGGL(bct_to_seq_kernel, dim3(xva_cdiv(Tp, 32), xva_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, x, seq, dt, C, T, Tp, pad, lens);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_seq_to_bct(const void* seq, float* x, int dt, int B, int C, int T, int pad, int accumulate, void* stream) {
    XVA_CHECK_ARG(x && seq && B > 0 && C > 0 && T > 0 && pad >= 0, "seq_to_bct: bad arguments");
    hipLaunchKernelGGL(seq_to_bct_kernel, dim3(xva_cdiv(T, 32), xva_cdiv(C, 32), B), dim3(256), 0, (hipStream_t)stream, seq, x, dt, C, T, T + 2 * pad, pad, accumulate);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- sequence mask: zero every row outside [pad, pad + lens[b]) (the reference's `* x_mask` after a biased conv) -----------------------------
__global__ void seq_mask_kernel(void* __restrict__ x, int dt, int64_t rows, int C, int Tp, int pad, const int32_t* __restrict__ lens) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * C) return;
    const int64_t r = idx / C;
    const int b = (int)(r / Tp), t = (int)(r - (int64_t)b * Tp) - pad;
    if (!(t >= 0 && t < lens[b])) st(x, idx, dt, 0.f);
}
extern "C" int xva_seq_mask(void* x, int dt, int B, int Tp, int pad, int C, const int32_t* lens, void* stream) {
    XVA_CHECK_ARG(x && lens && B > 0 && Tp > 0 && C > 0, "seq_mask: bad arguments");
    const int64_t rows = (int64_t)B * Tp;
    hipLaunchKernelGGL(seq_mask_kernel, dim3((unsigned)xva_cdiv(rows * C, 256)), dim3(256), 0, (hipStream_t)stream, x, dt, rows, C, Tp, pad, lens);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- nn.Dropout on a contiguous tensor: y[i] = x[i] * (0 or 1 / (1 - p)), the decision a keyed hash of (seed, site, i) (xva_common.h) ------------
// The same call on a gradient is the backward (same mask).  In place allowed.  Sites of python/xvapitch: glow_tts.py:473,477 (the
// sub-layer outputs of RelativePositionTransformer), sdp.py:90 (DilatedDepthSeparableConv).
__global__ __launch_bounds__(256) void dropout_apply_kernel(const void* __restrict__ x, void* __restrict__ y, int dt, int64_t n, float p, uint64_t seed,
                                                            uint32_t site) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        st(y, i, dt, ld(x, i, dt) * xva_dropout_scale(p, seed, site, (uint64_t)i));
}
extern "C" int xva_dropout_apply(const void* x, void* y, int dt, int64_t n, float p, uint64_t seed, uint32_t site, void* stream) {
    XVA_CHECK_ARG(x && y && n >= 0 && p >= 0.f && p < 1.f, "dropout_apply: bad arguments");
    if (n == 0) return XVA_OK;
    int64_t grid = (n + 255) / 256; if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x, y, dt, n, p, seed, site);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- per-item column sums of a time-major sequence: out[b * out_stride + c] += sum over the item's Tp rows of x[b][.][c] --------------------------
// (the gradient of WN's conditioning term, python/xvapitch/wavenet.py:91-97: g_l is broadcast over time, so d g_l[b] = sum_t d(in_act)[b, t])
// One workgroup per (item, 64 columns, chunk of 64 rows): 4 row phases x 64 columns, LDS tree over the phases, one atomic per column and chunk (a
// single workgroup per (item, 64 columns) walked 104 dependent loads per thread: 32 us for 5 MB, once per WaveNet layer on the critical chain of the backward pass).
__global__ __launch_bounds__(256) void seq_item_colsum_kernel(const void* __restrict__ x, int dt, float* __restrict__ out, int Tp, int C, int64_t out_stride) {
    __shared__ float part[4][64];
    const int b = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const int t0 = blockIdx.z * 64, t1 = t0 + 64 < Tp ? t0 + 64 : Tp;
    float acc = 0.f;
    if (c < C)
        for (int t = t0 + ph; t < t1; t += 4) acc += ld(x, ((int64_t)b * Tp + t) * C + c, dt);
    part[ph][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ph == 0 && c < C) atomicAdd(out + (int64_t)b * out_stride + c, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}
extern "C" int xva_seq_item_colsum(const void* x, int dt, float* out, int B, int Tp, int C, int64_t out_stride, void* stream) {
    XVA_CHECK_ARG(x && out && B > 0 && Tp > 0 && C > 0, "seq_item_colsum: bad arguments");
    hipLaunchKernelGGL(seq_item_colsum_kernel, dim3((unsigned)xva_cdiv(C, 64), (unsigned)B, (unsigned)xva_cdiv(Tp, 64)), dim3(256), 0, (hipStream_t)stream, x, dt, out, Tp, C,
                       out_stride);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- mean-only affine coupling (python/xvapitch/model.py:1519-1535, mean_only=True: log_scale = 0, logdet = 0) ------------------------------
// forward: out[b][c][t] = stats[b][pad + t][c] + x1[b][c][t] * mask   (stats is the masked `post` conv output, time-major; x1 / out (B, Ch, T))
// reverse: out = (x1 - m) * mask.   backward of forward: d_x1 = d_out * mask, d_stats[b][pad + t][c] = d_out[b][c][t] * mask (pads zero).
__global__ void coupling_kernel(const void* __restrict__ stats, const float* __restrict__ x1, float* __restrict__ out, int dt, int Ch, int T, int Tp, int pad,
                                const int32_t* __restrict__ lens, int reverse) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, c = c0 + tx;
        tile[r][tx] = (t < T && c < Ch) ? ld(stats, ((int64_t)b * Tp + pad + t) * Ch + c, dt) : 0.f;
    }
    __syncthreads();
    const int len = lens[b];
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        if (c < Ch && t < T) {
            const int64_t i = ((int64_t)b * Ch + c) * T + t;
            const float m = tile[tx][r], live = t < len ? 1.f : 0.f;
            out[i] = reverse ? (x1[i] - m) * live : m + x1[i] * live;
        }
    }
}
extern "C" int xva_coupling_mean_only(const void* stats, const float* x1, float* out, int dt, int B, int Ch, int T, int pad, const int32_t* lens, int reverse,
                                      void* stream) {
    XVA_CHECK_ARG(stats && x1 && out && lens, "coupling: null");
    hipLaunchKernelGGL(coupling_kernel, dim3(xva_cdiv(T, 32), xva_cdiv(Ch, 32), B), dim3(256), 0, (hipStream_t)stream, stats, x1, out, dt, Ch, T, T + 2 * pad,
                       pad, lens, reverse);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// backward of the mean-only coupling: d_x1 (B, Ch, T) = d_out * mask ; d_stats[b][pad + t][c] = (reverse ? -1 : 1) * d_out[b][c][t] * mask (pads zero)
__global__ void coupling_bwd_kernel(const float* __restrict__ d_out, float* __restrict__ d_x1, void* __restrict__ d_stats, int dt, int Ch, int T, int Tp, int pad,
                                    const int32_t* __restrict__ lens, int reverse) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32 - pad, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int len = lens[b];
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, t = t0 + tx;
        float v = 0.f;
        if (c < Ch && t >= 0 && t < T) {
            const int64_t i = ((int64_t)b * Ch + c) * T + t;
            v = t < len ? d_out[i] : 0.f;
            d_x1[i] = v;
        }
        tile[r][tx] = reverse ? -v : v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int tp = blockIdx.x * 32 + r, c = c0 + tx;
        if (tp < Tp && c < Ch) st(d_stats, ((int64_t)b * Tp + tp) * Ch + c, dt, tile[tx][r]);
    }
}
extern "C" int xva_coupling_mean_only_bwd(const float* d_out, float* d_x1, void* d_stats, int dt, int B, int Ch, int T, int pad, const int32_t* lens, int reverse,
                                          void* stream) {
    XVA_CHECK_ARG(d_out && d_x1 && d_stats && lens, "coupling_bwd: null");
    const int Tp = T + 2 * pad;
    hipLaunchKernelGGL(coupling_bwd_kernel, dim3(xva_cdiv(Tp, 32), xva_cdiv(Ch, 32), B), dim3(256), 0, (hipStream_t)stream, d_out, d_x1, d_stats, dt, Ch, T, Tp, pad,
                       lens, reverse);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- posterior sampling (PosteriorEncoder.forward, python/xvapitch/model.py:1470-1475) ----------------------------------------------------
// stats: masked `proj` output, time-major (B, pad + T + pad, 2 * Co) = [mean | log_scale]; eps (B, Co, T) the N(0, 1) draw (torch.randn_like in
// the reference: supplied by the caller).  z = (mean + eps * exp(log_scale)) * mask ; mean / log_scale returned in (B, Co, T).
__global__ void posterior_sample_kernel(const void* __restrict__ stats, const float* __restrict__ eps, float* __restrict__ z, float* __restrict__ mean,
                                        float* __restrict__ logs, int dt, int Co, int T, int Tp, int pad, const int32_t* __restrict__ lens) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Co * T) return;
    const int c = e / T, t = e - c * T;
    const int64_t row = ((int64_t)b * Tp + pad + t) * 2 * Co;
    const float m = ld(stats, row + c, dt), s = ld(stats, row + Co + c, dt);
    const int64_t i = ((int64_t)b * Co + c) * T + t;
    mean[i] = m; logs[i] = s;
    z[i] = t < lens[b] ? m + eps[i] * __expf(s) : 0.f;
}
// d_stats[b][pad + t][c] = (d_mean + d_z) * mask' ; d_stats[..][Co + c] = d_logs + d_z * eps * exp(log_scale)   (d_z only on live positions; the
// direct d_mean / d_logs terms are masked too: stats = proj(x) * mask)
__global__ void posterior_sample_bwd_kernel(const void* __restrict__ stats, const float* __restrict__ eps, const float* __restrict__ d_z,
                                            const float* __restrict__ d_mean, const float* __restrict__ d_logs, void* __restrict__ d_stats, int dt, int Co, int T,
                                            int Tp, int pad, const int32_t* __restrict__ lens) {
    const int b = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= Co * T) return;
    const int t = e / Co, c = e - t * Co;                      // channel-fastest: coalesced stores into the time-major gradient
    const int64_t row = ((int64_t)b * Tp + pad + t) * 2 * Co;
    const int64_t i = ((int64_t)b * Co + c) * T + t;
    float gm = 0.f, gs = 0.f;
    if (t < lens[b]) {
        const float s = ld(stats, row + Co + c, dt);
        const float dz = d_z ? d_z[i] : 0.f;
        gm = dz + (d_mean ? d_mean[i] : 0.f);
        gs = dz * eps[i] * __expf(s) + (d_logs ? d_logs[i] : 0.f);
    }
    st(d_stats, row + c, dt, gm);
    st(d_stats, row + Co + c, dt, gs);
}
extern "C" int xva_posterior_sample(const void* stats, const float* eps, float* z, float* mean, float* logs, int dt, int B, int Co, int T, int pad,
                                    const int32_t* lens, void* stream) {
    XVA_CHECK_ARG(stats && eps && z && mean && logs && lens, "posterior_sample: null");
    hipLaunchKernelGGL(posterior_sample_kernel, dim3(xva_cdiv((long)Co * T, 256), B), dim3(256), 0, (hipStream_t)stream, stats, eps, z, mean, logs, dt, Co, T,
                       T + 2 * pad, pad, lens);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_posterior_sample_bwd(const void* stats, const float* eps, const float* d_z, const float* d_mean, const float* d_logs, void* d_stats, int dt,
                                        int B, int Co, int T, int pad, const int32_t* lens, void* stream) {
    XVA_CHECK_ARG(stats && eps && d_stats && lens, "posterior_sample_bwd: null");
    hipLaunchKernelGGL(posterior_sample_bwd_kernel, dim3(xva_cdiv((long)Co * T, 256), B), dim3(256), 0, (hipStream_t)stream, stats, eps, d_z, d_mean, d_logs,
                       d_stats, dt, Co, T, T + 2 * pad, pad, lens);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- relative-position multi-head self-attention ----------------------------------------------------------------------------------
// RelativePositionMultiHeadAttention.attention (python/xvapitch/glow_tts.py:173-214) with its relative-key / relative-value terms
// (:216-292) written as what they compute: for r = j - i in [-w, w]
//     scores[i][j] = (q_i . k_j + q_i . emb_k[r + w]) / sqrt(dk);   masked_fill(mask == 0, -1e4) for i or j >= len;   p = softmax_j
//     out_i = sum_j p[i][j] v_j + sum_{|r| <= w} p[i][i + r] emb_v[r + w]
// (the reference pads the embeddings to 2T - 1 relative positions with zeros and runs two pad / reshape tricks to shift between
// relative and absolute indexing; outside the window the padded embeddings are zero, so nothing else contributes).
// Text-encoder sizes (T <= a few hundred tokens, dk ~ 100, 2 heads): a latency-bound VALU problem, not MFMA work; one workgroup per
// (item, head, query row).  q / k / v: fp32 rows of an activation matrix (row stride ld, head h at columns h*dk ..), item b's token t in row
// b*Tp + pad + t.  P (B, H, T, T) is kept for the backward.  emb: (Hr, 2w + 1, dk), Hr = 1 (heads share) or H.
namespace {
constexpr int RA_THREADS = 128;
__device__ __forceinline__ float ra_block_sum(float v, float* sh) {
    v = xva_wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1];
}
__device__ __forceinline__ float ra_block_max(float v, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    return fmaxf(sh[0], sh[1]);
}
}  // namespace

__global__ __launch_bounds__(RA_THREADS) void relattn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ v,
                                                                 int64_t ld, const float* __restrict__ emb_k, const float* __restrict__ emb_v,
                                                                 const int32_t* __restrict__ lens, float* __restrict__ P, float* __restrict__ out,
                                                                 int64_t ld_out, int T, int H, int dk, int w, int Hr, int Tp, int pad, float drop_p,
                                                                 uint64_t drop_seed, uint32_t drop_stream) {
    extern __shared__ float sm[];                 // scores / probabilities of this row [T] + q_i [dk] + reduction scratch [2]
    float* s = sm; float* qi = sm + T; float* red = qi + dk;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens[b];
    const int64_t row0 = (int64_t)b * Tp + pad;
    const float scale = rsqrtf((float)dk);
    for (int d = threadIdx.x; d < dk; d += RA_THREADS) qi[d] = q[(row0 + i) * ld + h * dk + d];
    __syncthreads();
    const float* ek = emb_k + (int64_t)(Hr == 1 ? 0 : h) * (2 * w + 1) * dk;
    const float* ev = emb_v + (int64_t)(Hr == 1 ? 0 : h) * (2 * w + 1) * dk;
    float mx = -3.0e38f;
    for (int j = threadIdx.x; j < T; j += RA_THREADS) {
        const float* kj = k + (row0 + j) * ld + h * dk;
        float acc = 0.f;
        for (int d = 0; d < dk; ++d) acc += qi[d] * kj[d];
        const int r = j - i;
        if (r >= -w && r <= w) {
            const float* e = ek + (int64_t)(r + w) * dk;
            float a2 = 0.f;
            for (int d = 0; d < dk; ++d) a2 += qi[d] * e[d];
            acc += a2;
        }
        acc *= scale;
        if (i >= len || j >= len) acc = -1e4f;
        s[j] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = ra_block_max(mx, red);
    float sum = 0.f;
    for (int j = threadIdx.x; j < T; j += RA_THREADS) { const float e = __expf(s[j] - mx); s[j] = e; sum += e; }
    sum = ra_block_sum(sum, red);
    const float inv = 1.f / sum;
    float* Pi = P + (((int64_t)b * H + h) * T + i) * T;
    // P keeps the softmax itself (its backward needs it); the products below use dropout(P) (glow_tts.py:204): element (b, h, i, j) of the
    // (B, H, T, T) weights is dropout index ((b H + h) T + i) T + j of site `drop_stream`
    const uint64_t drow = (((uint64_t)b * H + h) * T + i) * T;
    for (int j = threadIdx.x; j < T; j += RA_THREADS) {
        const float p = s[j] * inv;
        Pi[j] = p;
        s[j] = p * xva_dropout_scale(drop_p, drop_seed, drop_stream, drow + j);
    }
    __syncthreads();
    for (int d = threadIdx.x; d < dk; d += RA_THREADS) {
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc += s[j] * v[(row0 + j) * ld + h * dk + d];
        for (int r = -w; r <= w; ++r) {
            const int j = i + r;
            if (j >= 0 && j < T) acc += s[j] * ev[(int64_t)(r + w) * dk + d];
        }
        out[(row0 + i) * ld_out + h * dk + d] = acc;
    }
}

// backward, per query row: dP[j] = dO_i . v_j + [|r| <= w] dO_i . emb_v[r + w];  dS = P * (dP - sum_j P dP) / sqrt(dk)  (stored over P's twin dS);
// dq_i = sum_j dS[j] k_j + sum_r dS[i + r] emb_k[r + w]
__global__ __launch_bounds__(RA_THREADS) void relattn_bwd_row_kernel(const float* __restrict__ dO, int64_t ld_do, const float* __restrict__ k,
                                                                     const float* __restrict__ v, int64_t ld, const float* __restrict__ emb_k,
                                                                     const float* __restrict__ emb_v, const int32_t* __restrict__ lens,
                                                                     const float* __restrict__ P, float* __restrict__ dS, float* __restrict__ dq,
                                                                     int64_t ld_dq, int T, int H, int dk, int w, int Hr, int Tp, int pad, float drop_p,
                                                                     uint64_t drop_seed, uint32_t drop_stream) {
    extern __shared__ float sm[];
    float* s = sm; float* doi = sm + T; float* red = doi + dk;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens[b];
    const int64_t row0 = (int64_t)b * Tp + pad;
    const float scale = rsqrtf((float)dk);
    for (int d = threadIdx.x; d < dk; d += RA_THREADS) doi[d] = dO[(row0 + i) * ld_do + h * dk + d];
    __syncthreads();
    const float* ek = emb_k + (int64_t)(Hr == 1 ? 0 : h) * (2 * w + 1) * dk;
    const float* ev = emb_v + (int64_t)(Hr == 1 ? 0 : h) * (2 * w + 1) * dk;
    const float* Pi = P + (((int64_t)b * H + h) * T + i) * T;
    float dot = 0.f;
    for (int j = threadIdx.x; j < T; j += RA_THREADS) {
        const float* vj = v + (row0 + j) * ld + h * dk;
        float acc = 0.f;
        for (int d = 0; d < dk; ++d) acc += doi[d] * vj[d];
        const int r = j - i;
        if (r >= -w && r <= w) {
            const float* e = ev + (int64_t)(r + w) * dk;
            for (int d = 0; d < dk; ++d) acc += doi[d] * e[d];
        }
        acc *= xva_dropout_scale(drop_p, drop_seed, drop_stream, (((uint64_t)b * H + h) * T + i) * T + j);   // d dropout(P) -> d P
        s[j] = acc;
        dot += Pi[j] * acc;
    }
    dot = ra_block_sum(dot, red);
    float* dSi = dS + (((int64_t)b * H + h) * T + i) * T;
    for (int j = threadIdx.x; j < T; j += RA_THREADS) {
        float g = Pi[j] * (s[j] - dot) * scale;
        if (i >= len || j >= len) g = 0.f;                    // masked_fill: no gradient into the replaced scores
        s[j] = g; dSi[j] = g;
    }
    __syncthreads();
    for (int d = threadIdx.x; d < dk; d += RA_THREADS) {
        float acc = 0.f;
        for (int j = 0; j < T; ++j) acc += s[j] * k[(row0 + j) * ld + h * dk + d];
        for (int r = -w; r <= w; ++r) {
            const int j = i + r;
            if (j >= 0 && j < T) acc += s[j] * ek[(int64_t)(r + w) * dk + d];
        }
        dq[(row0 + i) * ld_dq + h * dk + d] = acc;
    }
}
// backward, per key row j: dk_j = sum_i dS[i][j] q_i ; dv_j = sum_i P[i][j] dO_i
__global__ __launch_bounds__(RA_THREADS) void relattn_bwd_col_kernel(const float* __restrict__ dO, int64_t ld_do, const float* __restrict__ q, int64_t ld,
                                                                     const float* __restrict__ P, const float* __restrict__ dS, float* __restrict__ dk_out,
                                                                     float* __restrict__ dv_out, int64_t ld_d, int T, int H, int dk, int Tp, int pad, float drop_p,
                                                                     uint64_t drop_seed, uint32_t drop_stream) {
    extern __shared__ float sm[];
    float* ps = sm; float* ds = sm + T;
    const int j = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int64_t row0 = (int64_t)b * Tp + pad;
    const float* Pb = P + ((int64_t)b * H + h) * T * T;
    const float* Sb = dS + ((int64_t)b * H + h) * T * T;
    const uint64_t d0 = ((uint64_t)b * H + h) * T * T;
    for (int i = threadIdx.x; i < T; i += RA_THREADS) {
        ps[i] = Pb[(int64_t)i * T + j] * xva_dropout_scale(drop_p, drop_seed, drop_stream, d0 + (uint64_t)i * T + j);   // dv sees dropout(P)
        ds[i] = Sb[(int64_t)i * T + j];
    }
    __syncthreads();
    for (int d = threadIdx.x; d < dk; d += RA_THREADS) {
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < T; ++i) {
            ak += ds[i] * q[(row0 + i) * ld + h * dk + d];
            av += ps[i] * dO[(row0 + i) * ld_do + h * dk + d];
        }
        dk_out[(row0 + j) * ld_d + h * dk + d] = ak;
        dv_out[(row0 + j) * ld_d + h * dk + d] = av;
    }
}
// gradients of the relative embeddings: d emb_k[r][d] += sum_{b, h, i} dS[i][i + r] q_i[d] ; d emb_v[r][d] += sum P[i][i + r] dO_i[d]
// grid (2w + 1, H, B): one relative position of one (item, head); atomics over (b, h) — Hr * (2w + 1) * dk addresses, B * H adders each
__global__ __launch_bounds__(RA_THREADS) void relattn_bwd_emb_kernel(const float* __restrict__ dO, int64_t ld_do, const float* __restrict__ q, int64_t ld,
                                                                     const float* __restrict__ P, const float* __restrict__ dS, float* __restrict__ d_emb_k,
                                                                     float* __restrict__ d_emb_v, int T, int H, int dk, int w, int Hr, int Tp, int pad, float drop_p,
                                                                     uint64_t drop_seed, uint32_t drop_stream) {
    const int r = (int)blockIdx.x - w, h = blockIdx.y, b = blockIdx.z;
    const int64_t row0 = (int64_t)b * Tp + pad;
    const float* Pb = P + ((int64_t)b * H + h) * T * T;
    const float* Sb = dS + ((int64_t)b * H + h) * T * T;
    const uint64_t d0 = ((uint64_t)b * H + h) * T * T;
    for (int d = threadIdx.x; d < dk; d += RA_THREADS) {
        float ak = 0.f, av = 0.f;
        for (int i = 0; i < T; ++i) {
            const int j = i + r;
            if (j < 0 || j >= T) continue;
            ak += Sb[(int64_t)i * T + j] * q[(row0 + i) * ld + h * dk + d];
            av += Pb[(int64_t)i * T + j] * xva_dropout_scale(drop_p, drop_seed, drop_stream, d0 + (uint64_t)i * T + j) * dO[(row0 + i) * ld_do + h * dk + d];
        }
        const int64_t o = ((int64_t)(Hr == 1 ? 0 : h) * (2 * w + 1) + (r + w)) * dk + d;
        atomicAdd(d_emb_k + o, ak);
        atomicAdd(d_emb_v + o, av);
    }
}

extern "C" int xva_relattn_fwd(const float* q, const float* k, const float* v, int64_t ld, const float* emb_k, const float* emb_v, const int32_t* lens,
                               float* P, float* out, int64_t ld_out, int B, int T, int H, int dk, int w, int Hr, int Tp, int pad, float drop_p,
                               uint64_t drop_seed, uint32_t drop_stream, void* stream) {
    XVA_CHECK_ARG(q && k && v && emb_k && emb_v && lens && P && out, "relattn_fwd: null");
    XVA_CHECK_ARG(drop_p >= 0.f && drop_p < 1.f, "relattn_fwd: dropout probability outside [0, 1)");
    XVA_CHECK_ARG(B > 0 && T > 0 && H > 0 && dk > 0 && w >= 0 && (Hr == 1 || Hr == H) && T <= 8192, "relattn_fwd: bad dims");
    const size_t shm = (size_t)(T + dk + 4) * 4;
    hipLaunchKernelGGL(relattn_fwd_kernel, dim3(T, H, B), dim3(RA_THREADS), shm, (hipStream_t)stream, q, k, v, ld, emb_k, emb_v, lens, P, out, ld_out, T, H, dk, w,
                       Hr, Tp, pad, drop_p, drop_seed, drop_stream);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
/* dS: scratch (B, H, T, T); d_emb_k / d_emb_v are ACCUMULATED into (parameter gradients), dq / dk / dv written */
extern "C" int xva_relattn_bwd(const float* dO, int64_t ld_do, const float* q, const float* k, const float* v, int64_t ld, const float* emb_k,
                               const float* emb_v, const int32_t* lens, const float* P, float* dS, float* dq, float* dk_out, float* dv_out, int64_t ld_d,
                               float* d_emb_k, float* d_emb_v, int B, int T, int H, int dk, int w, int Hr, int Tp, int pad, float drop_p,
                               uint64_t drop_seed, uint32_t drop_stream, void* stream) {
    XVA_CHECK_ARG(dO && q && k && v && emb_k && emb_v && lens && P && dS && dq && dk_out && dv_out && d_emb_k && d_emb_v, "relattn_bwd: null");
    XVA_CHECK_ARG(B > 0 && T > 0 && H > 0 && dk > 0 && w >= 0 && (Hr == 1 || Hr == H) && T <= 8192, "relattn_bwd: bad dims");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(relattn_bwd_row_kernel, dim3(T, H, B), dim3(RA_THREADS), (size_t)(T + dk + 4) * 4, st, dO, ld_do, k, v, ld, emb_k, emb_v, lens, P, dS, dq,
                       ld_d, T, H, dk, w, Hr, Tp, pad, drop_p, drop_seed, drop_stream);
    hipLaunchKernelGGL(relattn_bwd_col_kernel, dim3(T, H, B), dim3(RA_THREADS), (size_t)(2 * T) * 4, st, dO, ld_do, q, ld, P, dS, dk_out, dv_out, ld_d, T, H, dk,
                       Tp, pad, drop_p, drop_seed, drop_stream);
    hipLaunchKernelGGL(relattn_bwd_emb_kernel, dim3(2 * w + 1, H, B), dim3(RA_THREADS), 0, st, dO, ld_do, q, ld, P, dS, d_emb_k, d_emb_v, T, H, dk, w, Hr, Tp,
                       pad, drop_p, drop_seed, drop_stream);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- LayerNorm over the channels of each row (any C) ------------------------------------------------------------------------------
// LayerNorm2 (python/xvapitch/glow_tts.py:34-56): torch layer_norm over C with gamma / beta (C,), eps 1e-5.  One wave per row.
__global__ void ln_rows_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
                                   float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int C, float eps) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mu = xva_wave_sum(s) / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = xr[c] - mu; q += d * d; }
    const float rs = rsqrtf(xva_wave_sum(q) / C + eps);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    for (int c = lane; c < C; c += 64) y[row * C + c] = (xr[c] - mu) * rs * gamma[c] + beta[c];
}
// dX = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dY * gamma ; dgamma += sum_r dY * xhat ; dbeta += sum_r dY   (row blocks + atomics)
// One pass per row: a lane holds its (up to CPL) columns of dY and X in registers (independent loads: one memory round trip per row instead
// of the four dependent ones of a column-block loop), the NEXT row's operands are in flight while this one is reduced, and a wave walks only
// a few rows — the text-encoder sized launches (1 600 rows x 196 channels) were a 16-row dependent chain per wave: 49 us for 2.5 MB.
template <int CPL>
__global__ void ln_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean, const float* __restrict__ rstd,
                                   const float* __restrict__ gamma, float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows,
                                   int C, int rows_per_wave) {
    const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int64_t r0 = wv * rows_per_wave, r1 = r0 + rows_per_wave < rows ? r0 + rows_per_wave : rows;
    if (r0 >= rows) return;
    float gm[CPL], ag[CPL], ab[CPL];
#pragma unroll
    for (int u = 0; u < CPL; ++u) { const int c = u * 64 + lane; gm[u] = c < C ? gamma[c] : 0.f; ag[u] = 0.f; ab[u] = 0.f; }
    float gy[CPL], xv[CPL], ngy[CPL], nxv[CPL], mu, rs, nmu = 0.f, nrs = 0.f;
    auto fetch = [&](int64_t r, float (&g)[CPL], float (&v)[CPL], float& m, float& s) {
        if (r >= r1) return;
        m = mean[r]; s = rstd[r];
#pragma unroll
        for (int u = 0; u < CPL; ++u) { const int c = u * 64 + lane; const bool in = c < C; g[u] = in ? dy[r * C + c] : 0.f; v[u] = in ? x[r * C + c] : m; }
    };
    fetch(r0, gy, xv, mu, rs);
    for (int64_t r = r0; r < r1; ++r) {
        fetch(r + 1, ngy, nxv, nmu, nrs);
        float s1 = 0.f, s2 = 0.f, xh[CPL], g[CPL];
#pragma unroll
        for (int u = 0; u < CPL; ++u) {
            xh[u] = (xv[u] - mu) * rs; g[u] = gy[u] * gm[u];
            s1 += g[u]; s2 += g[u] * xh[u];
            ag[u] += gy[u] * xh[u]; ab[u] += gy[u];
        }
        s1 = xva_wave_sum(s1) / C; s2 = xva_wave_sum(s2) / C;
#pragma unroll
        for (int u = 0; u < CPL; ++u) { const int c = u * 64 + lane; if (c < C) dx[r * C + c] = rs * (g[u] - s1 - xh[u] * s2); }
#pragma unroll
        for (int u = 0; u < CPL; ++u) { gy[u] = ngy[u]; xv[u] = nxv[u]; }
        mu = nmu; rs = nrs;
    }
    // the four waves of a workgroup share one atomic per column
    __shared__ float sh[2][4][CPL * 64];
    const int w = threadIdx.x >> 6;
#pragma unroll
    for (int u = 0; u < CPL; ++u) { sh[0][w][u * 64 + lane] = ag[u]; sh[1][w][u * 64 + lane] = ab[u]; }
    __syncthreads();
    const int nw = blockDim.x >> 6;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int q = 0; q < nw; ++q) {
            const int64_t qr0 = ((int64_t)blockIdx.x * nw + q) * rows_per_wave;
            if (qr0 < rows) { a += sh[0][q][c]; b += sh[1][q][c]; }
        }
        atomicAdd(dgamma + c, a); atomicAdd(dbeta + c, b);
    }
}
extern "C" int xva_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int C, float eps,
                               void* stream) {
    XVA_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows >= 0 && C > 0, "ln_rows_fwd: bad args");
    if (rows == 0) return XVA_OK;
    hipLaunchKernelGGL(ln_rows_fwd_kernel, dim3((unsigned)xva_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y, mean, rstd, rows, C, eps);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
/* dgamma / dbeta are accumulated into */
extern "C" int xva_ln_rows_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx, float* dgamma,
                               float* dbeta, int64_t rows, int C, void* stream) {
    XVA_CHECK_ARG(dy && x && mean && rstd && gamma && dx && dgamma && dbeta && rows >= 0 && C > 0, "ln_rows_bwd: bad args");
    if (rows == 0) return XVA_OK;
    XVA_CHECK_ARG(C <= 1024, "ln_rows_bwd: C up to 1024 (got %d)", C);
    // ~2048 waves (8 per CU) when there are enough rows; at most 16 rows per wave
    int rpw = (int)xva_cdiv(rows, 2048); rpw = rpw < 2 ? 2 : (rpw > 16 ? 16 : rpw);
    const int64_t waves = xva_cdiv(rows, rpw);
#define XVA_LNR(CPL) hipLaunchKernelGGL((ln_rows_bwd_kernel<CPL>), dim3((unsigned)xva_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, dx, \
                                        dgamma, dbeta, rows, C, rpw)
    if (C <= 256) XVA_LNR(4); else if (C <= 512) XVA_LNR(8); else if (C <= 768) XVA_LNR(12); else XVA_LNR(16);
#undef XVA_LNR
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- stochastic duration predictor pieces (python/xvapitch/sdp.py) ----------------------------------------------------------------------
// Tensors are fp32, time-major (B, T, C) without pad rows (token-level sizes: a few hundred rows, 192 channels); `lens` carries x_mask.
// Depthwise dilated Conv1d (groups = channels; DilatedDepthSeparableConv.convs_sep, sdp.py:66-69 on x * x_mask, :85):
//   y[b][t][c] = bias[c] + sum_j w[c][j] * xm[b][t + (j - (k-1)/2) * d][c],   xm = x inside [0, len_b), 0 outside (x * x_mask and the zero padding)
__global__ void dwconv_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ y,
                                  const int32_t* __restrict__ lens, int B, int T, int C, int k, int d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T * C) return;
    const int c = (int)(i % C);
    const int64_t bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const int len = lens[b], P = (k - 1) / 2;
    float acc = bias[c];
    for (int j = 0; j < k; ++j) {
        const int tt = t + (j - P) * d;
        if (tt >= 0 && tt < len) acc += w[c * k + j] * x[((int64_t)b * T + tt) * C + c];
    }
    y[i] = acc;
}
// dx[b][t][c] = [t < len] * sum_j w[c][j] * dy[b][t - (j - P) * d][c]
__global__ void dwconv_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx, const int32_t* __restrict__ lens,
                                       int B, int T, int C, int k, int d) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T * C) return;
    const int c = (int)(i % C);
    const int64_t bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const int P = (k - 1) / 2;
    float acc = 0.f;
    if (t < lens[b]) {
        for (int j = 0; j < k; ++j) {
            const int tt = t - (j - P) * d;
            if (tt >= 0 && tt < T) acc += w[c * k + j] * dy[((int64_t)b * T + tt) * C + c];
        }
    }
    dx[i] = acc;
}
// dw[c][j] += sum_{b,t} dy[b][t][c] * xm[b][t + (j - P) d][c] ; db[c] += sum dy   — grid (ceil(C / 64), row chunks), one atomic per (chunk, c, j)
__global__ void dwconv_bwd_weight_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db,
                                         const int32_t* __restrict__ lens, int B, int T, int C, int k, int d, int rows_per_block) {
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;                        // 4 row lanes
    __shared__ float sh[4][9][64];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, ab = 0.f;
    const int P = (k - 1) / 2;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = min((int64_t)B * T, r0 + rows_per_block);
    if (c < C) {
        for (int64_t r = r0 + rl; r < r1; r += 4) {
            const int b = (int)(r / T), t = (int)(r % T);
            const float g = dy[r * C + c];
            ab += g;
            const int len = lens[b];
            for (int j = 0; j < k; ++j) {
                const int tt = t + (j - P) * d;
                if (tt >= 0 && tt < len) acc[j] += g * x[((int64_t)b * T + tt) * C + c];
            }
        }
    }
    for (int j = 0; j < 8; ++j) sh[rl][j][threadIdx.x & 63] = acc[j];
    sh[rl][8][threadIdx.x & 63] = ab;
    __syncthreads();
    if (rl == 0 && c < C) {
        for (int j = 0; j < k; ++j) atomicAdd(dw + c * k + j, sh[0][j][threadIdx.x] + sh[1][j][threadIdx.x] + sh[2][j][threadIdx.x] + sh[3][j][threadIdx.x]);
        atomicAdd(db + c, sh[0][8][threadIdx.x] + sh[1][8][threadIdx.x] + sh[2][8][threadIdx.x] + sh[3][8][threadIdx.x]);
    }
}
// exact (erf) GELU, torch's F.gelu default (sdp.py:87,90)
__global__ void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const float v = x[i]; y[i] = 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
}
__global__ void gelu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float v = x[i];
        dx[i] = dy[i] * (0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v));
    }
}
extern "C" int xva_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, const int32_t* lens, int B, int T, int C, int k, int d, void* stream) {
    XVA_CHECK_ARG(x && w && bias && y && lens && B > 0 && T > 0 && C > 0 && k >= 1 && k <= 8 && (k & 1) && d >= 1, "dwconv_fwd: bad args");
    const int64_t n = (int64_t)B * T * C;
    hipLaunchKernelGGL(dwconv_fwd_kernel, dim3((unsigned)xva_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, lens, B, T, C, k, d);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
/* dx written; dw (C, k) and db (C) accumulated into */
extern "C" int xva_dwconv_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, const int32_t* lens, int B, int T, int C, int k,
                              int d, void* stream) {
    XVA_CHECK_ARG(dy && x && w && dx && dw && db && lens && B > 0 && T > 0 && C > 0 && k >= 1 && k <= 8 && (k & 1) && d >= 1, "dwconv_bwd: bad args");
    const int64_t n = (int64_t)B * T * C;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(dwconv_bwd_data_kernel, dim3((unsigned)xva_cdiv(n, 256)), dim3(256), 0, st, dy, w, dx, lens, B, T, C, k, d);
    const int rpb = 64;
    hipLaunchKernelGGL(dwconv_bwd_weight_kernel, dim3(xva_cdiv(C, 64), (unsigned)xva_cdiv((int64_t)B * T, rpb)), dim3(256), 0, st, dy, x, dw, db, lens, B, T, C, k, d,
                       rpb);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_gelu_fwd(const float* x, float* y, int64_t n, void* stream) {
    XVA_CHECK_ARG(x && y && n >= 0, "gelu_fwd: bad args");
    if (n) hipLaunchKernelGGL(gelu_fwd_kernel, dim3((unsigned)xva_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
    XVA_CHECK_ARG(x && dy && dx && n >= 0, "gelu_bwd: bad args");
    if (n) hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)xva_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, n);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- rational-quadratic spline with linear tails (forward direction) ----------------------------------------------------------------
// piecewise_rational_quadratic_transform(..., inverse=False, tails="linear") of python/xvapitch/util.py:203-391 as ConvFlow calls it
// (sdp.py:151-167): per element x and its 3K - 1 raw parameters h = [K widths | K heights | K - 1 derivatives] (the first two blocks scaled by
// `wh_scale` = 1 / sqrt(hidden) by the caller's convention, sdp.py:155-156): outside [-bound, bound] identity with log|det| 0; inside, the
// monotone rational-quadratic map of the bin that holds x.  One thread per element; K <= 16.
namespace {
constexpr int RQ_MAXK = 16;
constexpr float RQ_MIN_W = 1e-3f, RQ_MIN_H = 1e-3f, RQ_MIN_D = 1e-3f;
struct RqBins { float cw[RQ_MAXK + 1], ch[RQ_MAXK + 1], sw[RQ_MAXK], shh[RQ_MAXK], dv[RQ_MAXK + 1]; };
__device__ __forceinline__ void rq_softmax(const float* u, float scale, int K, float* s) {
    float mx = -3.0e38f;
    for (int i = 0; i < K; ++i) mx = fmaxf(mx, u[i] * scale);
    float sum = 0.f;
    for (int i = 0; i < K; ++i) { s[i] = __expf(u[i] * scale - mx); sum += s[i]; }
    const float inv = 1.f / sum;
    for (int i = 0; i < K; ++i) s[i] *= inv;
}
__device__ __forceinline__ float rq_softplus(float v) { return v > 20.f ? v : log1pf(__expf(v)); }
// cumulative bin edges cw / ch (K + 1), softmaxes and knot derivatives dv (K + 1; the two boundary ones are 1: linear tails)
__device__ __forceinline__ void rq_bins(const float* h, int K, float wh_scale, float bound, RqBins& b) {
    rq_softmax(h, wh_scale, K, b.sw);
    rq_softmax(h + K, wh_scale, K, b.shh);
    float aw = 0.f, ah = 0.f;
    b.cw[0] = -bound; b.ch[0] = -bound;
    for (int i = 0; i < K; ++i) {
        aw += RQ_MIN_W + (1.f - RQ_MIN_W * K) * b.sw[i];
        ah += RQ_MIN_H + (1.f - RQ_MIN_H * K) * b.shh[i];
        b.cw[i + 1] = 2.f * bound * aw - bound;
        b.ch[i + 1] = 2.f * bound * ah - bound;
    }
    b.cw[K] = bound; b.ch[K] = bound;
    const float edge = RQ_MIN_D + rq_softplus(logf(__expf(1.f - RQ_MIN_D) - 1.f));
    b.dv[0] = edge; b.dv[K] = edge;
    for (int j = 1; j < K; ++j) b.dv[j] = RQ_MIN_D + rq_softplus(h[2 * K + j - 1]);
}
__device__ __forceinline__ int rq_find(const RqBins& b, int K, float x) {
    int idx = -1;
    for (int j = 0; j <= K; ++j) idx += (x >= (j == K ? b.cw[j] + 1e-6f : b.cw[j])) ? 1 : 0;     // searchsorted (util.py:235-237)
    return min(max(idx, 0), K - 1);
}
}  // namespace

__global__ void rq_spline_fwd_kernel(const float* __restrict__ x, const float* __restrict__ h, float* __restrict__ y, float* __restrict__ logdet, int64_t n, int K,
                                     float wh_scale, float bound) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float xv = x[i];
    if (!(xv >= -bound && xv <= bound)) { y[i] = xv; logdet[i] = 0.f; return; }
    RqBins b;
    rq_bins(h + i * (3 * K - 1), K, wh_scale, bound, b);
    const int k = rq_find(b, K, xv);
    const float wk = b.cw[k + 1] - b.cw[k], hk = b.ch[k + 1] - b.ch[k], dk = b.dv[k], dk1 = b.dv[k + 1];
    const float th = (xv - b.cw[k]) / wk, om = th * (1.f - th), dl = hk / wk;
    const float num = hk * (dl * th * th + dk * om), den = dl + (dk + dk1 - 2.f * dl) * om;
    y[i] = b.ch[k] + num / den;
    const float D = dl * dl * (dk1 * th * th + 2.f * dl * om + dk * (1.f - th) * (1.f - th));
    logdet[i] = logf(D) - 2.f * logf(den);
}
// The inverse direction (util.py:325-350, what ConvFlow runs with reverse=True when the duration predictor samples, sdp.py:311-321): the bin is
// searched over the cumulative HEIGHTS, x is the root of the bin's quadratic a t^2 + b t + c = 0 taken as 2c / (-b - sqrt(b^2 - 4ac)).
// No log|det| output: the sampling direction discards it (sdp.py:174-176).
__global__ void rq_spline_inv_kernel(const float* __restrict__ y, const float* __restrict__ h, float* __restrict__ x, int64_t n, int K, float wh_scale,
                                     float bound) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float yv = y[i];
    if (!(yv >= -bound && yv <= bound)) { x[i] = yv; return; }
    RqBins b;
    rq_bins(h + i * (3 * K - 1), K, wh_scale, bound, b);
    int k = -1;
    for (int j = 0; j <= K; ++j) k += (yv >= (j == K ? b.ch[j] + 1e-6f : b.ch[j])) ? 1 : 0;     // searchsorted over cumheights (util.py:322)
    k = min(max(k, 0), K - 1);
    const float wk = b.cw[k + 1] - b.cw[k], hk = b.ch[k + 1] - b.ch[k], dk = b.dv[k], dk1 = b.dv[k + 1], dl = hk / wk;
    const float dy = yv - b.ch[k], A = dk + dk1 - 2.f * dl;
    const float qa = dy * A + hk * (dl - dk), qb = hk * dk - dy * A, qc = -dl * dy;
    const float disc = fmaxf(qb * qb - 4.f * qa * qc, 0.f);
    const float root = (2.f * qc) / (-qb - sqrtf(disc));
    x[i] = root * wk + b.cw[k];
}
// dx and dh (n, 3K - 1) from dy (gradient of y) and dl (gradient of log|det|)
__global__ void rq_spline_bwd_kernel(const float* __restrict__ x, const float* __restrict__ h, const float* __restrict__ gy_, const float* __restrict__ gl_,
                                     float* __restrict__ dx, float* __restrict__ dh, int64_t n, int K, float wh_scale, float bound) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int NP = 3 * K - 1;
    float* dhi = dh + i * NP;
    const float xv = x[i], gy = gy_[i], gL = gl_[i];
    if (!(xv >= -bound && xv <= bound)) {
        dx[i] = gy;
        for (int j = 0; j < NP; ++j) dhi[j] = 0.f;
        return;
    }
    const float* hi = h + i * NP;
    RqBins b;
    rq_bins(hi, K, wh_scale, bound, b);
    const int k = rq_find(b, K, xv);
    const float wk = b.cw[k + 1] - b.cw[k], hk = b.ch[k + 1] - b.ch[k], dk = b.dv[k], dk1 = b.dv[k + 1];
    const float th = (xv - b.cw[k]) / wk, om = th * (1.f - th), om_t = 1.f - 2.f * th, dl = hk / wk, A = dk + dk1 - 2.f * dl;
    const float num = hk * (dl * th * th + dk * om), den = dl + A * om;
    const float E = dk1 * th * th + 2.f * dl * om + dk * (1.f - th) * (1.f - th), D = dl * dl * E;
    // reverse mode through y = ch_k + num / den and L = log D - 2 log den
    const float g_num = gy / den, g_den = -gy * num / (den * den) - 2.f * gL / den, g_D = gL / D;
    const float E_t = 2.f * dk1 * th + 2.f * dl * om_t - 2.f * dk * (1.f - th);
    const float g_th = g_num * hk * (2.f * dl * th + dk * om_t) + g_den * A * om_t + g_D * dl * dl * E_t;
    const float g_dl = g_num * hk * th * th + g_den * (1.f - 2.f * om) + g_D * (2.f * dl * E + dl * dl * 2.f * om);
    float g_hk = g_num * (dl * th * th + dk * om) + g_dl / wk;
    const float g_dk = g_num * hk * om + g_den * om + g_D * dl * dl * (1.f - th) * (1.f - th);
    const float g_dk1 = g_den * om + g_D * dl * dl * th * th;
    dx[i] = g_th / wk;
    const float g_cwk = -g_th / wk, g_wk = -g_th * th / wk - g_dl * dl / wk, g_chk = gy;
    // bin k's width / height / edges -> the normalised widths / heights -> softmax inputs
    float gcw[RQ_MAXK + 1], gch[RQ_MAXK + 1];
    for (int j = 0; j <= K; ++j) { gcw[j] = 0.f; gch[j] = 0.f; }
    gcw[k + 1] += g_wk; gcw[k] += g_cwk - g_wk;
    gch[k + 1] += g_hk; gch[k] += g_chk - g_hk;
    float tw = 0.f, thh = 0.f, gsw[RQ_MAXK], gsh[RQ_MAXK], dotw = 0.f, doth = 0.f;
    for (int ii = K - 1; ii >= 0; --ii) {                    // g w_i = 2 bound * sum_{j > i, j <= K - 1} g cw_j   (cw_0 and cw_K are constants)
        gsw[ii] = 2.f * bound * tw * (1.f - RQ_MIN_W * K);
        gsh[ii] = 2.f * bound * thh * (1.f - RQ_MIN_H * K);
        if (ii >= 1) { tw += gcw[ii]; thh += gch[ii]; }
    }
    // note: the loop above adds edge ii AFTER using the running sum, so bin ii sees edges ii + 1 .. K - 1
    for (int ii = 0; ii < K; ++ii) { dotw += b.sw[ii] * gsw[ii]; doth += b.shh[ii] * gsh[ii]; }
    for (int ii = 0; ii < K; ++ii) {
        dhi[ii] = b.sw[ii] * (gsw[ii] - dotw) * wh_scale;
        dhi[K + ii] = b.shh[ii] * (gsh[ii] - doth) * wh_scale;
    }
    for (int j = 1; j < K; ++j) {
        const float g = (j == k ? g_dk : 0.f) + (j == k + 1 ? g_dk1 : 0.f);
        const float u = hi[2 * K + j - 1];
        dhi[2 * K + j - 1] = g / (1.f + __expf(-u));          // softplus' = sigmoid
    }
}
extern "C" int xva_rq_spline_fwd(const float* x, const float* h, float* y, float* logdet, int64_t n, int K, float wh_scale, float bound, void* stream) {
    XVA_CHECK_ARG(x && h && y && logdet && n >= 0 && K >= 2 && K <= RQ_MAXK && bound > 0.f, "rq_spline_fwd: bad args");
    if (n) hipLaunchKernelGGL(rq_spline_fwd_kernel, dim3((unsigned)xva_cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, x, h, y, logdet, n, K, wh_scale, bound);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_rq_spline_inv(const float* y, const float* h, float* x, int64_t n, int K, float wh_scale, float bound, void* stream) {
    XVA_CHECK_ARG(y && h && x && n >= 0 && K >= 2 && K <= RQ_MAXK && bound > 0.f, "rq_spline_inv: bad args");
    if (n) hipLaunchKernelGGL(rq_spline_inv_kernel, dim3((unsigned)xva_cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, y, h, x, n, K, wh_scale, bound);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_rq_spline_bwd(const float* x, const float* h, const float* dy, const float* dlogdet, float* dx, float* dh, int64_t n, int K, float wh_scale,
                                 float bound, void* stream) {
    XVA_CHECK_ARG(x && h && dy && dlogdet && dx && dh && n >= 0 && K >= 2 && K <= RQ_MAXK && bound > 0.f, "rq_spline_bwd: bad args");
    if (n) hipLaunchKernelGGL(rq_spline_bwd_kernel, dim3((unsigned)xva_cdiv(n, 128)), dim3(128), 0, (hipStream_t)stream, x, h, dy, dlogdet, dx, dh, n, K, wh_scale,
                              bound);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- ElementwiseAffine (sdp.py:95-114) and the dequantisation step of the duration predictor (sdp.py:283-296), (B, T, C) fp32 -------------------
// y = (x * exp(log_scale[c]) + translation[c]) * mask ; logdet[b] = len_b * sum_c log_scale[c]
__global__ void affine_fwd_kernel(const float* __restrict__ x, const float* __restrict__ ls, const float* __restrict__ tr, float* __restrict__ y,
                                  float* __restrict__ logdet, const int32_t* __restrict__ lens, int B, int T, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T * C) return;
    const int c = (int)(i % C);
    const int64_t bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    y[i] = t < lens[b] ? x[i] * __expf(ls[c]) + tr[c] : 0.f;
    if (t == 0 && c == 0) { float s = 0.f; for (int k = 0; k < C; ++k) s += ls[k]; logdet[b] = s * (float)lens[b]; }
}
// dx = dy * exp(ls) * mask ; d ls[c] += sum dy * x * exp(ls) * mask + sum_b dlogdet[b] * len_b ; d tr[c] += sum dy * mask     (tiny tensors: atomics)
__global__ void affine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ ls, const float* __restrict__ dy, const float* __restrict__ dlogdet,
                                  float* __restrict__ dx, float* __restrict__ dls, float* __restrict__ dtr, const int32_t* __restrict__ lens, int B, int T, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T * C) return;
    const int c = (int)(i % C);
    const int64_t bt = i / C;
    const int t = (int)(bt % T), b = (int)(bt / T);
    const bool live = t < lens[b];
    const float e = __expf(ls[c]), g = live ? dy[i] : 0.f;
    dx[i] = g * e;
    if (live) { atomicAdd(dls + c, g * x[i] * e); atomicAdd(dtr + c, g); }
    if (t == 0) atomicAdd(dls + c, dlogdet[b] * (float)lens[b]);
}
// u = sigmoid(z_u) * m ; z0 = (dr - u) * m ; out0 = log(max(z0, 1e-5)) * m ; out1 = (logsigmoid(z_u) + logsigmoid(-z_u)) * m          (per token)
__global__ void sdp_dequant_fwd_kernel(const float* __restrict__ zu, const float* __restrict__ dr, float* __restrict__ z0log, float* __restrict__ lsig,
                                       const int32_t* __restrict__ lens, int B, int T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int t = (int)(i % T), b = (int)(i / T);
    if (t >= lens[b]) { z0log[i] = 0.f; lsig[i] = 0.f; return; }
    const float v = zu[i];
    const float sp = fmaxf(v, 0.f) + log1pf(__expf(-fabsf(v)));      // softplus(v): logsigmoid(v) = v - sp, logsigmoid(-v) = -sp
    const float s = 1.f / (1.f + __expf(-v));
    z0log[i] = logf(fmaxf(dr[i] - s, 1e-5f));
    lsig[i] = v - 2.f * sp;
}
__global__ void sdp_dequant_bwd_kernel(const float* __restrict__ zu, const float* __restrict__ dr, const float* __restrict__ d_z0log, const float* __restrict__ d_lsig,
                                       float* __restrict__ d_zu, const int32_t* __restrict__ lens, int B, int T) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * T) return;
    const int t = (int)(i % T), b = (int)(i / T);
    if (t >= lens[b]) { d_zu[i] = 0.f; return; }
    const float v = zu[i], s = 1.f / (1.f + __expf(-v)), z0 = dr[i] - s;
    float g = d_lsig[i] * (1.f - 2.f * s);
    if (z0 > 1e-5f) g += d_z0log[i] * (-s * (1.f - s)) / z0;
    d_zu[i] = g;
}
extern "C" int xva_affine_fwd(const float* x, const float* log_scale, const float* translation, float* y, float* logdet, const int32_t* lens, int B, int T, int C,
                              void* stream) {
    XVA_CHECK_ARG(x && log_scale && translation && y && logdet && lens && B > 0 && T > 0 && C > 0, "affine_fwd: bad args");
    hipLaunchKernelGGL(affine_fwd_kernel, dim3((unsigned)xva_cdiv((int64_t)B * T * C, 256)), dim3(256), 0, (hipStream_t)stream, x, log_scale, translation, y, logdet, lens,
                       B, T, C);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
/* d_log_scale / d_translation (C) are accumulated into */
extern "C" int xva_affine_bwd(const float* x, const float* log_scale, const float* dy, const float* dlogdet, float* dx, float* d_log_scale, float* d_translation,
                              const int32_t* lens, int B, int T, int C, void* stream) {
    XVA_CHECK_ARG(x && log_scale && dy && dlogdet && dx && d_log_scale && d_translation && lens && B > 0 && T > 0 && C > 0, "affine_bwd: bad args");
    hipLaunchKernelGGL(affine_bwd_kernel, dim3((unsigned)xva_cdiv((int64_t)B * T * C, 256)), dim3(256), 0, (hipStream_t)stream, x, log_scale, dy, dlogdet, dx, d_log_scale,
                       d_translation, lens, B, T, C);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_sdp_dequant_fwd(const float* z_u, const float* dr, float* z0_log, float* logsig, const int32_t* lens, int B, int T, void* stream) {
    XVA_CHECK_ARG(z_u && dr && z0_log && logsig && lens && B > 0 && T > 0, "sdp_dequant_fwd: bad args");
    hipLaunchKernelGGL(sdp_dequant_fwd_kernel, dim3((unsigned)xva_cdiv((int64_t)B * T, 256)), dim3(256), 0, (hipStream_t)stream, z_u, dr, z0_log, logsig, lens, B, T);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_sdp_dequant_bwd(const float* z_u, const float* dr, const float* d_z0_log, const float* d_logsig, float* d_z_u, const int32_t* lens, int B, int T,
                                   void* stream) {
    XVA_CHECK_ARG(z_u && dr && d_z0_log && d_logsig && d_z_u && lens && B > 0 && T > 0, "sdp_dequant_bwd: bad args");
    hipLaunchKernelGGL(sdp_dequant_bwd_kernel, dim3((unsigned)xva_cdiv((int64_t)B * T, 256)), dim3(256), 0, (hipStream_t)stream, z_u, dr, d_z0_log, d_logsig, d_z_u, lens,
                       B, T);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
