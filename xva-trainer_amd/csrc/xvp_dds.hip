// xvp_dds.hip — DilatedDepthSeparableConv of xVAPitch's stochastic duration predictor (python/xvapitch/sdp.py:40-93) as two engine calls.
//
// The duration predictor runs ten of these stacks per training iteration (its own `convs` / `post_convs` and one inside each of the eight ConvFlows,
// sdp.py:116-176,196-260) on (B, T, hidden) = (16, 100, 192) tensors: 3 layers x {depthwise dilated conv -> LayerNorm2 -> GELU -> 1x1 conv -> LayerNorm2 -> GELU ->
// [Dropout] -> + x} and the mirror image backwards — ~660 launches of 3 - 15 us whose issue from Python (xva-trainer_amd/xvapitch/sdp.py _dds_fwd / _dds_bwd) cost
// more host time than any other module of the iteration.  Same kernels, same order; the host side is this C++ loop.  Tensors are (B, T, C) fp32 without pad rows;
// `lens` = the rows of x_mask.  The workspace needs no initialisation: every buffer is written before it is read.
#include "xva_common.h"
#include "xva_gemm.h"
#include "xva_hip.h"

namespace {
struct Lay { float *cur, *t1, *m1, *r1, *n1, *a1, *t2, *m2, *r2, *n2, *a2, *nxt; };
struct W {
    float* x0;                                   // x [+ g]
    Lay l[16];
    float *d, *da2, *dn2, *dt2, *da1, *dn1, *dt1, *dxb[2];
    int64_t floats;
};
struct Carver {
    float* base; int64_t off;
    float* take(int64_t n) { n = (n + 63) / 64 * 64; float* p = base ? base + off : nullptr; off += n; return p; }
};
void carve(const xva_xvp_dds_dims* d, float* base, W& w) {
    Carver c{base, 0};
    const int64_t rows = (int64_t)d->B * d->T, n = rows * d->C;
    w.x0 = c.take(n);
    for (int i = 0; i < d->L; i++) {
        Lay& l = w.l[i];
        l.t1 = c.take(n); l.m1 = c.take(rows); l.r1 = c.take(rows); l.n1 = c.take(n); l.a1 = c.take(n); l.t2 = c.take(n); l.m2 = c.take(rows); l.r2 = c.take(rows);
        l.n2 = c.take(n); l.a2 = c.take(n);
        l.nxt = d->p_drop > 0.f ? c.take(n) : l.a2;
        l.cur = i == 0 ? w.x0 : w.l[i - 1].nxt;
    }
    w.d = c.take(n); w.da2 = c.take(n); w.dn2 = c.take(n); w.dt2 = c.take(n); w.da1 = c.take(n); w.dn1 = c.take(n); w.dt1 = c.take(n);
    w.dxb[0] = c.take(n); w.dxb[1] = c.take(n);
    w.floats = c.off;
}
int check_dims(const xva_xvp_dds_dims* d) {
    XVA_CHECK_ARG(d && d->B > 0 && d->T > 0 && d->C > 0 && d->C % 4 == 0 && d->L >= 1 && d->L <= 16 && (d->k & 1) && d->k <= 7 && d->p_drop >= 0.f && d->p_drop < 1.f,
                  "xvp_dds: B, T > 0; C a multiple of 4; 1 <= L <= 16; k odd <= 7; p_drop in [0, 1)");
    return XVA_OK;
}
xva_gemm_params gemm_base() {
    xva_gemm_params p;
    memset(&p, 0, sizeof(p));
    p.batch = 1; p.batch2 = 1; p.alpha = 1.f; p.beta = 1.f; p.splitk = 1; p.mask_mul = 1; p.mask_pad = 1;
    return p;
}
__global__ void add2_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 x = a[i], y = b[i];
    o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}
// dst = src * x_mask on (B, T, C) rows
__global__ void copy_mask_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4, int C4, int T, const int32_t* __restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int64_t r = i / C4;
    const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
    dst[i] = t < lens[b] ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__device__ __forceinline__ float gelu_f(float v) { return 0.5f * v * (1.f + erff(v * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_d(float v) { return 0.5f * (1.f + erff(v * 0.70710678118654752f)) + v * 0.3989422804014327f * __expf(-0.5f * v * v); }

// ---- fused row kernels (one wave per (b, t) row; the same per-lane summation orders as dwconv_fwd / ln_rows_fwd / gelu / dropout_apply / add, so the
// results are the separate kernels' bit for bit) -------------------------------------------------------------------------------------------------
// front half of a layer: t1 = depthwise dilated conv(cur * x_mask) ; n1 = LayerNorm2(t1) ; a1 = GELU(n1)
__global__ void dds_front_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, float* __restrict__ t1, float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ n1,
                                 float* __restrict__ a1, const int32_t* __restrict__ lens, int64_t rows, int T, int C, int k, int d) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int b = (int)(row / T), t = (int)(row - (int64_t)b * T), len = lens[b], P = (k - 1) / 2;
    float* tr = t1 + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) {
        float acc = bias[c];
        for (int j = 0; j < k; ++j) {
            const int tt = t + (j - P) * d;
            if (tt >= 0 && tt < len) acc += w[c * k + j] * x[((int64_t)b * T + tt) * C + c];
        }
        tr[c] = acc; s += acc;
    }
    const float mu = xva_wave_sum(s) / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float e = tr[c] - mu; q += e * e; }       // a lane re-reads only what it wrote
    const float rs = rsqrtf(xva_wave_sum(q) / C + 1e-5f);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    for (int c = lane; c < C; c += 64) {
        const float v = (tr[c] - mu) * rs * gamma[c] + beta[c];
        n1[row * C + c] = v; a1[row * C + c] = gelu_f(v);
    }
}
// back half: n2 = LayerNorm2(t2) ; nxt = Dropout(GELU(n2)) + cur
__global__ void dds_back_kernel(const float* __restrict__ t2, const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ cur,
                                float* __restrict__ mean, float* __restrict__ rstd, float* __restrict__ n2, float* __restrict__ nxt, int64_t rows, int C, float p,
                                uint64_t seed, uint32_t site) {
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* xr = t2 + row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += xr[c];
    const float mu = xva_wave_sum(s) / C;
    float q = 0.f;
    for (int c = lane; c < C; c += 64) { const float e = xr[c] - mu; q += e * e; }
    const float rs = rsqrtf(xva_wave_sum(q) / C + 1e-5f);
    if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
    for (int c = lane; c < C; c += 64) {
        const int64_t i = row * C + c;
        const float v = (xr[c] - mu) * rs * gamma[c] + beta[c];
        n2[i] = v;
        nxt[i] = gelu_f(v) * xva_dropout_scale(p, seed, site, (uint64_t)i) + cur[i];
    }
}
// backward through [Dropout ->] GELU -> LayerNorm2 in one pass: g = dy * drop * gelu'(n), then dx = rstd (g gamma - mean(g gamma) - xhat mean(g gamma xhat)),
// dgamma += sum_r g xhat, dbeta += sum_r g  (C <= 256: a lane holds its four columns; a wave walks `rpw` rows, a workgroup shares one atomic per column)
__global__ void dds_ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ nrm, const float* __restrict__ x, const float* __restrict__ mean,
                                  const float* __restrict__ rstd, const float* __restrict__ gamma, float* __restrict__ dx, float* __restrict__ dgamma,
                                  float* __restrict__ dbeta, int64_t rows, int C, int rpw, float p, uint64_t seed, uint32_t site) {
    const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t r0 = wv * rpw, r1 = r0 + rpw < rows ? r0 + rpw : rows;
    float gm[4], ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = u * 64 + lane; gm[u] = c < C ? gamma[c] : 0.f; }
    for (int64_t r = r0; r < r1; ++r) {
        const float mu = mean[r], rs = rstd[r];
        float s1 = 0.f, s2 = 0.f, xh[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = u * 64 + lane;
            float gy = 0.f; xh[u] = 0.f;
            if (c < C) {
                const int64_t i = r * C + c;
                gy = dy[i] * xva_dropout_scale(p, seed, site, (uint64_t)i) * gelu_d(nrm[i]);
                xh[u] = (x[i] - mu) * rs;
            }
            g[u] = gy * gm[u];
            s1 += g[u]; s2 += g[u] * xh[u];
            ag[u] += gy * xh[u]; ab[u] += gy;
        }
        s1 = xva_wave_sum(s1) / C; s2 = xva_wave_sum(s2) / C;
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int c = u * 64 + lane; if (c < C) dx[r * C + c] = rs * (g[u] - s1 - xh[u] * s2); }
    }
    __shared__ float sh[2][4][256];
#pragma unroll
    for (int u = 0; u < 4; ++u) { sh[0][w][u * 64 + lane] = ag[u]; sh[1][w][u * 64 + lane] = ab[u]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { a += sh[0][q][c]; b += sh[1][q][c]; }
        atomicAdd(dgamma + c, a); atomicAdd(dbeta + c, b);
    }
}
// d W (M, N) += dt^T a ; d b (M) += column sums of dt, for the 1x1 convolutions (rows x M and rows x N operands, M, N <= a few hundred): 32 x 32 output tiles,
// the row range split over blockIdx.z, fp32 FMAs, one atomic per output and split.  Replaces a split-K product + its slab reduce + the column-sum launch on a
// (16 x 100) x 192 problem where each of the three was launch latency.
__global__ __launch_bounds__(256) void small_wgrad_kernel(const float* __restrict__ dt, const float* __restrict__ a, float* __restrict__ dW, float* __restrict__ db,
                                                          int64_t rows, int M, int N, int rows_per_split) {
    __shared__ float sd[32][33], sa[32][33];
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_split, r1 = r0 + rows_per_split < rows ? r0 + rows_per_split : rows;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                 // thread: outputs (m0 + ty + 8 i, n0 + tx), i < 4
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, bs = 0.f;
    for (int64_t r = r0; r < r1; r += 32) {
        for (int i = ty; i < 32; i += 8) {                                  // tile rows r .. r + 31, columns m0.. / n0..
            const int64_t rr = r + i;
            sd[i][tx] = (rr < r1 && m0 + tx < M) ? dt[rr * M + m0 + tx] : 0.f;
            sa[i][tx] = (rr < r1 && n0 + tx < N) ? a[rr * N + n0 + tx] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int q = 0; q < 32; ++q) {
            const float av = sa[q][tx];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] += sd[q][ty + 8 * i] * av;
        }
        if (blockIdx.x == 0 && ty == 0) {
#pragma unroll 8
            for (int q = 0; q < 32; ++q) bs += sd[q][tx];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int m = m0 + ty + 8 * i, n = n0 + tx; if (m < M && n < N) atomicAdd(dW + (int64_t)m * N + n, acc[i]); }
    if (blockIdx.x == 0 && ty == 0 && m0 + tx < M) atomicAdd(db + m0 + tx, bs);
}
// backward of ConvFlow's `pre` (Conv1d(1, H, 1): h[r, c] = b[c] + x0[r] w[c]) joined with the assembly of d z: d z[r] = (sum_c dh[r, c] w[c] + d x0'[r], d x1[r]),
// d w[c] += sum_r dh[r, c] x0[r], d b[c] += sum_r dh[r, c]   (H <= 256: a lane holds its four columns; a wave walks `rpw` rows; one atomic per column and workgroup)
__global__ void cf_pre_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ w, const float* __restrict__ x0, const float* __restrict__ dx0p,
                                  const float* __restrict__ dx1, float* __restrict__ dz, float* __restrict__ dw, float* __restrict__ db, int64_t rows, int H, int rpw) {
    const int64_t wv = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63, wi = threadIdx.x >> 6;
    const int64_t r0 = wv * rpw, r1 = r0 + rpw < rows ? r0 + rpw : rows;
    float wv4[4], aw[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int c = u * 64 + lane; wv4[u] = c < H ? w[c] : 0.f; }
    for (int64_t r = r0; r < r1; ++r) {
        const float xr = x0[r];
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = u * 64 + lane;
            const float d = c < H ? dh[r * H + c] : 0.f;
            s += d * wv4[u]; aw[u] += d * xr; ab[u] += d;
        }
        s = xva_wave_sum(s);
        if (lane == 0) { dz[2 * r] = s + dx0p[r]; dz[2 * r + 1] = dx1[r]; }
    }
    __shared__ float sh[2][4][256];
#pragma unroll
    for (int u = 0; u < 4; ++u) { sh[0][wi][u * 64 + lane] = aw[u]; sh[1][wi][u * 64 + lane] = ab[u]; }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) { a += sh[0][q][c]; b += sh[1][q][c]; }
        atomicAdd(dw + c, a); atomicAdd(db + c, b);
    }
}
const bool g_fused = [] { const char* e = getenv("XVA_XVP_DDS_FUSED"); return e ? atoi(e) != 0 : true; }();
}  // namespace

extern "C" int64_t xva_xvp_dds_workspace_bytes(const xva_xvp_dds_dims* d) {
    if (check_dims(d) != XVA_OK) return -1;
    W w;
    carve(d, nullptr, w);
    return w.floats * 4;
}

extern "C" int xva_xvp_dds_forward(const xva_xvp_dds_dims* d, const float* const* prm, const float* x, const float* g, const int32_t* lens, float* out, void* workspace,
                                   int64_t workspace_bytes, void* stream) {
    XVA_TRY(check_dims(d));
    XVA_CHECK_ARG(prm && x && lens && out && workspace && ((uintptr_t)workspace % 16) == 0, "xvp_dds_forward: null / unaligned argument");
    W w;
    carve(d, (float*)workspace, w);
    XVA_CHECK_ARG(workspace_bytes >= w.floats * 4, "xvp_dds_forward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int B = d->B, T = d->T, C = d->C, k = d->k;
    const int64_t rows = (int64_t)B * T, n = rows * C;
    if (g) {
        hipLaunchKernelGGL(add2_kernel, dim3((unsigned)xva_cdiv(n / 4, 256)), dim3(256), 0, s, (const float4*)x, (const float4*)g, (float4*)w.x0, n / 4);
        XVA_LAUNCH_CHECK();
    } else if (hipMemcpyAsync(w.x0, x, (size_t)n * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) { xva_set_error("xvp_dds_forward: copy failed"); return XVA_ERR_HIP; }
    int dil = 1;
    for (int i = 0; i < d->L; i++, dil *= k) {
        const Lay& l = w.l[i];
        const float* const* p = prm + 8 * i;                       // convs_sep w (C, 1, k), b ; convs_1x1 w (C, C, 1), b ; norms_1 gamma, beta ; norms_2 gamma, beta
        const bool fused = g_fused;
        if (fused) {
            hipLaunchKernelGGL(dds_front_kernel, dim3((unsigned)xva_cdiv(rows, 4)), dim3(256), 0, s, l.cur, p[0], p[1], p[4], p[5], l.t1, l.m1, l.r1, l.n1, l.a1, lens, rows, T, C,
                               k, dil);                                                                             // sdp.py:85-87
            XVA_LAUNCH_CHECK();
        } else {
            XVA_TRY(xva_dwconv_fwd(l.cur, p[0], p[1], l.t1, lens, B, T, C, k, dil, stream));                        // sdp.py:85
            XVA_TRY(xva_ln_rows_fwd(l.t1, p[4], p[5], l.n1, l.m1, l.r1, rows, C, 1e-5f, stream));                   // :86
            XVA_TRY(xva_gelu_fwd(l.n1, l.a1, n, stream));                                                           // :87
        }
        xva_gemm_params q = gemm_base();                                                                            // :88
        q.A = l.a1; q.B = p[2]; q.C = l.t2; q.M = (int32_t)rows; q.N = C; q.K = C; q.lda = C; q.ldb = C; q.ldc = C; q.layout = XVA_GEMM_NT; q.bias = p[3];
        XVA_TRY(xva_gemm(&q, stream));
        if (fused) {
            hipLaunchKernelGGL(dds_back_kernel, dim3((unsigned)xva_cdiv(rows, 4)), dim3(256), 0, s, l.t2, p[6], p[7], l.cur, l.m2, l.r2, l.n2, l.nxt, rows, C, d->p_drop, d->seed,
                               d->site0 + i);                                                                       // :89-92
            XVA_LAUNCH_CHECK();
        } else {
            XVA_TRY(xva_ln_rows_fwd(l.t2, p[6], p[7], l.n2, l.m2, l.r2, rows, C, 1e-5f, stream));                   // :89
            XVA_TRY(xva_gelu_fwd(l.n2, l.a2, n, stream));                                                           // :90
            if (d->p_drop > 0.f) XVA_TRY(xva_dropout_apply(l.a2, l.nxt, 0, n, d->p_drop, d->seed, d->site0 + i, stream));   // :91
            XVA_TRY(xva_fp_add_act(l.nxt, l.cur, 0, n, stream));                                                    // x = x + y (:92)
        }
    }
    hipLaunchKernelGGL(copy_mask_kernel, dim3((unsigned)xva_cdiv(n / 4, 256)), dim3(256), 0, s, (const float4*)w.l[d->L - 1].nxt, (float4*)out, n / 4, C / 4, T, lens);   // :93
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

extern "C" int xva_xvp_dds_backward(const xva_xvp_dds_dims* d, const float* const* prm, float* const* grd, const float* dy, const int32_t* lens, float* dx, void* workspace,
                                    int64_t workspace_bytes, void* sk_ws, int64_t sk_ws_bytes, void* stream) {
    XVA_TRY(check_dims(d));
    XVA_CHECK_ARG(prm && grd && dy && lens && dx && workspace, "xvp_dds_backward: null argument");
    W w;
    carve(d, (float*)workspace, w);
    XVA_CHECK_ARG(workspace_bytes >= w.floats * 4, "xvp_dds_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int B = d->B, T = d->T, C = d->C, k = d->k;
    const int64_t rows = (int64_t)B * T, n = rows * C;
    hipLaunchKernelGGL(copy_mask_kernel, dim3((unsigned)xva_cdiv(n / 4, 256)), dim3(256), 0, s, (const float4*)dy, (float4*)w.d, n / 4, C / 4, T, lens);
    XVA_LAUNCH_CHECK();
    float* dcur = w.d;
    int dil = 1;
    for (int i = 1; i < d->L; i++) dil *= k;
    for (int i = d->L - 1; i >= 0; i--, dil /= k) {
        const Lay& l = w.l[i];
        const float* const* p = prm + 8 * i;
        float* const* gr = grd + 8 * i;
        const bool fused = g_fused && C <= 256;
        int rpw = (int)xva_cdiv(rows, 2048); rpw = rpw < 2 ? 2 : (rpw > 16 ? 16 : rpw);
        const unsigned lnb = (unsigned)xva_cdiv(xva_cdiv(rows, rpw), 4);
        if (fused) {
            hipLaunchKernelGGL(dds_ln_bwd_kernel, dim3(lnb), dim3(256), 0, s, dcur, l.n2, l.t2, l.m2, l.r2, p[6], w.dt2, gr[6], gr[7], rows, C, rpw, d->p_drop, d->seed,
                               d->site0 + i);
            XVA_LAUNCH_CHECK();
        } else {
            const float* da2 = dcur;
            if (d->p_drop > 0.f) { XVA_TRY(xva_dropout_apply(dcur, w.da2, 0, n, d->p_drop, d->seed, d->site0 + i, stream)); da2 = w.da2; }
            XVA_TRY(xva_gelu_bwd(l.n2, da2, w.dn2, n, stream));
            XVA_TRY(xva_ln_rows_bwd(w.dn2, l.t2, l.m2, l.r2, p[6], w.dt2, gr[6], gr[7], rows, C, stream));
        }
        xva_gemm_params q = gemm_base();                                                                            // d a1 = d t2 W
        q.A = w.dt2; q.B = p[2]; q.C = w.da1; q.M = (int32_t)rows; q.N = C; q.K = C; q.lda = C; q.ldb = C; q.ldc = C; q.layout = XVA_GEMM_NN;
        XVA_TRY(xva_gemm(&q, stream));
        if (fused) {                                                                                                // d W += d t2^T a1 ; d b += colsum(d t2)
            const int rps = 224;
            hipLaunchKernelGGL(small_wgrad_kernel, dim3((unsigned)xva_cdiv(C, 32), (unsigned)xva_cdiv(C, 32), (unsigned)xva_cdiv(rows, rps)), dim3(256), 0, s, w.dt2, l.a1, gr[2],
                               gr[3], rows, C, C, rps);
            XVA_LAUNCH_CHECK();
        } else {
            xva_gemm_params t = gemm_base();
            t.A = w.dt2; t.B = l.a1; t.C = gr[2]; t.M = C; t.N = C; t.K = (int32_t)rows; t.lda = C; t.ldb = C; t.ldc = C; t.layout = XVA_GEMM_TN; t.accumulate = 1; t.splitk = 0;
            t.sk_ws = sk_ws; t.sk_ws_bytes = sk_ws_bytes;
            XVA_TRY(xva_gemm(&t, stream));
            XVA_TRY(xva_hg_colsum(w.dt2, 0, gr[3], rows, C, 1.f, stream));
        }
        if (fused) {
            hipLaunchKernelGGL(dds_ln_bwd_kernel, dim3(lnb), dim3(256), 0, s, w.da1, l.n1, l.t1, l.m1, l.r1, p[4], w.dt1, gr[4], gr[5], rows, C, rpw, 0.f, (uint64_t)0, 0u);
            XVA_LAUNCH_CHECK();
        } else {
            XVA_TRY(xva_gelu_bwd(l.n1, w.da1, w.dn1, n, stream));
            XVA_TRY(xva_ln_rows_bwd(w.dn1, l.t1, l.m1, l.r1, p[4], w.dt1, gr[4], gr[5], rows, C, stream));
        }
        float* dxb = i == 0 ? dx : w.dxb[i & 1];
        XVA_TRY(xva_dwconv_bwd(w.dt1, l.cur, p[0], dxb, gr[0], gr[1], lens, B, T, C, k, dil, stream));
        XVA_TRY(xva_fp_add_act(dxb, dcur, 0, n, stream));                                                           // d(x) = d(residual) + d(branch)
        dcur = dxb;
    }
    return XVA_OK;
}

// ---- two pieces of ConvFlow's backward (python/xvapitch/sdp.py:116-176) that were seven and four launches of torch glue -----------------------------------------
extern "C" int xva_small_wgrad(const float* dy, const float* x, float* dW, float* db, int64_t rows, int M, int N, void* stream) {
    XVA_CHECK_ARG(dy && x && dW && db && rows >= 0 && M > 0 && N > 0, "small_wgrad: bad arguments");
    if (rows == 0) return XVA_OK;
    const int rps = 224;
    hipLaunchKernelGGL(small_wgrad_kernel, dim3((unsigned)xva_cdiv(N, 32), (unsigned)xva_cdiv(M, 32), (unsigned)xva_cdiv(rows, rps)), dim3(256), 0, (hipStream_t)stream, dy, x, dW, db,
                       rows, M, N, rps);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_cf_pre_bwd(const float* dh, const float* pre_w, const float* x0, const float* d_x0_pass, const float* d_x1, float* dz, float* d_pre_w, float* d_pre_b,
                              int64_t rows, int H, void* stream) {
    XVA_CHECK_ARG(dh && pre_w && x0 && d_x0_pass && d_x1 && dz && d_pre_w && d_pre_b && rows >= 0 && H > 0 && H <= 256, "cf_pre_bwd: bad arguments (H <= 256)");
    if (rows == 0) return XVA_OK;
    int rpw = (int)xva_cdiv(rows, 2048); rpw = rpw < 2 ? 2 : (rpw > 16 ? 16 : rpw);
    hipLaunchKernelGGL(cf_pre_bwd_kernel, dim3((unsigned)xva_cdiv(xva_cdiv(rows, rpw), 4)), dim3(256), 0, (hipStream_t)stream, dh, pre_w, x0, d_x0_pass, d_x1, dz, d_pre_w, d_pre_b, rows,
                       H, rpw);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- the rest of ConvFlow's glue (sdp.py:147-176), one launch where torch needed two to five -----------------------------------------------------------------------
namespace {
// z (rows, 2) -> x0, x1 (rows) contiguous and h (rows, H) = b + x0 w   (the `pre` Conv1d(1, H, 1))
__global__ void cf_pre_fwd_kernel(const float* __restrict__ z, const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ x0, float* __restrict__ x1,
                                  float* __restrict__ h, int64_t rows, int H) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * H) return;
    const int64_t r = i / H; const int c = (int)(i - r * H);
    const float a = z[2 * r];
    h[i] = b[c] + a * w[c];
    if (c == 0) { x0[r] = a; x1[r] = z[2 * r + 1]; }
}
// hs (rows, NP) = hp (rows, NPp)[:, :NP] * x_mask
__global__ void cf_mask_slice_kernel(const float* __restrict__ hp, float* __restrict__ hs, int64_t rows, int NPp, int NP, int T, const int32_t* __restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * NP) return;
    const int64_t r = i / NP; const int c = (int)(i - r * NP);
    const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
    hs[i] = t < lens[b] ? hp[r * NPp + c] : 0.f;
}
// out (rows, 2) = [x0, y1] * x_mask ; ld *= x_mask ; ldsum[b] = sum_t ld[b, t]      (one workgroup per item)
__global__ void cf_post_fwd_kernel(const float* __restrict__ x0, const float* __restrict__ y1, float* __restrict__ ld, float* __restrict__ out, float* __restrict__ ldsum, int T,
                                   const int32_t* __restrict__ lens) {
    __shared__ float sh[16];
    const int b = blockIdx.x, len = lens[b];
    float s = 0.f;
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        const int64_t r = (int64_t)b * T + t;
        const bool live = t < len;
        const float l = live ? ld[r] : 0.f;
        ld[r] = l; s += l;
        out[2 * r] = live ? x0[r] : 0.f; out[2 * r + 1] = live ? y1[r] : 0.f;
    }
    s = xva_block_sum(s, sh);
    if (threadIdx.x == 0) ldsum[b] = s;
}
// d_out (rows, 2), d_logdet (B) -> d x0' (rows), d y1 (rows), d ld (rows), all * x_mask
__global__ void cf_bwd_head_kernel(const float* __restrict__ d_out, const float* __restrict__ d_logdet, float* __restrict__ dm, float* __restrict__ d_ld, int64_t rows, int T,
                                   const int32_t* __restrict__ lens) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
    const bool live = t < lens[b];
    dm[r] = live ? d_out[2 * r] : 0.f; dm[rows + r] = live ? d_out[2 * r + 1] : 0.f;
    d_ld[r] = live ? d_logdet[b] : 0.f;
}
// dhp (rows, NPp) = [dhs (rows, NP) * x_mask | 0]
__global__ void cf_pad_mask_kernel(const float* __restrict__ dhs, float* __restrict__ dhp, int64_t rows, int NPp, int NP, int T, const int32_t* __restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * NPp) return;
    const int64_t r = i / NPp; const int c = (int)(i - r * NPp);
    const int b = (int)(r / T), t = (int)(r - (int64_t)b * T);
    dhp[i] = (c < NP && t < lens[b]) ? dhs[r * NP + c] : 0.f;
}
}  // namespace
#define CF_LAUNCH(kernel, n, ...) do { hipLaunchKernelGGL(kernel, dim3((unsigned)xva_cdiv((n), 256)), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); XVA_LAUNCH_CHECK(); return XVA_OK; } while (0)
extern "C" int xva_cf_pre_fwd(const float* z, const float* pre_w, const float* pre_b, float* x0, float* x1, float* h, int64_t rows, int H, void* stream) {
    XVA_CHECK_ARG(z && pre_w && pre_b && x0 && x1 && h && rows > 0 && H > 0, "cf_pre_fwd: bad arguments");
    CF_LAUNCH(cf_pre_fwd_kernel, rows * H, z, pre_w, pre_b, x0, x1, h, rows, H);
}
extern "C" int xva_cf_mask_slice(const float* hp, float* hs, int B, int T, int NPp, int NP, const int32_t* lens, void* stream) {
    XVA_CHECK_ARG(hp && hs && lens && B > 0 && T > 0 && NP > 0 && NPp >= NP, "cf_mask_slice: bad arguments");
    CF_LAUNCH(cf_mask_slice_kernel, (int64_t)B * T * NP, hp, hs, (int64_t)B * T, NPp, NP, T, lens);
}
extern "C" int xva_cf_post_fwd(const float* x0, const float* y1, float* ld, float* out, float* ldsum, int B, int T, const int32_t* lens, void* stream) {
    XVA_CHECK_ARG(x0 && y1 && ld && out && ldsum && lens && B > 0 && T > 0, "cf_post_fwd: bad arguments");
    hipLaunchKernelGGL(cf_post_fwd_kernel, dim3((unsigned)B), dim3(128), 0, (hipStream_t)stream, x0, y1, ld, out, ldsum, T, lens);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_cf_bwd_head(const float* d_out, const float* d_logdet, float* dm, float* d_ld, int B, int T, const int32_t* lens, void* stream) {
    XVA_CHECK_ARG(d_out && d_logdet && dm && d_ld && lens && B > 0 && T > 0, "cf_bwd_head: bad arguments");
    CF_LAUNCH(cf_bwd_head_kernel, (int64_t)B * T, d_out, d_logdet, dm, d_ld, (int64_t)B * T, T, lens);
}
extern "C" int xva_cf_pad_mask(const float* dhs, float* dhp, int B, int T, int NPp, int NP, const int32_t* lens, void* stream) {
    XVA_CHECK_ARG(dhs && dhp && lens && B > 0 && T > 0 && NP > 0 && NPp >= NP, "cf_pad_mask: bad arguments");
    CF_LAUNCH(cf_pad_mask_kernel, (int64_t)B * T * NPp, dhs, dhp, (int64_t)B * T, NPp, NP, T, lens);
}
