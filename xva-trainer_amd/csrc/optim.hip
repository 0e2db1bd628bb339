// optim.hip — fused multi-tensor optimizers over flat parameter buffers (gfx950, HBM-bound).
//
// LAMB: python/fastpitch1_1/lamb.py:40-106 (no bias correction, weight-norm clamp 10, trust ratio 1 when a
// norm is 0), preceded by torch.nn.utils.clip_grad_norm_(params, 1000) (xva_train.py:857,861).  The reference
// runs ~10 ATen kernels per tensor x 181 tensors from Python; here one step is 3 launches over the flat
// buffers: (1) global grad-norm, (2) moments + per-tensor norms, (3) trust-ratio update.  Every element is
// read/written once per pass: 16 B read + 8 B written (pass 2), 16 B read + 4 B written (pass 3) per parameter.
#include "xva_common.h"
#include "../../include/xva_hip.h"

#define OPT_CHUNK 4096
#define OPT_THREADS 256

__global__ void sumsq_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ out) {
    __shared__ float sh[16];
    float a = 0.f;
    int64_t n4 = n / 4;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {            // four 16-byte loads in flight per thread
        float4 v0 = x4[i], v1 = x4[i + stride], v2 = x4[i + 2 * stride], v3 = x4[i + 3 * stride];
        a += v0.x * v0.x + v0.y * v0.y + v0.z * v0.z + v0.w * v0.w;
        a += v1.x * v1.x + v1.y * v1.y + v1.z * v1.z + v1.w * v1.w;
        a += v2.x * v2.x + v2.y * v2.y + v2.z * v2.z + v2.w * v2.w;
        a += v3.x * v3.x + v3.y * v3.y + v3.z * v3.z + v3.w * v3.w;
    }
    for (; i < n4; i += stride) {
        float4 v = x4[i];
        a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) for (int64_t j = n4 * 4 + threadIdx.x; j < n; j += blockDim.x) a += x[j] * x[j];
    a = xva_block_sum(a, sh);
    if (threadIdx.x == 0) atomicAdd(out, a);
}

// scal[0] = sum g^2 (all ranks' reduced grads), scal[1] <- clip coefficient
__global__ void clip_coef_kernel(float* scal, float max_norm, float inv_scale) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        float norm = sqrtf(scal[0]) * inv_scale;
        float coef = max_norm / (norm + 1e-6f);
        scal[1] = (coef < 1.f ? coef : 1.f) * inv_scale;
        scal[2] = norm;
        // a non-finite gradient (an overflow of the loss-scaled fp16 gradient buffers, a diverged batch): the step is SKIPPED on the device — moments
        // and parameters untouched — as torch.cuda.amp.GradScaler.step does for the reference's fp16 path (xva_train.py:856-859); the host reads the
        // flag when it reads the norm and backs the loss scale off
        scal[3] = (norm == norm && norm <= 3.0e38f) ? 0.f : 1.f;
    }
}

// chunk c covers elements [cstart[c], cstart[c] + clen[c]) of tensor ctid[c].  Tensors start on 16-byte boundaries and chunks on multiples of
// OPT_CHUNK inside them, so a chunk is read as float4 vectors — all OPT_CHUNK / (4 * OPT_THREADS) = 4 per thread and operand issued before the
// first use — plus a scalar tail (the last chunk of a tensor whose length is not a multiple of 4).
#define OPT_V (OPT_CHUNK / (4 * OPT_THREADS))
__global__ __launch_bounds__(OPT_THREADS) void lamb_pass1_kernel(const float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, const int32_t* __restrict__ ctid, const int64_t* __restrict__ cstart,
                                  const int32_t* __restrict__ clen, const float* __restrict__ scal, float* __restrict__ norms,
                                  float b1, float b2, float eps, float wd) {
    __shared__ float sh[16];
    int c = blockIdx.x;
    int tid = ctid[c];
    int64_t s = cstart[c];
    int len = clen[c];
    if (scal[3] != 0.f) return;      // skipped step (uniform)
    float gs = scal[1];
    float wn = 0.f, un = 0.f;
    const int len4 = len / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g + s); const float4* p4 = reinterpret_cast<const float4*>(p + s);
    float4* m4 = reinterpret_cast<float4*>(m + s); float4* v4 = reinterpret_cast<float4*>(v + s);
    float4 G[OPT_V], P[OPT_V], M[OPT_V], V[OPT_V];
#pragma unroll
    for (int q = 0; q < OPT_V; ++q) {
        const int i = threadIdx.x + q * OPT_THREADS;
        if (i < len4) { G[q] = g4[i]; P[q] = p4[i]; M[q] = m4[i]; V[q] = v4[i]; }
    }
#pragma unroll
    for (int q = 0; q < OPT_V; ++q) {
        const int i = threadIdx.x + q * OPT_THREADS;
        if (i >= len4) continue;
        float gg[4] = {G[q].x * gs, G[q].y * gs, G[q].z * gs, G[q].w * gs}, pp[4] = {P[q].x, P[q].y, P[q].z, P[q].w};
        float mm[4] = {M[q].x, M[q].y, M[q].z, M[q].w}, vv[4] = {V[q].x, V[q].y, V[q].z, V[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mm[e] = b1 * mm[e] + (1.f - b1) * gg[e];
            vv[e] = b2 * vv[e] + (1.f - b2) * gg[e] * gg[e];
            float u = mm[e] / (sqrtf(vv[e]) + eps) + wd * pp[e];
            wn += pp[e] * pp[e]; un += u * u;
        }
        m4[i] = make_float4(mm[0], mm[1], mm[2], mm[3]); v4[i] = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
    for (int i = len4 * 4 + threadIdx.x; i < len; i += blockDim.x) {
        int64_t k = s + i;
        float gg = g[k] * gs, pp = p[k];
        float mm = b1 * m[k] + (1.f - b1) * gg;
        float vv = b2 * v[k] + (1.f - b2) * gg * gg;
        m[k] = mm; v[k] = vv;
        float u = mm / (sqrtf(vv) + eps) + wd * pp;
        wn += pp * pp; un += u * u;
    }
    wn = xva_block_sum(wn, sh);
    un = xva_block_sum(un, sh);
    if (threadIdx.x == 0) { atomicAdd(norms + 2 * tid, wn); atomicAdd(norms + 2 * tid + 1, un); }
}
__global__ __launch_bounds__(OPT_THREADS) void lamb_pass2_kernel(float* __restrict__ p, const float* __restrict__ m, const float* __restrict__ v,
                                  const int32_t* __restrict__ ctid, const int64_t* __restrict__ cstart,
                                  const int32_t* __restrict__ clen, const float* __restrict__ norms, const float* __restrict__ scal, float lr, float eps, float wd) {
    if (scal[3] != 0.f) return;      // skipped step (uniform)
    int c = blockIdx.x;
    int tid = ctid[c];
    int64_t s = cstart[c];
    int len = clen[c];
    float wn = fminf(sqrtf(norms[2 * tid]), 10.f), un = sqrtf(norms[2 * tid + 1]);
    float trust = (wn == 0.f || un == 0.f) ? 1.f : wn / un;
    float step = lr * trust;
    const int len4 = len / 4;
    float4* p4 = reinterpret_cast<float4*>(p + s);
    const float4* m4 = reinterpret_cast<const float4*>(m + s); const float4* v4 = reinterpret_cast<const float4*>(v + s);
    float4 P[OPT_V], M[OPT_V], V[OPT_V];
#pragma unroll
    for (int q = 0; q < OPT_V; ++q) {
        const int i = threadIdx.x + q * OPT_THREADS;
        if (i < len4) { P[q] = p4[i]; M[q] = m4[i]; V[q] = v4[i]; }
    }
#pragma unroll
    for (int q = 0; q < OPT_V; ++q) {
        const int i = threadIdx.x + q * OPT_THREADS;
        if (i >= len4) continue;
        float pp[4] = {P[q].x, P[q].y, P[q].z, P[q].w}, mm[4] = {M[q].x, M[q].y, M[q].z, M[q].w}, vv[4] = {V[q].x, V[q].y, V[q].z, V[q].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float u = mm[e] / (sqrtf(vv[e]) + eps) + wd * pp[e];
            pp[e] = pp[e] - step * u;
        }
        p4[i] = make_float4(pp[0], pp[1], pp[2], pp[3]);
    }
    for (int i = len4 * 4 + threadIdx.x; i < len; i += blockDim.x) {
        int64_t k = s + i;
        float pp = p[k];
        float u = m[k] / (sqrtf(v[k]) + eps) + wd * pp;
        p[k] = pp - step * u;
    }
}

extern "C" int xva_opt_chunk_size(void) { return OPT_CHUNK; }

// Host helper: expand (offset, numel) of the ACTIVE tensors into chunk descriptors. Returns the chunk count
// (also when cap is too small, so callers can size the arrays with a first call using cap = 0).
extern "C" int64_t xva_opt_build_chunks(const int64_t* offsets, const int64_t* numels, const int32_t* active, int n,
                                        int32_t* ctid, int64_t* cstart, int32_t* clen, int64_t cap) {
    int64_t k = 0;
    for (int t = 0; t < n; ++t) {
        if (active && !active[t]) continue;
        for (int64_t s = 0; s < numels[t]; s += OPT_CHUNK) {
            if (k < cap) {
                ctid[k] = t; cstart[k] = offsets[t] + s;
                int64_t l = numels[t] - s; clen[k] = (int32_t)(l < OPT_CHUNK ? l : OPT_CHUNK);
            }
            ++k;
        }
    }
    return k;
}

// scal: >= 4 floats of device scratch; on return scal[2] = pre-clip global grad norm, scal[1] = applied coefficient, scal[3] = 1 if the gradient was
// non-finite and the step was skipped (nothing written), else 0.
// norms: 2 * n_tensors floats of device scratch (per-tensor ||p||^2, ||u||^2; sqrt/clamp applied on use).
// inv_scale: 1 / (loss scale * world averaging), folded into the gradient read (GradScaler.unscale_).
// (Measured and not kept: running the two passes group by group over the chunk list so that a group's update pass finds its parameters and moments in
// the 256 MB Infinity Cache — 4 / 6 / 7 / 9 / 13 groups: 439 / 467 / 488 / 543 / 590 us against 409 us for one pass each over everything.)
extern "C" int xva_lamb_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t total_floats,
                             const int32_t* ctid, const int64_t* cstart, const int32_t* clen, int64_t n_chunks, int n_tensors,
                             float* scal, float* norms, float lr, float beta1, float beta2, float eps, float weight_decay,
                             float max_grad_norm, float inv_scale, void* stream) {
    XVA_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && ctid && cstart && clen && scal && norms, "lamb_step: null");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scal, 0, 4 * sizeof(float), st) != hipSuccess || hipMemsetAsync(norms, 0, 2 * n_tensors * sizeof(float), st) != hipSuccess) {
        xva_set_error("lamb_step: memset failed");
        return XVA_ERR_HIP;
    }
    int grid = (int)((total_floats / 4 + OPT_THREADS - 1) / OPT_THREADS);
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(OPT_THREADS), 0, st, grads, total_floats, scal);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, st, scal, max_grad_norm, inv_scale);
    if (n_chunks > 0) {
        hipLaunchKernelGGL(lamb_pass1_kernel, dim3((unsigned)n_chunks), dim3(OPT_THREADS), 0, st, params, grads, exp_avg, exp_avg_sq,
                           ctid, cstart, clen, scal, norms, beta1, beta2, eps, weight_decay);
        hipLaunchKernelGGL(lamb_pass2_kernel, dim3((unsigned)n_chunks), dim3(OPT_THREADS), 0, st, params, exp_avg, exp_avg_sq, ctid,
                           cstart, clen, norms, scal, lr, eps, weight_decay);
    }
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- AdamW (torch.optim.AdamW, hifigan/xva_train.py:298-300): decoupled weight decay, bias correction ----
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                             int64_t n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float pp = p[i] * (1.f - lr * wd);
        float gg = g[i];
        float mm = b1 * m[i] + (1.f - b1) * gg;
        float vv = b2 * v[i] + (1.f - b2) * gg * gg;
        m[i] = mm; v[i] = vv;
        float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = pp - (lr / bc1) * (mm / denom);
    }
}
extern "C" int xva_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                              float lr, float beta1, float beta2, float eps, float weight_decay, void* stream) {
    XVA_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && step >= 1, "adamw_step: bad args");
    float bc1 = 1.f - powf(beta1, (float)step);
    float bc2 = 1.f - powf(beta2, (float)step);
    int grid = (int)((n + OPT_THREADS - 1) / OPT_THREADS);
    if (grid > 4096) grid = 4096;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3(grid), dim3(OPT_THREADS), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n, lr,
                       beta1, beta2, eps, weight_decay, bc1, sqrtf(bc2));
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
