// align_ops.hip — the non-GEMM kernels of FastPitch training stage 1 (the aligner): ConvAttention's distance / log-softmax / prior
// / softmax rows (attention.py:196-219), monotonic alignment search (alignment.py:76-104), the forward-sum (CTC) loss and its
// gradient (attn_loss_function.py:20-44, torch.nn.CTCLoss), and the backward of the first log-softmax.  fp32 throughout: the
// aligner is a few MFLOP per frame and its loss is a log-domain dynamic programme.
//
// Layouts: query / key encodings are padded token-major (B, T + 2, 80) like every other sequence; attention maps are
// (B, Tm, ld) fp32 with ld = Tt rounded up to 4 (GEMM leading-dimension rule), columns >= Tt never read.
#include "xva_common.h"
#include "../../include/xva_gemm.h"
#include "../../include/xva_hip.h"

namespace {
__device__ __forceinline__ float lse2f(float a, float b) {
    if (a == -INFINITY) return b;
    if (b == -INFINITY) return a;
    const float m = fmaxf(a, b);
    return m + __logf(__expf(a - m) + __expf(b - m));
}
__device__ __forceinline__ float lse3f(float a, float b, float c) { return lse2f(lse2f(a, b), c); }
}  // namespace

// text embedding without positional term: out (B, Tt+2, C), structural rows zero (model.py:300: encoder.word_emb(inputs))
__global__ void al_embed_kernel(const int* __restrict__ ids, const float* __restrict__ emb, float* __restrict__ out, int T, int C) {
    const int Tp = T + 2;
    const int64_t r = blockIdx.x;
    const int b = (int)(r / Tp), tp = (int)(r % Tp);
    const bool live = tp >= 1 && tp <= T;
    const float* e = emb + (int64_t)(live ? ids[b * T + tp - 1] : 0) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[r * C + c] = live ? e[c] : 0.f;
}
// mel (B, C, Tm) -> padded time-major (B, Tm+2, C)
__global__ void al_mel_to_tm_kernel(const float* __restrict__ mel, float* __restrict__ out, int C, int Tm) {
    const int Tp = Tm + 2;
    const int64_t r = blockIdx.x;
    const int b = (int)(r / Tp), tp = (int)(r % Tp);
    const bool live = tp >= 1 && tp <= Tm;
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[r * C + c] = live ? mel[((int64_t)b * C + c) * Tm + tp - 1] : 0.f;
}
// out[r] = sum_c X[r][c]^2 (one wave per row)
__global__ void al_sqnorm_kernel(const float* __restrict__ X, float* __restrict__ out, int64_t rows, int C) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float a = 0.f;
    for (int c = threadIdx.x & 63; c < C; c += 64) { const float v = X[r * C + c]; a += v * v; }
    a = xva_wave_sum(a);
    if ((threadIdx.x & 63) == 0) out[r] = a;
}

// One (b, t1) row of ConvAttention.forward (attention.py:196-219):
//   s[t2]   = S[t1][t2] - 0.0005 (|q|^2 + |k|^2)        (S = 0.001 q.k from the GEMM)  = -0.0005 ||q - k||^2
//   logprob = log_softmax(s over ALL Tt keys) + log(prior + 1e-8)
//   soft    = softmax(logprob with padded keys at -inf)
// plus lse1 (of s) for the backward and lse2 = logsumexp(blank = -1, logprob[0..L-1]) for the CTC loss.
__global__ void al_attn_rows_kernel(const float* __restrict__ S, const float* __restrict__ qn, const float* __restrict__ kn,
                                    const float* __restrict__ prior, const int* __restrict__ in_lens, float* __restrict__ logprob,
                                    float* __restrict__ soft, float* __restrict__ lse1, float* __restrict__ lse2, int Tm, int Tt, int ld) {
    __shared__ float sh[16];
    const int b = blockIdx.y, t1 = blockIdx.x;
    const int L = in_lens[b];
    const float* Sr = S + ((int64_t)b * Tm + t1) * ld;
    const float* pr = prior + ((int64_t)b * Tm + t1) * Tt;
    float* lr = logprob + ((int64_t)b * Tm + t1) * ld;
    float* sr = soft + ((int64_t)b * Tm + t1) * ld;
    const float q2 = qn[(int64_t)b * (Tm + 2) + t1 + 1];
    const float* k2 = kn + (int64_t)b * (Tt + 2) + 1;
    float m = -INFINITY;
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) m = fmaxf(m, Sr[j] - 0.0005f * (q2 + k2[j]));
    m = xva_block_max(m, sh);
    float z = 0.f;
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) z += __expf(Sr[j] - 0.0005f * (q2 + k2[j]) - m);
    z = xva_block_sum(z, sh);
    const float l1 = m + __logf(z);
    float m2 = -INFINITY;
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) {
        const float lp = Sr[j] - 0.0005f * (q2 + k2[j]) - l1 + __logf(pr[j] + 1e-8f);
        lr[j] = lp;
        if (j < L) m2 = fmaxf(m2, lp);
    }
    m2 = xva_block_max(m2, sh);
    float z2 = 0.f;
    for (int j = threadIdx.x; j < L; j += blockDim.x) z2 += __expf(lr[j] - m2);
    z2 = xva_block_sum(z2, sh);
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) sr[j] = (j < L && z2 > 0.f) ? __expf(lr[j] - m2) / z2 : 0.f;
    if (threadIdx.x == 0) {
        lse1[(int64_t)b * Tm + t1] = l1;
        const float mm = fmaxf(m2, -1.f);
        lse2[(int64_t)b * Tm + t1] = mm + __logf(z2 * __expf(m2 - mm) + __expf(-1.f - mm));
    }
}

// Monotonic alignment search (alignment.py:76-104, mas_width1) of one item per workgroup over soft[b][:M][:L]; writes the token
// durations (column sums of the hard alignment).  `choice` (B, Tm, Tt) bytes: 1 = came from j-1.
__global__ void al_mas_kernel(const float* __restrict__ soft, const int* __restrict__ in_lens, const int* __restrict__ mel_lens,
                              uint8_t* __restrict__ choice, int* __restrict__ durs, int Tm, int Tt, int ld) {
    extern __shared__ float rows[];   // two rows of Tt log-probabilities
    const int b = blockIdx.x;
    const int L = in_lens[b], M = mel_lens[b];
    const float* A = soft + (int64_t)b * Tm * ld;
    uint8_t* ch = choice + (int64_t)b * Tm * Tt;
    float* prev = rows;
    float* cur = rows + Tt;
    for (int j = threadIdx.x; j < L; j += blockDim.x) prev[j] = j == 0 ? __logf(A[0]) : -INFINITY;
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) durs[b * Tt + j] = 0;
    __syncthreads();
    for (int i = 1; i < M; ++i) {
        for (int j = threadIdx.x; j < L; j += blockDim.x) {
            float best = prev[j];
            uint8_t diag = 0;
            if (j >= 1 && prev[j - 1] >= best) { best = prev[j - 1]; diag = 1; }
            cur[j] = __logf(A[(int64_t)i * ld + j]) + best;
            ch[(int64_t)i * Tt + j] = diag;
        }
        __syncthreads();
        float* t = prev; prev = cur; cur = t;
    }
    if (threadIdx.x == 0 && M > 0 && L > 0) {
        int c = L - 1;
        for (int i = M - 1; i >= 1; --i) { durs[b * Tt + c] += 1; c -= ch[(int64_t)i * Tt + c]; }
        durs[b * Tt + c] += 1;               // row 0 at the path's column ...
        if (c != 0) durs[b * Tt + 0] += 1;   // ... and (alignment.py:103) opt[0, prev_ind[0, c]] = opt[0, 0]
    }
}

// Forward-sum loss of one item per workgroup: CTC (blank = class 0 at log-prob -1 before normalisation, targets 1..L) over
// lp2[t][c] = x[t][c] - lse2[t], x = [-1, logprob[t][0..L-1]].  alpha / beta: (B, Tm, 2*Tt+1) scratch.
//   loss += nll / L / B ; dlogprob[t][k] = gscale / (L B) * (softmax(x)[t][k+1] - occupancy[t][k+1])   (0 outside t < M, k < L)
__global__ void al_ctc_kernel(const float* __restrict__ logprob, const float* __restrict__ lse2, const int* __restrict__ in_lens,
                              const int* __restrict__ mel_lens, float* __restrict__ alpha, float* __restrict__ beta,
                              float* __restrict__ dlogprob, float* __restrict__ loss, int B, int Tm, int Tt, int ld, float gscale) {
    const int b = blockIdx.x;
    const int L = in_lens[b], M = mel_lens[b];
    const int S = 2 * L + 1, Smax = 2 * Tt + 1;
    const float* lp = logprob + (int64_t)b * Tm * ld;
    const float* l2 = lse2 + (int64_t)b * Tm;
    float* al = alpha + (int64_t)b * Tm * Smax;
    float* be = beta + (int64_t)b * Tm * Smax;
    float* dl = dlogprob + (int64_t)b * Tm * ld;
    auto logp = [&](int t, int s) { return ((s & 1) ? lp[(int64_t)t * ld + (s >> 1)] : -1.f) - l2[t]; };
    for (int64_t i = threadIdx.x; i < (int64_t)Tm * ld; i += blockDim.x) dl[i] = 0.f;
    __syncthreads();
    if (L <= 0 || M <= 0) return;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        al[s] = s < 2 ? logp(0, s) : -INFINITY;
        be[(int64_t)(M - 1) * Smax + s] = s >= S - 2 ? logp(M - 1, s) : -INFINITY;
    }
    __syncthreads();
    for (int t = 1; t < M; ++t) {
        const float* ap = al + (int64_t)(t - 1) * Smax;
        float* ac = al + (int64_t)t * Smax;
        const float* bn = be + (int64_t)(M - t) * Smax;       // beta at time M - t (next of tb)
        float* bc = be + (int64_t)(M - 1 - t) * Smax;
        const int tb = M - 1 - t;
        for (int s = threadIdx.x; s < S; s += blockDim.x) {
            float a = lse2f(ap[s], s >= 1 ? ap[s - 1] : -INFINITY);
            if ((s & 1) && s >= 3) a = lse2f(a, ap[s - 2]);                 // labels are 1..L, all distinct: the skip is always legal
            ac[s] = a + logp(t, s);
            float v = lse2f(bn[s], s + 1 < S ? bn[s + 1] : -INFINITY);
            if ((s & 1) && s + 2 < S) v = lse2f(v, bn[s + 2]);
            bc[s] = v + logp(tb, s);
        }
        __syncthreads();
    }
    const float* aT = al + (int64_t)(M - 1) * Smax;
    const float ll = lse2f(aT[S - 1], S >= 2 ? aT[S - 2] : -INFINITY);
    const float nll = -ll;
    if (!(nll < INFINITY) || nll != nll) return;    // zero_infinity = True: no loss, no gradient
    if (threadIdx.x == 0) atomicAdd(loss, nll / (float)L / (float)B);
    const float c = gscale / ((float)L * (float)B);
    for (int64_t i = threadIdx.x; i < (int64_t)M * L; i += blockDim.x) {
        const int t = (int)(i / L), k = (int)(i % L);
        const int s = 2 * k + 1;
        const float lps = logp(t, s);
        const float occ = __expf(al[(int64_t)t * Smax + s] + be[(int64_t)t * Smax + s] - lps + nll);
        dl[(int64_t)t * ld + k] = c * (__expf(lps) - occ);
    }
}

// Backward of the first log-softmax, in place over dlogprob -> G = d(loss)/d s:  G = g - softmax(s) * sum(g); also colsum[b][t2] += G
__global__ void al_logsoftmax_bwd_kernel(const float* __restrict__ S, const float* __restrict__ qn, const float* __restrict__ kn,
                                         const float* __restrict__ lse1, float* __restrict__ G, float* __restrict__ colsum, int Tm, int Tt,
                                         int ld) {
    __shared__ float sh[16];
    const int b = blockIdx.y, t1 = blockIdx.x;
    const float* Sr = S + ((int64_t)b * Tm + t1) * ld;
    float* g = G + ((int64_t)b * Tm + t1) * ld;
    const float q2 = qn[(int64_t)b * (Tm + 2) + t1 + 1];
    const float* k2 = kn + (int64_t)b * (Tt + 2) + 1;
    const float l1 = lse1[(int64_t)b * Tm + t1];
    float sum = 0.f;
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) sum += g[j];
    sum = xva_block_sum(sum, sh);
    if (sum == 0.f) {   // rows without gradient (t1 >= mel length) stay zero
        bool any = false;
        for (int j = threadIdx.x; j < Tt; j += blockDim.x) any |= g[j] != 0.f;
        if (!__syncthreads_or(any)) return;
    }
    for (int j = threadIdx.x; j < Tt; j += blockDim.x) {
        const float p1 = __expf(Sr[j] - 0.0005f * (q2 + k2[j]) - l1);
        const float v = g[j] - p1 * sum;
        g[j] = v;
        if (v != 0.f) atomicAdd(colsum + (int64_t)b * Tt + j, v);
    }
}
// dk[b][t2][c] -= 0.001 * colsum[b][t2] * k[b][t2][c]   (the |k|^2 term of the distance; the |q|^2 term cancels: rows of G sum to 0)
__global__ void al_dk_fix_kernel(float* __restrict__ dk, const float* __restrict__ k, const float* __restrict__ colsum, int Tt, int C, float scale) {
    const int b = blockIdx.y, t2 = blockIdx.x;
    const int64_t r = (int64_t)b * (Tt + 2) + t2 + 1;
    const float cs = 0.001f * scale * colsum[(int64_t)b * Tt + t2];
    for (int c = threadIdx.x; c < C; c += blockDim.x) dk[r * C + c] -= cs * k[r * C + c];
}

#define AL_LAUNCH(kernel, grid, block, shmem, ...) do { hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)stream, __VA_ARGS__); XVA_LAUNCH_CHECK(); } while (0)

extern "C" int xva_al_embed(const int32_t* ids, const float* emb, float* out, int B, int T, int C, void* stream) {
    XVA_CHECK_ARG(ids && emb && out, "al_embed: null");
    AL_LAUNCH(al_embed_kernel, dim3(B * (T + 2)), dim3(128), 0, ids, emb, out, T, C);
    return XVA_OK;
}
extern "C" int xva_al_mel_to_tm(const float* mel, float* out, int B, int C, int Tm, void* stream) {
    XVA_CHECK_ARG(mel && out, "al_mel_to_tm: null");
    AL_LAUNCH(al_mel_to_tm_kernel, dim3(B * (Tm + 2)), dim3(128), 0, mel, out, C, Tm);
    return XVA_OK;
}
extern "C" int xva_al_sqnorm(const float* X, float* out, int64_t rows, int C, void* stream) {
    XVA_CHECK_ARG(X && out, "al_sqnorm: null");
    AL_LAUNCH(al_sqnorm_kernel, dim3((unsigned)xva_cdiv(rows, 4)), dim3(256), 0, X, out, rows, C);
    return XVA_OK;
}
extern "C" int xva_al_attn_rows(const float* S, const float* qn, const float* kn, const float* prior, const int32_t* in_lens, float* logprob,
                                float* soft, float* lse1, float* lse2, int B, int Tm, int Tt, int ld, void* stream) {
    XVA_CHECK_ARG(S && qn && kn && prior && in_lens && logprob && soft && lse1 && lse2, "al_attn_rows: null");
    AL_LAUNCH(al_attn_rows_kernel, dim3(Tm, B), dim3(256), 0, S, qn, kn, prior, in_lens, logprob, soft, lse1, lse2, Tm, Tt, ld);
    return XVA_OK;
}
extern "C" int xva_al_mas(const float* soft, const int32_t* in_lens, const int32_t* mel_lens, uint8_t* choice, int32_t* durs, int B, int Tm,
                          int Tt, int ld, void* stream) {
    XVA_CHECK_ARG(soft && in_lens && mel_lens && choice && durs, "al_mas: null");
    AL_LAUNCH(al_mas_kernel, dim3(B), dim3(256), (size_t)2 * Tt * sizeof(float), soft, in_lens, mel_lens, choice, durs, Tm, Tt, ld);
    return XVA_OK;
}
extern "C" int xva_al_ctc(const float* logprob, const float* lse2, const int32_t* in_lens, const int32_t* mel_lens, float* alpha, float* beta,
                          float* dlogprob, float* loss, int B, int Tm, int Tt, int ld, float gscale, void* stream) {
    XVA_CHECK_ARG(logprob && lse2 && in_lens && mel_lens && alpha && beta && dlogprob && loss, "al_ctc: null");
    AL_LAUNCH(al_ctc_kernel, dim3(B), dim3(512), 0, logprob, lse2, in_lens, mel_lens, alpha, beta, dlogprob, loss, B, Tm, Tt, ld, gscale);
    return XVA_OK;
}
extern "C" int xva_al_logsoftmax_bwd(const float* S, const float* qn, const float* kn, const float* lse1, float* G, float* colsum, int B, int Tm,
                                     int Tt, int ld, void* stream) {
    XVA_CHECK_ARG(S && qn && kn && lse1 && G && colsum, "al_logsoftmax_bwd: null");
    AL_LAUNCH(al_logsoftmax_bwd_kernel, dim3(Tm, B), dim3(256), 0, S, qn, kn, lse1, G, colsum, Tm, Tt, ld);
    return XVA_OK;
}
extern "C" int xva_al_dk_fix(float* dk, const float* k, const float* colsum, int B, int Tt, int C, float scale, void* stream) {
    XVA_CHECK_ARG(dk && k && colsum, "al_dk_fix: null");
    AL_LAUNCH(al_dk_fix_kernel, dim3(Tt, B), dim3(128), 0, dk, k, colsum, Tt, C, scale);
    return XVA_OK;
}
