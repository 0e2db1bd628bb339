// data_ops.hip — on-device batch preparation: what the reference does on the CPU inside DataLoader workers.
//
// Reference:
//   TTSCollate.__call__     python/fastpitch1_1/fastpitch/data_function.py:565-695  (sort by text length, zero-pad text / mel /
//                           pitch / energy / durations / attention prior)
//   TTSDataset.__getitem__  :297-352 (energy = ||mel||_2 over channels :327), get_mel :385-429 (int16 / 32768 -> TacotronSTFT)
//   beta_binomial_prior_distribution :84-94 (scipy.stats.betabinom pmf rows)
//   MelDataset.__getitem__  python/hifigan/meldataset.py:340-373 (int16 / 32768 -> peak normalise * 0.95 -> random crop / zero pad)
//
// The host hands over RAGGED data exactly as it comes off disk — int16 clips, symbol ids, cached pitch / durations,
// concatenated into flat buffers with an offsets array — and these kernels build the padded, sorted batch tensors the
// engines consume.  All of it is HBM-bound integer / byte work: one coalesced pass, one workgroup row per item.
#include "xva_common.h"
#include "../../include/xva_hip.h"
#include <math.h>

// order[r] = index of the item with the r-th LONGEST length; ties keep ascending item order (a stable descending sort;
// torch.sort(descending=True) in TTSCollate leaves tie order unspecified).  B <= 1024: one workgroup, O(B^2) compares.
__global__ void data_rank_desc_kernel(const int32_t* __restrict__ lens, int B, int32_t* __restrict__ order) {
    for (int i = threadIdx.x; i < B; i += blockDim.x) {
        const int li = lens[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
            const int lj = lens[j];
            rank += (lj > li) || (lj == li && j < i);
        }
        order[rank] = i;
    }
}
extern "C" int xva_data_rank_desc(const int32_t* lens, int B, int32_t* order, void* stream) {
    XVA_CHECK_ARG(lens && order && B > 0 && B <= 4096, "data_rank_desc: bad arguments (B=%d)", B);
    hipLaunchKernelGGL(data_rank_desc_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, lens, B, order);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

struct TruncF {                      // fp32 storage of a value that went through a LongTensor
    float v;
    __device__ TruncF(float x) : v(truncf(x)) {}
    __device__ TruncF(int x) : v((float)x) {}
};
// dst[r][c][j] = j < lens[i] ? src[offsets[i] * inner + c * lens[i] + j] : 0   with i = order[r] (or r)
// Items are (inner, len) row-major slabs of a flat buffer (pitch is (n_formants, T); ids / durations have inner = 1).
template <typename S, typename D>
__global__ void data_pad_gather_kernel(const S* __restrict__ flat, const int64_t* __restrict__ offsets, const int32_t* __restrict__ lens,
                                       const int32_t* __restrict__ order, D* __restrict__ dst, int inner, int max_len, int32_t* __restrict__ lens_out) {
    const int r = blockIdx.y, i = order ? order[r] : r;
    const int len = lens[i];
    const int64_t base = offsets[i] * inner;
    if (lens_out && blockIdx.x == 0 && threadIdx.x == 0) lens_out[r] = len < max_len ? len : max_len;
    const int64_t n = (int64_t)inner * max_len;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(e / max_len), j = (int)(e - (int64_t)c * max_len);
        dst[(int64_t)r * n + e] = j < len ? (D)flat[base + (int64_t)c * len + j] : (D)0;
    }
}
extern "C" int xva_data_pad_gather(const void* flat, int src_dtype, const int64_t* offsets, const int32_t* lens, const int32_t* order, void* dst,
                                   int dst_dtype, int B, int inner, int max_len, int32_t* lens_out, void* stream) {
    XVA_CHECK_ARG(flat && offsets && lens && dst && B > 0 && inner > 0 && max_len > 0, "data_pad_gather: bad arguments");
    const int64_t n = (int64_t)inner * max_len;
    dim3 grid((unsigned)(n / 256 < 1 ? 1 : (n / 256 > 64 ? 64 : n / 256)), B), block(256);
    hipStream_t st = (hipStream_t)stream;
#define XVA_PG(S, D) hipLaunchKernelGGL((data_pad_gather_kernel<S, D>), grid, block, 0, st, (const S*)flat, offsets, lens, order, (D*)dst, inner, max_len, lens_out)
    if (src_dtype == XVA_DATA_I32 && dst_dtype == XVA_DATA_I32) XVA_PG(int32_t, int32_t);
    else if (src_dtype == XVA_DATA_I64 && dst_dtype == XVA_DATA_I32) XVA_PG(int64_t, int32_t);
    else if (src_dtype == XVA_DATA_F32 && dst_dtype == XVA_DATA_F32) XVA_PG(float, float);
    else if (src_dtype == XVA_DATA_F64 && dst_dtype == XVA_DATA_F32) XVA_PG(double, float);
    else if (src_dtype == XVA_DATA_I16 && dst_dtype == XVA_DATA_F32) XVA_PG(int16_t, float);
    else if (src_dtype == XVA_DATA_I64 && dst_dtype == XVA_DATA_F32) XVA_PG(int64_t, float);
    else if (src_dtype == XVA_DATA_F32 && dst_dtype == XVA_DATA_I32) XVA_PG(float, int32_t);     // durations: float .npy into a LongTensor (truncation)
    else if (src_dtype == XVA_DATA_F32 && dst_dtype == XVA_DATA_F32_TRUNC) XVA_PG(float, TruncF);  // pitch: same LongTensor quirk, kept as fp32
    else { xva_set_error("data_pad_gather: unsupported dtype pair %d -> %d", src_dtype, dst_dtype); return XVA_ERR_ARG; }
#undef XVA_PG
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- HiFi-GAN item preparation -----------------------------------------------------------------------------------------------
// peak[i] = max |x| over the whole clip (int16 magnitudes: exact)
__global__ void wav_peak_kernel(const int16_t* __restrict__ flat, const int64_t* __restrict__ offsets, const int32_t* __restrict__ lens,
                                int32_t* __restrict__ peak) {
    __shared__ float sh[16];
    const int i = blockIdx.x;
    const int16_t* x = flat + offsets[i];
    const int n = lens[i];
    int m = 0;
    for (int j = threadIdx.x; j < n; j += blockDim.x) { int v = x[j]; v = v < 0 ? -v : v; m = v > m ? v : m; }
    const float r = xva_block_max((float)m, sh);     // |int16| <= 32768: exact in fp32
    if (threadIdx.x == 0) peak[i] = (int)r;
}
// out[r][t] = float( (x / 32768) / (peak / 32768) * gain ) for t inside the crop, 0 in the zero padding.  The arithmetic is the
// reference's, in the reference's precision: numpy float64 (`audio / MAX_WAV_VALUE`, librosa.util.normalize, `* 0.95`) rounded
// to fp32 by torch.FloatTensor (meldataset.py:347-351).  An all-zero clip stays zero (librosa leaves norms below `tiny` alone).
__global__ void wav_crop_norm_kernel(const int16_t* __restrict__ flat, const int64_t* __restrict__ offsets, const int32_t* __restrict__ lens,
                                     const int32_t* __restrict__ starts, const int32_t* __restrict__ peak, float* __restrict__ out, int seg,
                                     double gain, int normalize) {
    const int r = blockIdx.y;
    const int16_t* x = flat + offsets[r];
    const int n = lens[r], s0 = starts ? starts[r] : 0;
    const double a_peak = (double)peak[r] / 32768.0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < seg; t += gridDim.x * blockDim.x) {
        const int src = s0 + t;
        float v = 0.f;
        if (src < n) {
            double a = (double)x[src] / 32768.0;
            if (normalize) { if (peak[r] > 0) a = a / a_peak; a = a * gain; }
            v = (float)a;
        }
        out[(int64_t)r * seg + t] = v;
    }
}
extern "C" int xva_wav_peak_i16(const int16_t* flat, const int64_t* offsets, const int32_t* lens, int B, int32_t* peak, void* stream) {
    XVA_CHECK_ARG(flat && offsets && lens && peak && B > 0, "wav_peak_i16: bad arguments");
    hipLaunchKernelGGL(wav_peak_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, flat, offsets, lens, peak);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
extern "C" int xva_wav_crop_norm(const int16_t* flat, const int64_t* offsets, const int32_t* lens, const int32_t* starts, const int32_t* peak,
                                 float* out, int B, int seg, double gain, int normalize, void* stream) {
    XVA_CHECK_ARG(flat && offsets && lens && peak && out && B > 0 && seg > 0, "wav_crop_norm: bad arguments");
    dim3 grid((unsigned)xva_cdiv(seg, 1024) > 32 ? 32 : xva_cdiv(seg, 1024), B);
    hipLaunchKernelGGL(wav_crop_norm_kernel, grid, dim3(256), 0, (hipStream_t)stream, flat, offsets, lens, starts, peak, out, seg, gain, normalize);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- beta-binomial attention prior (training stage 1) ------------------------------------------------------------------------
// out[r][m][k] = betabinom(n = P, a = m + 1, b = M - m).pmf(k) for m < M = mel_lens[r], k < P = text_lens[r]; 0 elsewhere.
// (data_function.py:84-94: for i in 1..M: a = i, b = M + 1 - i, x = arange(P), n = P — the last value k = P is never taken.)
//   log pmf = lgamma(n+1) - lgamma(k+1) - lgamma(n-k+1) + lbeta(k + a, n - k + b) - lbeta(a, b)    in fp64, rounded to fp32
__device__ __forceinline__ double xva_lbeta(double a, double b) { return lgamma(a) + lgamma(b) - lgamma(a + b); }
__global__ void betabinom_prior_kernel(const int32_t* __restrict__ text_lens, const int32_t* __restrict__ mel_lens, float* __restrict__ out,
                                       int Tm, int Tt) {
    const int r = blockIdx.z, m = blockIdx.y;
    const int P = text_lens[r], M = mel_lens[r];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < Tt; k += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (m < M && k < P) {
            const double n = P, a = m + 1, b = M - m;
            const double lp = lgamma(n + 1) - lgamma((double)k + 1) - lgamma(n - k + 1) + xva_lbeta(k + a, n - k + b) - xva_lbeta(a, b);
            v = (float)exp(lp);
        }
        out[((int64_t)r * Tm + m) * Tt + k] = v;
    }
}
extern "C" int xva_data_betabinom_prior(const int32_t* text_lens, const int32_t* mel_lens, float* out, int B, int Tm, int Tt, void* stream) {
    XVA_CHECK_ARG(text_lens && mel_lens && out && B > 0 && Tm > 0 && Tt > 0, "betabinom_prior: bad arguments");
    hipLaunchKernelGGL(betabinom_prior_kernel, dim3(xva_cdiv(Tt, 64), Tm, B), dim3(64), 0, (hipStream_t)stream, text_lens, mel_lens, out, Tm, Tt);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ---- ragged mel: finishing pass -------------------------------------------------------------------------------------------------
// mel (B, n_mel, T): frames t >= n_frames[b] are zeroed (TTSCollate's zero padding); energy[b][t] = ||mel[b, :, t]||_2 for live
// frames (data_function.py:327), 0 beyond.  energy_trunc: TTSCollate accumulates pitch and energy into tensors created with the TEXT's
// dtype (`dtype=batch[0][0].dtype`, `zeros_like(pitch_padded[:, 0, :])`, data_function.py:594-606), i.e. LongTensors: the values a
// reference batch carries are truncated toward zero before batch_to_gpu's .float().  Set for bit-parity with the reference batch.
__global__ void mel_finish_ragged_kernel(float* __restrict__ mel, const int32_t* __restrict__ n_frames, float* __restrict__ energy, int n_mel, int T,
                                         int energy_trunc) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    float* m = mel + (int64_t)b * n_mel * T + t;
    if (t >= n_frames[b]) {
        for (int c = 0; c < n_mel; ++c) m[(int64_t)c * T] = 0.f;
        if (energy) energy[(int64_t)b * T + t] = 0.f;
        return;
    }
    if (energy) {
        float acc = 0.f;
        for (int c = 0; c < n_mel; ++c) { const float v = m[(int64_t)c * T]; acc += v * v; }
        const float e = sqrtf(acc);
        energy[(int64_t)b * T + t] = energy_trunc ? truncf(e) : e;
    }
}
extern "C" int xva_mel_finish_ragged(float* mel, const int32_t* n_frames, float* energy, int B, int n_mel, int T, int energy_trunc, void* stream) {
    XVA_CHECK_ARG(mel && n_frames && B > 0 && n_mel > 0 && T > 0, "mel_finish_ragged: bad arguments");
    hipLaunchKernelGGL(mel_finish_ragged_kernel, dim3(xva_cdiv(T, 64), B), dim3(64), 0, (hipStream_t)stream, mel, n_frames, energy, n_mel, T, energy_trunc);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
