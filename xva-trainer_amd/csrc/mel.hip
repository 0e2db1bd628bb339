// mel.hip — mel-spectrogram front end (STFT + mel filterbank + dynamic-range compression).
//
// Reference (three variants of one pipeline, see include/xva_hip.h for the knobs):
//   M1 TacotronSTFT.mel_spectrogram   python/fastpitch1_1/common/layers.py:121-138,
//      STFT.transform                 python/fastpitch1_1/common/stft.py:86-114,
//      dynamic_range_compression      python/fastpitch1_1/common/audio_processing.py:105-111
//   M2 mel_spectrogram                python/hifigan/meldataset.py:217-240
//   M3 TorchSTFT.__call__             python/xvapitch/audio.py:138-181
//
// MI355X mapping: frames are never materialised — the reflect-padded waveform is read with OVERLAPPING rows (row pitch = hop), each sample
// from HBM once, from L2 for its 4 overlapping frames.  The windowed DFT is a 1024-point real FFT, one wavefront per frame
// (xva_stft_fft1024_kernel below: HBM-bound on the spectrum it writes); n_fft != 1024 or xva_mel_set_dft(1) take the reference's own formulation,
// a GEMM against the windowed DFT basis on the exact-fp32 MFMA (bf16 would not hold the 1e-3 log-mel tolerance) — that GEMM is also what the
// differentiable mel's backward uses.  Pipeline:  reflect-pad -> DFT (re | im) -> magnitude -> mel GEMM with
// the log-clamp fused in its epilogue, written directly in the reference's (B, n_mel, T)
// layout (the mel GEMM is batched per clip with the filterbank as its A operand).
#include "xva_common.h"
#include "../../include/xva_gemm.h"
#include "../../include/xva_hip.h"

// y[b][i] = x[b][reflect(i - pad)]   (torch 'reflect': no edge repeat)
__global__ void xva_reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int N, int pad,
                                       int64_t ldx, int64_t ldy, int Np) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t total = (int64_t)B * ldy;
    if (idx >= total) return;
    int b = (int)(idx / ldy);
    int i = (int)(idx - (int64_t)b * ldy);
    float v = 0.f;
    if (i < Np) {
        int s = i - pad;
        if (s < 0) s = -s;
        if (s >= N) s = 2 * (N - 1) - s;
        v = x[(int64_t)b * ldx + s];
    }
    y[idx] = v;
}

// spec: rows of [re(0..nb-1) | im(0..nb-1)] with leading dim lds -> mag rows with leading
// dim ldm, columns >= nb zeroed (K padding for the mel GEMM).
__global__ void xva_magnitude_kernel(const float* __restrict__ spec, float* __restrict__ mag, int64_t rows, int nb,
                                     int64_t lds, int64_t ldm, float eps_add, float clamp_min) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * ldm) return;
    int64_t r = idx / ldm;
    int j = (int)(idx - r * ldm);
    float v = 0.f;
    if (j < nb) {
        float re = spec[r * lds + j], im = spec[r * lds + nb + j];
        float p = re * re + im * im + eps_add;
        if (clamp_min > 0.f) p = fmaxf(p, clamp_min);
        v = sqrtf(p);
    }
    mag[idx] = v;
}


// ---- the windowed DFT of every frame as a 1024-point real FFT (one wavefront per frame) ------------------------------------------------------
// The reference's STFT is torch.stft / a conv1d with a windowed DFT basis; as a dense GEMM it costs 2 * 1024 * 1026 flops per frame on the
// exact-fp32 MFMA (0.95 ms for FastPitch's 27 520-frame batch, the whole front end's time).  The same 513 bins from an FFT are ~30 kflop per
// frame and the kernel is bound by the spectrum it writes.  Per frame: z[n] = (w x)[2n] + i (w x)[2n + 1], n < 512; Z = FFT_512(z) as three
// radix-8 passes (512 = 8 * 8 * 8; lane l of the wave owns 8 points per pass, the two regroupings go through LDS); then
// X[k] = (Z[k] + conj Z[512 - k]) / 2 - i W_1024^k (Z[k] - conj Z[512 - k]) / 2, k = 0 .. 512.  The window is row 0 of the caller's basis
// (cos(0) * w[n]); twiddles W_1024^m are computed once per workgroup into LDS (sincospif: 1 ulp).  Output: the DFT GEMM's layout,
// spec[frame][re(0 .. nb - 1) | im(0 .. nb - 1)] with the GEMM's sign (im = -sum x w sin), so every consumer is unchanged.
#define FFT_WAVES 4
// complex numbers as native 2-vectors: the adds / subtracts / multiplies of the butterflies map onto the packed fp32 instructions
// (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: two floats per lane per issue) — the fused front-end kernel below is VALU-bound
typedef float cf __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cf cadd(cf a, cf b) { return a + b; }
__device__ __forceinline__ cf csub(cf a, cf b) { return a - b; }
__device__ __forceinline__ cf cmul(cf a, cf b) { const cf s = {-a.y, a.y}; return a.xx * b + s * b.yx; }   // (ax bx - ay by, ax by + ay bx)
__device__ __forceinline__ cf cmni(cf a) { return (cf){a.y, -a.x}; }             // a * (-i)
__device__ __forceinline__ void dft8(cf* a) {                                     // in place, natural order in and out, forward (e^{-2 pi i nk / 8})
    const float h = 0.70710678118654752440f;
    cf t0 = cadd(a[0], a[4]), t1 = csub(a[0], a[4]), t2 = cadd(a[2], a[6]), t3 = cmni(csub(a[2], a[6]));
    cf t4 = cadd(a[1], a[5]), t5 = csub(a[1], a[5]), t6 = cadd(a[3], a[7]), t7 = cmni(csub(a[3], a[7]));
    cf u0 = cadd(t0, t2), u2 = csub(t0, t2), u1 = cadd(t1, t3), u3 = csub(t1, t3);
    cf v0 = cadd(t4, t6), v2 = cmni(csub(t4, t6)), v1 = cadd(t5, t7), v3 = csub(t5, t7);
    v1 = {h * (v1.x + v1.y), h * (v1.y - v1.x)};                                  // * W8   = (1 - i) / sqrt 2
    v3 = {h * (v3.y - v3.x), -h * (v3.x + v3.y)};                                 // * W8^3 = (-1 - i) / sqrt 2
    a[0] = cadd(u0, v0); a[4] = csub(u0, v0); a[1] = cadd(u1, v1); a[5] = csub(u1, v1);
    a[2] = cadd(u2, v2); a[6] = csub(u2, v2); a[3] = cadd(u3, v3); a[7] = csub(u3, v3);
}
__global__ __launch_bounds__(64 * FFT_WAVES) void xva_stft_fft1024_kernel(const float* __restrict__ ypad, const float* __restrict__ window, float* __restrict__ spec,
                                                                          int B, int T, int hop, int64_t ldy, int64_t lds, int nb) {
    __shared__ cf tw[1024];                                                       // W_1024^m = e^{-2 pi i m / 1024}
    __shared__ cf buf[FFT_WAVES][584];                                            // per-wave exchange (8 x 72 / 64 x 9 padded) and the 513-point spectrum
    for (int m = threadIdx.x; m < 1024; m += 64 * FFT_WAVES) { float sn, cs; sincospif(-(float)m * (1.0f / 512.0f), &sn, &cs); tw[m] = {cs, sn}; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    cf* sb = buf[wave];
    const int n2 = lane >> 3, n3 = lane & 7;                                      // pass 1: lane = (n2, n3); pass 2: lane = (k1, n3) ; pass 3: lane = k1 + 8 k2
    float2 w2[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) w2[n1] = *reinterpret_cast<const float2*>(window + 2 * (64 * n1 + lane));
    __syncthreads();
    const int64_t nframes = (int64_t)B * T;
    const int64_t per = (int64_t)gridDim.x * FFT_WAVES;
    const int64_t iters = (nframes + per - 1) / per;
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t f = it * per + (int64_t)blockIdx.x * FFT_WAVES + wave;
        const bool live = f < nframes;
        const int b = live ? (int)(f / T) : 0, t = live ? (int)(f - (int64_t)b * T) : 0;
        const float* x = ypad + (int64_t)b * ldy + (int64_t)t * hop;
        cf a[8];
#pragma unroll
        for (int n1 = 0; n1 < 8; ++n1) {                                          // hop % 4 == 0 and ldy % 4 == 0: 8-byte aligned
            const float2 v = *reinterpret_cast<const float2*>(x + 2 * (64 * n1 + lane));
            a[n1] = {v.x * w2[n1].x, v.y * w2[n1].y};
        }
        dft8(a);                                                                  // over n1: A[k1; n2, n3]
#pragma unroll
        for (int k1 = 0; k1 < 8; ++k1) sb[72 * k1 + lane] = k1 ? cmul(a[k1], tw[(16 * n2 * k1) & 1023]) : a[0];     // * W_64^{n2 k1}
        __syncthreads();
        const int k1 = lane >> 3;
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = sb[72 * k1 + 8 * q + n3];              // over n2
        dft8(a);                                                                  // B[k1, k2; n3]
        __syncthreads();
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) sb[9 * (k1 + 8 * k2) + n3] = cmul(a[k2], tw[(2 * n3 * (k1 + 8 * k2)) & 1023]);   // * W_512^{n3 (k1 + 8 k2)}
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) a[q] = sb[9 * lane + q];                      // over n3
        dft8(a);                                                                  // Z[lane + 64 k3]
        __syncthreads();
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) sb[lane + 64 * k3] = a[k3];
        if (lane == 0) sb[512] = a[0];                                            // Z[512] = Z[0]
        __syncthreads();
        if (live) {
            float* row = spec + f * lds;
#pragma unroll
            for (int k3 = 0; k3 <= 8; ++k3) {
                const int k = lane + 64 * k3;
                if (k3 == 8 && lane != 0) break;
                const cf zk = sb[k], zc = sb[512 - k];
                const cf e = {0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y)};        // (Z[k] + conj Z[512 - k]) / 2
                const cf d = {0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y)};        // (Z[k] - conj Z[512 - k]) / 2
                const cf o = cmul(cmni(d), tw[k]);                                // -i W^k (.)
                if (k < nb) { row[k] = e.x + o.x; row[nb + k] = e.y + o.y; }
            }
        }
        __syncthreads();
    }
}
// ---- the whole forward front end in ONE kernel: reflect-indexed frame -> windowed FFT -> magnitude -> mel filterbank -> log ----------------
// The four-launch pipeline above writes and re-reads three intermediates per frame (padded clip, 1026-float spectrum, 544-float magnitude row:
// 12.6 KB against 1.3 KB of algorithmic traffic, 9.4 x); only the differentiable mel's backward needs them.  Here a frame's spectrum never leaves
// LDS: the wave that transformed it takes the 513 magnitudes and scatters them into the (at most two) triangular mel filters each bin belongs to
// — a per-bin tap table built on the device from the caller's dense filterbank by a one-block pre-kernel, with a dense fallback for any bin
// that has more than two non-zero weights — and writes log(max(., clamp)).  A workgroup (4 waves) walks 16 consecutive frames of one clip, so the
// reference's (B, n_mel, T) layout is written in 64-byte row segments.  Frames are read straight from the caller's clip (float rows, or the ragged
// int16 batch with / 32768) with torch's reflect index map applied per sample: no padded copy.
struct MelTap { int m0; float w0; int m1; float w1; };           // m < 0: unused ; m0 == -2: more than two filters touch this bin (dense fallback)
__global__ void xva_mel_taps_kernel(const float* __restrict__ mel_basis, int64_t ldm, int n_mel, int nb, MelTap* __restrict__ tab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nb) return;
    MelTap t = {-1, 0.f, -1, 0.f};
    int cnt = 0;
    for (int mb = 0; mb < n_mel; mb += 16) {                     // 16 independent loads in flight (a plain loop paid one L2 round trip per filter: 26 us)
        float w[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) w[q] = mb + q < n_mel ? mel_basis[(int64_t)(mb + q) * ldm + k] : 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (w[q] != 0.f) {
                if (cnt == 0) { t.m0 = mb + q; t.w0 = w[q]; } else if (cnt == 1) { t.m1 = mb + q; t.w1 = w[q]; }
                ++cnt;
            }
    }
    if (cnt > 2 || (cnt == 2 && t.m1 != t.m0 + 1)) t.m0 = -2;       // the fast path assumes adjacent triangles (filters m, m + 1)
    tab[k] = t;
}
#define MELF_FRAMES 16
#define MELF_MAXMEL 96
template <int SRC>      // 0: float clips, row stride ldx, N samples each ; 1: ragged int16 clips (flat + offsets[order[r]], lens) scaled by 1 / 32768
__global__ __launch_bounds__(64 * FFT_WAVES) void xva_mel_fused_kernel(const void* __restrict__ src, int64_t ldx, const int64_t* __restrict__ offsets,
                                                                       const int32_t* __restrict__ lens, const int32_t* __restrict__ order,
                                                                       const float* __restrict__ window, const MelTap* __restrict__ tab,
                                                                       const float* __restrict__ mel_basis, int64_t ldm, float* __restrict__ mel_out,
                                                                       int32_t* __restrict__ n_frames_out, int B, int T, int N0, int pad, int hop, int nb,
                                                                       int n_mel, float eps_add, float clamp_min, float log_clamp) {
    __shared__ cf tw[1024];
    __shared__ cf buf[FFT_WAVES][584];
    __shared__ MelTap stab[584];                                                  // bin k at k + (k >> 3): a lane's 8 consecutive bins, conflict-free
    __shared__ float macc[FFT_WAVES][MELF_MAXMEL];
    __shared__ float melt[MELF_MAXMEL][MELF_FRAMES + 1];
    for (int m = threadIdx.x; m < 1024; m += 64 * FFT_WAVES) { float sn, cs; sincospif(-(float)m * (1.0f / 512.0f), &sn, &cs); tw[m] = {cs, sn}; }
    for (int k = threadIdx.x; k < nb; k += 64 * FFT_WAVES) stab[k + (k >> 3)] = tab[k];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int m = lane; m < MELF_MAXMEL; m += 64) macc[wave][m] = 0.f;
    cf* sb = buf[wave];
    const int n2 = lane >> 3, n3 = lane & 7;
    float2 w2[8];
#pragma unroll
    for (int n1 = 0; n1 < 8; ++n1) w2[n1] = *reinterpret_cast<const float2*>(window + 2 * (64 * n1 + lane));
    __syncthreads();
    const int gpc = (T + MELF_FRAMES - 1) / MELF_FRAMES;                          // frame groups per clip
    for (int64_t grp = blockIdx.x; grp < (int64_t)B * gpc; grp += gridDim.x) {
        const int b = (int)(grp / gpc), t0 = (int)(grp - (int64_t)b * gpc) * MELF_FRAMES;
        int N = N0;
        const float* xf = nullptr; const int16_t* xi = nullptr;
        if constexpr (SRC == 0) xf = reinterpret_cast<const float*>(src) + (int64_t)b * ldx;
        else {
            const int i0 = order ? order[b] : b;
            N = lens[i0];
            xi = reinterpret_cast<const int16_t*>(src) + offsets[i0];
            if (n_frames_out && t0 == 0 && threadIdx.x == 0) n_frames_out[b] = (N + 2 * pad - 1024) / hop + 1;
        }
        auto sample = [&](int i) -> float {                                      // x[reflect(i)], torch 'reflect' (no edge repeat); far-out frames of a
            int sidx = i < 0 ? -i : i;                                           // ragged batch (zeroed later) are clamped into the clip
            if (sidx >= N) sidx = 2 * (N - 1) - sidx;
            sidx = min(max(sidx, 0), N - 1);
            if constexpr (SRC == 0) return xf[sidx]; else return (float)xi[sidx] * (1.0f / 32768.0f);
        };
#pragma unroll 1
        for (int it = 0; it < MELF_FRAMES / FFT_WAVES; ++it) {
            const int fi = it * FFT_WAVES + wave, t = t0 + fi;
            const bool live = t < T;
            const int base = (live ? t : 0) * hop - pad;
            cf a[8];
            if (SRC == 0 && base >= 0 && base + 1024 <= N && ((ldx | hop | pad) & 1) == 0) {      // interior frame: 8-byte vector loads
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    const float2 v = *reinterpret_cast<const float2*>(xf + base + 2 * (64 * n1 + lane));
                    a[n1] = {v.x * w2[n1].x, v.y * w2[n1].y};
                }
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 8; ++n1) {
                    const int i = base + 2 * (64 * n1 + lane);
                    a[n1] = {sample(i) * w2[n1].x, sample(i + 1) * w2[n1].y};
                }
            }
            dft8(a);
#pragma unroll
            for (int k1 = 0; k1 < 8; ++k1) sb[72 * k1 + lane] = k1 ? cmul(a[k1], tw[(16 * n2 * k1) & 1023]) : a[0];
            __builtin_amdgcn_wave_barrier();                                      // the exchange buffer is the wave's own: LDS is in order within a wave
            const int k1 = lane >> 3;
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = sb[72 * k1 + 8 * q + n3];
            dft8(a);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k2 = 0; k2 < 8; ++k2) sb[9 * (k1 + 8 * k2) + n3] = cmul(a[k2], tw[(2 * n3 * (k1 + 8 * k2)) & 1023]);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = sb[9 * lane + q];
            dft8(a);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k3 = 0; k3 < 8; ++k3) { const int k = lane + 64 * k3; sb[k + (k >> 3)] = a[k3]; }      // Z[k] at k + (k >> 3)
            if (lane == 0) sb[512 + 64] = a[0];                                   // Z[512] = Z[0]
            __builtin_amdgcn_wave_barrier();
            // a lane takes 8 CONSECUTIVE bins (lane 63 also bin 512): consecutive bins share their one or two triangular filters, so a lane sums
            // them in registers and issues a handful of LDS adds per frame (the bins of a wave instruction would otherwise collide on one address)
            int cur = -1; float accA = 0.f, accB = 0.f;
            auto flush = [&](int m, float v) { if (m >= 0 && m < n_mel && v != 0.f) unsafeAtomicAdd(&macc[wave][m], v); };
#pragma unroll 1
            for (int j = 0; j < 9; ++j) {
                const int k = 8 * lane + j;
                if (j == 8 && lane != 63) break;
                if (k >= nb) break;
                const int kc = 512 - k;
                const cf zk = sb[k + (k >> 3)], zc = sb[kc + (kc >> 3)];
                const cf e = {0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y)};
                const cf d = {0.5f * (zk.x - zc.x), 0.5f * (zk.y + zc.y)};
                const cf o = cmul(cmni(d), tw[k]);
                const float re = e.x + o.x, im = e.y + o.y;
                float pw = re * re + im * im + eps_add;
                if (clamp_min > 0.f) pw = fmaxf(pw, clamp_min);
                const float mg = sqrtf(pw);
                const MelTap tp = stab[k + (k >> 3)];
                if (tp.m0 >= 0) {
                    if (tp.m0 != cur) {
                        flush(cur, accA);
                        if (tp.m0 == cur + 1) { accA = accB; } else { flush(cur + 1, accB); accA = 0.f; }
                        accB = 0.f; cur = tp.m0;
                    }
                    accA += tp.w0 * mg;
                    if (tp.m1 >= 0) accB += tp.w1 * mg;
                } else if (tp.m0 == -2) {
                    for (int m = 0; m < n_mel; ++m) { const float w = mel_basis[(int64_t)m * ldm + k]; if (w != 0.f) unsafeAtomicAdd(&macc[wave][m], w * mg); }
                }
            }
            flush(cur, accA); flush(cur + 1, accB);
            __builtin_amdgcn_wave_barrier();
            for (int m = lane; m < n_mel; m += 64) {
                melt[m][fi] = logf(fmaxf(macc[wave][m], log_clamp));
                macc[wave][m] = 0.f;
            }
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < n_mel * MELF_FRAMES; idx += 64 * FFT_WAVES) {
            const int m = idx / MELF_FRAMES, f = idx - m * MELF_FRAMES, t = t0 + f;
            if (t < T) mel_out[((int64_t)b * n_mel + m) * T + t] = melt[m][f];
        }
        __syncthreads();
    }
}
// g_mel_dft: 0 (default) = the FFT wherever n_fft == 1024 (all three variants), and the forward mel as the one fused kernel above; 2 = the FFT with
// the four-launch pipeline (pad, FFT, magnitude, filterbank GEMM); 1 = always the dense DFT GEMM (xva_mel_set_dft)
static int g_mel_dft = 0;
extern "C" int xva_mel_set_dft(int mode) { int old = g_mel_dft; g_mel_dft = mode; return old; }
static inline int64_t al4(int64_t x) { return (x + 3) & ~(int64_t)3; }
static inline int64_t al32(int64_t x) { return (x + 31) & ~(int64_t)31; }

struct MelPlan {
    int T, Np, nb;
    int64_t ldy, lds, ldm;
    int64_t off_pad, off_spec, off_mag, total;  // in floats
};

static int mel_plan(const xva_mel_config* c, int B, int N, MelPlan* pl) {
    XVA_CHECK_ARG(c->n_fft > 0 && c->n_fft % 4 == 0 && c->hop > 0 && c->hop % 4 == 0, "mel: n_fft and hop must be multiples of 4");
    XVA_CHECK_ARG(c->pad >= 0 && c->pad < N, "mel: reflect pad %d needs N > pad (N=%d)", c->pad, N);
    pl->Np = N + 2 * c->pad;
    XVA_CHECK_ARG(pl->Np >= c->n_fft, "mel: clip shorter than one frame");
    pl->T = (pl->Np - c->n_fft) / c->hop + 1;
    pl->nb = c->n_fft / 2 + 1;
    pl->ldy = al4(pl->Np);
    pl->lds = al4(2 * pl->nb);
    pl->ldm = al32(pl->nb);
    pl->off_pad = 0;
    pl->off_spec = al4((int64_t)B * pl->ldy + c->n_fft);  // slack: last frame row never over-reads
    pl->off_mag = pl->off_spec + (int64_t)B * pl->T * pl->lds;
    pl->total = pl->off_mag + (int64_t)B * pl->T * pl->ldm;
    return XVA_OK;
}

// spec[b][t][re | im] = windowed DFT of frame t of clip b: the 1024-point FFT kernel, or (other sizes, xva_mel_set_dft(1)) the overlapping-row GEMM
static int stft_spec(const xva_mel_config* c, const MelPlan& pl, int B, const float* ypad, const float* dft_basis, float* spec, void* stream) {
    if (c->n_fft == 1024 && g_mel_dft != 1 && c->hop % 4 == 0) {
        const int64_t nframes = (int64_t)B * pl.T;
        int64_t grid = (nframes + FFT_WAVES - 1) / FFT_WAVES;
        if (grid > 256 * 8) grid = 256 * 8;
        hipLaunchKernelGGL(xva_stft_fft1024_kernel, dim3((unsigned)grid), dim3(64 * FFT_WAVES), 0, (hipStream_t)stream, ypad, dft_basis, spec, B, pl.T, c->hop,
                           pl.ldy, pl.lds, pl.nb);
        XVA_LAUNCH_CHECK();
        return XVA_OK;
    }
    xva_gemm_params g;
    memset(&g, 0, sizeof(g));
    g.A = ypad; g.B = dft_basis; g.C = spec;
    g.M = pl.T; g.N = 2 * pl.nb; g.K = c->n_fft;
    g.lda = c->hop; g.ldb = c->n_fft; g.ldc = pl.lds;
    g.batch = B; g.sA = pl.ldy; g.sB = 0; g.sC = (int64_t)pl.T * pl.lds;
    g.alpha = 1.f; g.splitk = 1; g.compute = 0; g.layout = XVA_GEMM_NT;
    return xva_gemm(&g, stream);
}

extern "C" int xva_mel_num_frames(const xva_mel_config* c, int N) {
    MelPlan pl;
    if (!c || mel_plan(c, 1, N, &pl) != XVA_OK) return -1;
    return pl.T;
}

extern "C" int64_t xva_mel_workspace_bytes(const xva_mel_config* c, int B, int N) {
    MelPlan pl;
    if (!c || mel_plan(c, B, N, &pl) != XVA_OK) return -1;
    return pl.total * (int64_t)sizeof(float);
}

static int mel_core(const xva_mel_config* c, const MelPlan& pl, int B, const float* dft_basis, const float* mel_basis_padded, float* mel_out,
                    float* workspace, void* stream);
// the fused forward kernel: n_fft 1024 (its FFT), up to MELF_MAXMEL filters, the default front-end mode
static bool mel_fused_ok(const xva_mel_config* c, const MelPlan& pl) {
    return c->n_fft == 1024 && g_mel_dft == 0 && c->hop % 4 == 0 && c->n_mel <= MELF_MAXMEL && pl.nb <= 513;
}
static unsigned mel_fused_grid(int B, int T) {
    const int64_t groups = (int64_t)B * ((T + MELF_FRAMES - 1) / MELF_FRAMES);
    return (unsigned)(groups < 256 * 6 ? (groups < 1 ? 1 : groups) : 256 * 6);
}

static int mel_spectrogram_impl(const xva_mel_config* c, const float* wav, int B, int N, int64_t ld_wav, const float* dft_basis, const float* mel_basis_padded,
                                float* mel_out, float* workspace, int64_t workspace_bytes, void* stream, bool keep_intermediates);
extern "C" int xva_mel_spectrogram(const xva_mel_config* c, const float* wav, int B, int N, int64_t ld_wav,
                                   const float* dft_basis, const float* mel_basis_padded, float* mel_out,
                                   float* workspace, int64_t workspace_bytes, void* stream) {
    return mel_spectrogram_impl(c, wav, B, N, ld_wav, dft_basis, mel_basis_padded, mel_out, workspace, workspace_bytes, stream, false);
}
// keep_intermediates: the differentiable mel's backward reads the spectrum and the magnitudes out of the workspace (four-launch pipeline)
static int mel_spectrogram_impl(const xva_mel_config* c, const float* wav, int B, int N, int64_t ld_wav, const float* dft_basis, const float* mel_basis_padded,
                                float* mel_out, float* workspace, int64_t workspace_bytes, void* stream, bool keep_intermediates) {
    XVA_CHECK_ARG(c && wav && dft_basis && mel_basis_padded && mel_out && workspace, "mel: null pointer");
    MelPlan pl;
    XVA_TRY(mel_plan(c, B, N, &pl));
    XVA_CHECK_ARG(workspace_bytes >= pl.total * (int64_t)sizeof(float), "mel: workspace too small (%ld < %ld)",
                  (long)workspace_bytes, (long)(pl.total * sizeof(float)));
    XVA_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "mel: workspace must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (!keep_intermediates && mel_fused_ok(c, pl)) {
        MelTap* tab = reinterpret_cast<MelTap*>(workspace + pl.off_spec);
        hipLaunchKernelGGL(xva_mel_taps_kernel, dim3(xva_cdiv(pl.nb, 64)), dim3(64), 0, st, mel_basis_padded, pl.ldm, c->n_mel, pl.nb, tab);
        hipLaunchKernelGGL((xva_mel_fused_kernel<0>), dim3(mel_fused_grid(B, pl.T)), dim3(64 * FFT_WAVES), 0, st, wav, ld_wav, nullptr, nullptr, nullptr, dft_basis,
                           tab, mel_basis_padded, pl.ldm, mel_out, nullptr, B, pl.T, N, c->pad, c->hop, pl.nb, c->n_mel, c->mag_eps_add, c->mag_clamp_min,
                           c->log_clamp);
        XVA_LAUNCH_CHECK();
        return XVA_OK;
    }
    float* ypad = workspace + pl.off_pad;
    {   // 1. reflect pad (plus zero the slack so over-reads of tail vectors are benign)
        int64_t total = (int64_t)B * pl.ldy;
        hipLaunchKernelGGL(xva_reflect_pad_kernel, dim3(xva_cdiv(total, 256)), dim3(256), 0, st, wav, ypad, B, N, c->pad,
                           ld_wav, pl.ldy, pl.Np);
        XVA_LAUNCH_CHECK();
    }
    return mel_core(c, pl, B, dft_basis, mel_basis_padded, mel_out, workspace, stream);
}

// Ragged int16 clips straight off disk -> zero-padded batch mel (+ per-frame energy), the device side of TTSDataset.get_mel +
// TTSCollate (python/fastpitch1_1/fastpitch/data_function.py:385-429,565-600): row r of the batch is clip order[r] (or r),
// y = int16 / 32768 reflect-padded by ITS OWN length; frames past the clip's own count are zeroed by the finishing pass.
__global__ void xva_reflect_pad_i16_ragged_kernel(const int16_t* __restrict__ flat, const int64_t* __restrict__ offsets, const int32_t* __restrict__ lens,
                                                  const int32_t* __restrict__ order, float* __restrict__ y, int pad, int64_t ldy,
                                                  int32_t* __restrict__ n_frames, int n_fft, int hop) {
    const int r = blockIdx.y, i0 = order ? order[r] : r;
    const int N = lens[i0], Np = N + 2 * pad;
    const int16_t* x = flat + offsets[i0];
    if (n_frames && blockIdx.x == 0 && threadIdx.x == 0) n_frames[r] = (Np - n_fft) / hop + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ldy; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < Np) {
            int s = (int)i - pad;
            if (s < 0) s = -s;
            if (s >= N) s = 2 * (N - 1) - s;
            v = (float)x[s] * (1.0f / 32768.0f);      // exact: audio / max_wav_value with max_wav_value = 2^15 (data_function.py:408-416)
        }
        y[(int64_t)r * ldy + i] = v;
    }
}
extern "C" int xva_mel_finish_ragged(float* mel, const int32_t* n_frames, float* energy, int B, int n_mel, int T, int energy_trunc, void* stream);
extern "C" int xva_mel_spectrogram_ragged(const xva_mel_config* c, const int16_t* flat, const int64_t* offsets, const int32_t* n_samples,
                                          const int32_t* order, int B, int Nmax, const float* dft_basis, const float* mel_basis_padded,
                                          float* mel_out, float* energy_out, int energy_trunc, int32_t* n_frames_out, float* workspace,
                                          int64_t workspace_bytes, void* stream) {
    XVA_CHECK_ARG(c && flat && offsets && n_samples && dft_basis && mel_basis_padded && mel_out && n_frames_out && workspace, "mel_ragged: null pointer");
    MelPlan pl;
    XVA_TRY(mel_plan(c, B, Nmax, &pl));
    XVA_CHECK_ARG(workspace_bytes >= pl.total * (int64_t)sizeof(float), "mel_ragged: workspace too small");
    XVA_CHECK_ARG(((uintptr_t)workspace % 16) == 0, "mel_ragged: workspace must be 16-byte aligned");
    if (mel_fused_ok(c, pl)) {
        hipStream_t st = (hipStream_t)stream;
        MelTap* tab = reinterpret_cast<MelTap*>(workspace + pl.off_spec);
        hipLaunchKernelGGL(xva_mel_taps_kernel, dim3(xva_cdiv(pl.nb, 64)), dim3(64), 0, st, mel_basis_padded, pl.ldm, c->n_mel, pl.nb, tab);
        hipLaunchKernelGGL((xva_mel_fused_kernel<1>), dim3(mel_fused_grid(B, pl.T)), dim3(64 * FFT_WAVES), 0, st, flat, (int64_t)0, offsets, n_samples, order,
                           dft_basis, tab, mel_basis_padded, pl.ldm, mel_out, n_frames_out, B, pl.T, Nmax, c->pad, c->hop, pl.nb, c->n_mel, c->mag_eps_add,
                           c->mag_clamp_min, c->log_clamp);
        XVA_LAUNCH_CHECK();
        return xva_mel_finish_ragged(mel_out, n_frames_out, energy_out, B, c->n_mel, pl.T, energy_trunc, stream);
    }
    hipLaunchKernelGGL(xva_reflect_pad_i16_ragged_kernel, dim3((unsigned)(pl.ldy / 256 < 1 ? 1 : (pl.ldy / 256 > 128 ? 128 : pl.ldy / 256)), B), dim3(256), 0,
                       (hipStream_t)stream, flat, offsets, n_samples, order, workspace + pl.off_pad, c->pad, pl.ldy, n_frames_out, c->n_fft, c->hop);
    XVA_LAUNCH_CHECK();
    XVA_TRY(mel_core(c, pl, B, dft_basis, mel_basis_padded, mel_out, workspace, stream));
    return xva_mel_finish_ragged(mel_out, n_frames_out, energy_out, B, c->n_mel, pl.T, energy_trunc, stream);
}

static int mel_core(const xva_mel_config* c, const MelPlan& pl, int B, const float* dft_basis, const float* mel_basis_padded, float* mel_out,
                    float* workspace, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    float* ypad = workspace + pl.off_pad;
    float* spec = workspace + pl.off_spec;
    float* mag = workspace + pl.off_mag;
    XVA_TRY(stft_spec(c, pl, B, ypad, dft_basis, spec, stream));   // 2. windowed DFT of every frame: spec[b][t][re | im]
    {   // 3. magnitude
        int64_t rows = (int64_t)B * pl.T;
        int64_t total = rows * pl.ldm;
        hipLaunchKernelGGL(xva_magnitude_kernel, dim3(xva_cdiv(total, 256)), dim3(256), 0, st, spec, mag, rows, pl.nb,
                           pl.lds, pl.ldm, c->mag_eps_add, c->mag_clamp_min);
        XVA_LAUNCH_CHECK();
    }
    {   // 4. mel filterbank with fused log(clamp(.)), output (B, n_mel, T)
        xva_gemm_params g;
        memset(&g, 0, sizeof(g));
        g.A = mel_basis_padded; g.B = mag; g.C = mel_out;
        g.M = c->n_mel; g.N = pl.T; g.K = (int)pl.ldm;
        g.lda = pl.ldm; g.ldb = pl.ldm; g.ldc = pl.T;
        g.batch = B; g.sA = 0; g.sB = (int64_t)pl.T * pl.ldm; g.sC = (int64_t)c->n_mel * pl.T;
        g.alpha = 1.f; g.act = XVA_ACT_LOGCLAMP; g.act_slope = c->log_clamp; g.splitk = 1; g.compute = 0; g.layout = XVA_GEMM_NT;
        XVA_TRY(xva_gemm(&g, stream));
    }
    return XVA_OK;
}


// Linear magnitude spectrogram in the reference's (B, n_fft/2+1, T) layout: the 513-bin posterior-encoder input of xVAPitch
// (TorchSTFT with use_mel=False, python/xvapitch/audio.py:138-171; the dataset-side AudioProcessor.spectrogram :632-652 is the same
// |STFT| without the 1e-8 clamp).  Same reflect pad + DFT GEMM; the magnitude pass transposes 32 x 32 tiles through LDS so both
// the [frame][bin] reads and the [bin][frame] writes are coalesced.
__global__ void xva_magnitude_t_kernel(const float* __restrict__ spec, float* __restrict__ out, int T, int nb, int64_t lds, float eps_add,
                                       float clamp_min) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, j0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 256 threads = 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, j = j0 + tx;
        float v = 0.f;
        if (t < T && j < nb) {
            const float* row = spec + ((int64_t)b * T + t) * lds;
            float p = row[j] * row[j] + row[nb + j] * row[nb + j] + eps_add;
            if (clamp_min > 0.f) p = fmaxf(p, clamp_min);
            v = sqrtf(p);
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r, t = t0 + tx;
        if (j < nb && t < T) out[((int64_t)b * nb + j) * T + t] = tile[tx][r];
    }
}
extern "C" int xva_linear_spectrogram(const xva_mel_config* c, const float* wav, int B, int N, int64_t ld_wav, const float* dft_basis,
                                      float* lin_out, float* workspace, int64_t workspace_bytes, void* stream) {
    XVA_CHECK_ARG(c && wav && dft_basis && lin_out && workspace, "linear_spectrogram: null pointer");
    MelPlan pl;
    XVA_TRY(mel_plan(c, B, N, &pl));
    XVA_CHECK_ARG(workspace_bytes >= pl.total * (int64_t)sizeof(float), "linear_spectrogram: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ypad = workspace + pl.off_pad;
    float* spec = workspace + pl.off_spec;
    hipLaunchKernelGGL(xva_reflect_pad_kernel, dim3(xva_cdiv((int64_t)B * pl.ldy, 256)), dim3(256), 0, st, wav, ypad, B, N, c->pad, ld_wav, pl.ldy, pl.Np);
    XVA_LAUNCH_CHECK();
    XVA_TRY(stft_spec(c, pl, B, ypad, dft_basis, spec, stream));
    hipLaunchKernelGGL(xva_magnitude_t_kernel, dim3(xva_cdiv(pl.T, 32), xva_cdiv(pl.nb, 32), B), dim3(256), 0, st, spec, lin_out, pl.T, pl.nb, pl.lds,
                       c->mag_eps_add, c->mag_clamp_min);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// Ragged batch of float clips (xVAPitch: python/xvapitch/dataset.py:251 computes `self.ap.spectrogram(wav)` per clip — librosa.stft, center,
// reflect padding at the clip's OWN ends — and the collate zero-pads the spectrograms, :470-475): row r of a dense (B, ld_wav) batch holds
// n_samples[r] valid samples; its 1 + n_samples[r] / hop frames are those of the clip alone, later frames are zero.
__global__ void xva_reflect_pad_f32_ragged_kernel(const float* __restrict__ wav, int64_t ld_wav, const int32_t* __restrict__ lens, float* __restrict__ y, int pad,
                                                  int64_t ldy, int32_t* __restrict__ n_frames, int n_fft, int hop) {
    const int r = blockIdx.y;
    const int N = lens[r], Np = N + 2 * pad;
    const float* x = wav + (int64_t)r * ld_wav;
    if (n_frames && blockIdx.x == 0 && threadIdx.x == 0) n_frames[r] = (Np - n_fft) / hop + 1;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ldy; i += (int64_t)gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < Np) {
            int s = (int)i - pad;
            if (s < 0) s = -s;
            if (s >= N) s = 2 * (N - 1) - s;
            v = x[s];
        }
        y[(int64_t)r * ldy + i] = v;
    }
}
__global__ void xva_zero_frames_kernel(float* __restrict__ out, const int32_t* __restrict__ n_frames, int C, int T) {
    const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T && t >= n_frames[b]) out[((int64_t)b * C + c) * T + t] = 0.f;
}
extern "C" int xva_linear_spectrogram_ragged(const xva_mel_config* c, const float* wav, const int32_t* n_samples, int B, int Nmax, int64_t ld_wav,
                                             const float* dft_basis, float* lin_out, int32_t* n_frames_out, float* workspace, int64_t workspace_bytes,
                                             void* stream) {
    XVA_CHECK_ARG(c && wav && n_samples && dft_basis && lin_out && n_frames_out && workspace, "linear_spectrogram_ragged: null pointer");
    MelPlan pl;
    XVA_TRY(mel_plan(c, B, Nmax, &pl));
    XVA_CHECK_ARG(workspace_bytes >= pl.total * (int64_t)sizeof(float), "linear_spectrogram_ragged: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* ypad = workspace + pl.off_pad;
    float* spec = workspace + pl.off_spec;
    hipLaunchKernelGGL(xva_reflect_pad_f32_ragged_kernel, dim3((unsigned)(pl.ldy / 256 < 1 ? 1 : (pl.ldy / 256 > 128 ? 128 : pl.ldy / 256)), B), dim3(256), 0, st,
                       wav, ld_wav, n_samples, ypad, c->pad, pl.ldy, n_frames_out, c->n_fft, c->hop);
    XVA_LAUNCH_CHECK();
    XVA_TRY(stft_spec(c, pl, B, ypad, dft_basis, spec, stream));
    hipLaunchKernelGGL(xva_magnitude_t_kernel, dim3(xva_cdiv(pl.T, 32), xva_cdiv(pl.nb, 32), B), dim3(256), 0, st, spec, lin_out, pl.T, pl.nb, pl.lds,
                       c->mag_eps_add, c->mag_clamp_min);
    XVA_LAUNCH_CHECK();
    hipLaunchKernelGGL(xva_zero_frames_kernel, dim3(xva_cdiv(pl.T, 256), pl.nb, B), dim3(256), 0, st, lin_out, n_frames_out, pl.nb, pl.T);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}

// ====================================================================================================================
// Differentiable mel: L1 mel loss of a generated waveform and its gradient w.r.t. the waveform
//   loss = scale * mean |mel_tgt - mel(wav)|        (F.l1_loss(y_mel, y_g_hat_mel) * 45, python/hifigan/xva_train.py:480,504)
// Backward of the same pipeline, transposed: d(log-clamp) -> mel^T GEMM -> magnitude -> DFT^T GEMM -> overlap-add -> reflect fold.
// ====================================================================================================================
struct MelBwdPlan { MelPlan f; int64_t off_dM, off_dmag, off_dfr, total, ldT; };
static int mel_bwd_plan(const xva_mel_config* c, int B, int N, MelBwdPlan* p) {
    XVA_TRY(mel_plan(c, B, N, &p->f));
    p->ldT = al4(p->f.T);        // dM rows are padded to a multiple of 4 frames (M3: 33 frames per 8192-sample segment)
    p->off_dM = al4(p->f.total);
    p->off_dmag = p->off_dM + al4((int64_t)B * c->n_mel * p->ldT);
    p->off_dfr = p->off_dmag + (int64_t)B * p->f.T * p->f.ldm;
    p->total = p->off_dfr + (int64_t)B * p->f.T * c->n_fft + 16;
    return XVA_OK;
}
// dM = -(scale / numel) * sign(tgt - mel) * exp(-mel) * [mel > log(clamp)] ; loss += (scale / numel) * sum |tgt - mel|
__global__ void mel_l1_kernel(const float* __restrict__ mel, const float* __restrict__ tgt, float* __restrict__ dM, float* __restrict__ loss,
                              int64_t n, float k, float log_clamp, int T, int ldT) {
    __shared__ float sh[16];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float m = mel[i], d = tgt[i] - m;
        acc += fabsf(d);
        float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        const int64_t row = i / T;
        dM[row * ldT + (i - row * T)] = (m > log_clamp) ? -k * sg * expf(-m) : 0.f;
    }
    acc = xva_block_sum(acc, sh);
    if (threadIdx.x == 0 && loss) atomicAdd(loss, acc * k);
}
// spec rows [re | im] (in place) <- dmag * re / mag , dmag * im / mag
// (M3: magnitude = sqrt(clamp(re^2 + im^2, clamp_min)) passes no gradient below the clamp)
__global__ void mel_dspec_kernel(float* __restrict__ spec, const float* __restrict__ mag, const float* __restrict__ dmag, int64_t rows, int nb,
                                 int64_t lds, int64_t ldm, float eps_add, float clamp_min) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * nb) return;
    int64_t r = idx / nb;
    int j = (int)(idx - r * nb);
    float mg = mag[r * ldm + j], g = dmag[r * ldm + j];
    float sc = mg > 0.f ? g / mg : 0.f;
    if (clamp_min > 0.f) {
        const float re = spec[r * lds + j], im = spec[r * lds + nb + j];
        if (re * re + im * im + eps_add < clamp_min) sc = 0.f;
    }
    spec[r * lds + j] *= sc;
    spec[r * lds + nb + j] *= sc;
}
// d_wav[b][n] (+)= sum over padded positions aliasing sample n (direct + reflect images) of the overlap-add of frame gradients
__global__ void mel_fold_kernel(const float* __restrict__ dfr, float* __restrict__ dwav, int B, int N, int pad, int T, int n_fft, int hop,
                                int64_t ld_dwav, int accumulate) {
    int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= (int64_t)B * N) return;
    int b = (int)(gi / N), n = (int)(gi % N);
    int cand[3] = {n + pad, pad - n, pad + 2 * (N - 1) - n};
    bool ok[3] = {true, n >= 1 && n <= pad, n <= N - 2 && n >= N - 1 - pad};
    float acc = 0.f;
    const int Np = N + 2 * pad;
    for (int q = 0; q < 3; ++q) {
        if (!ok[q]) continue;
        int i = cand[q];
        if (i < 0 || i >= Np) continue;
        int t1 = i / hop; if (t1 > T - 1) t1 = T - 1;
        int t0 = (i - n_fft + hop) / hop; if (i - n_fft + 1 <= 0) t0 = 0;
        if (t0 < 0) t0 = 0;
        for (int t = t0; t <= t1; ++t) {
            int off = i - t * hop;
            if (off >= 0 && off < n_fft) acc += dfr[((int64_t)b * T + t) * n_fft + off];
        }
    }
    float* dst = dwav + (int64_t)b * ld_dwav + n;
    if (accumulate) *dst += acc; else *dst = acc;
}

extern "C" int64_t xva_mel_backward_workspace_bytes(const xva_mel_config* c, int B, int N) {
    MelBwdPlan p;
    if (!c || mel_bwd_plan(c, B, N, &p) != XVA_OK) return -1;
    return p.total * (int64_t)sizeof(float);
}

extern "C" int xva_mel_l1_loss_backward(const xva_mel_config* c, const float* wav, int B, int N, int64_t ld_wav, const float* mel_tgt,
                                        const float* dft_basis, const float* mel_basis_padded, float scale, float* mel_out, float* loss_out,
                                        float* d_wav, int64_t ld_dwav, int accumulate, float* workspace, int64_t workspace_bytes, void* stream) {
    XVA_CHECK_ARG(c && wav && mel_tgt && dft_basis && mel_basis_padded && mel_out && d_wav && workspace, "mel_l1_loss_backward: null pointer");
    MelBwdPlan p;
    XVA_TRY(mel_bwd_plan(c, B, N, &p));
    XVA_CHECK_ARG(workspace_bytes >= p.total * (int64_t)sizeof(float), "mel backward: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    XVA_TRY(mel_spectrogram_impl(c, wav, B, N, ld_wav, dft_basis, mel_basis_padded, mel_out, workspace, workspace_bytes, stream, true));
    const MelPlan& f = p.f;
    float* spec = workspace + f.off_spec;
    float* mag = workspace + f.off_mag;
    float* dM = workspace + p.off_dM;
    float* dmag = workspace + p.off_dmag;
    float* dfr = workspace + p.off_dfr;
    const int64_t nmel = (int64_t)B * c->n_mel * f.T;
    {
        int grid = (int)((nmel + 255) / 256); if (grid > 1024) grid = 1024;
        if (p.ldT != f.T && hipMemsetAsync(dM, 0, (size_t)B * c->n_mel * p.ldT * sizeof(float), st) != hipSuccess) {
            xva_set_error("mel backward: memset failed"); return XVA_ERR_HIP;
        }
        hipLaunchKernelGGL(mel_l1_kernel, dim3(grid), dim3(256), 0, st, mel_out, mel_tgt, dM, loss_out, nmel, scale / (float)nmel, logf(c->log_clamp),
                           f.T, (int)p.ldT);
        XVA_LAUNCH_CHECK();
    }
    {   // dmag[b] (T x ldm) = dM[b]^T (T x n_mel) * melW (n_mel x ldm)
        xva_gemm_params g;
        memset(&g, 0, sizeof(g));
        g.layout = XVA_GEMM_TN; g.A = dM; g.B = mel_basis_padded; g.C = dmag;
        g.M = f.T; g.N = (int)f.ldm; g.K = c->n_mel;
        g.lda = p.ldT; g.ldb = f.ldm; g.ldc = f.ldm;
        g.batch = B; g.sA = (int64_t)c->n_mel * p.ldT; g.sB = 0; g.sC = (int64_t)f.T * f.ldm;
        g.alpha = 1.f; g.beta = 1.f; g.splitk = 1; g.compute = 0; g.mask_mul = 1;
        XVA_TRY(xva_gemm(&g, stream));
    }
    {
        const int64_t rows = (int64_t)B * f.T;
        hipLaunchKernelGGL(mel_dspec_kernel, dim3(xva_cdiv(rows * f.nb, 256)), dim3(256), 0, st, spec, mag, dmag, rows, f.nb, f.lds, f.ldm, c->mag_eps_add, c->mag_clamp_min);
        XVA_LAUNCH_CHECK();
    }
    {   // dframes (rows x n_fft) = dspec (rows x 2 nb) * basis (2 nb x n_fft)
        xva_gemm_params g;
        memset(&g, 0, sizeof(g));
        g.layout = XVA_GEMM_NN; g.A = spec; g.B = dft_basis; g.C = dfr;
        g.M = (int)((int64_t)B * f.T); g.N = c->n_fft; g.K = 2 * f.nb;
        g.lda = f.lds; g.ldb = c->n_fft; g.ldc = c->n_fft;
        g.batch = 1; g.alpha = 1.f; g.beta = 1.f; g.splitk = 1; g.compute = 0; g.mask_mul = 1;
        XVA_TRY(xva_gemm(&g, stream));
    }
    hipLaunchKernelGGL(mel_fold_kernel, dim3(xva_cdiv((int64_t)B * N, 256)), dim3(256), 0, st, dfr, d_wav, B, N, c->pad, f.T, c->n_fft, c->hop,
                       ld_dwav, accumulate);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
