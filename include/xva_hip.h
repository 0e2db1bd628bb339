/* xva_hip.h — C ABI of libxvahip.so, the MI355X (gfx950) kernel library behind the
 * FastPitch1.1 + HiFi-GAN training hot path of xVATrainer.
 *
 * The reference has no FFI for this path (it is PyTorch ops called from Python; SURVEY.md
 * §8b): every entry point below names the reference Python call site it replaces, and
 * INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *  - Plain pointers + sizes only; no torch types.  Device pointers unless said otherwise.
 *  - The CALLER owns every buffer, including scratch: query `*_workspace_bytes`, allocate,
 *    pass it in.  No hidden allocation, no hidden synchronisation: every call only enqueues
 *    work on `stream` (a hipStream_t passed as void*) and returns.
 *  - Return value: 0 on success, negative XVA_ERR_* otherwise; `xva_last_error()` gives the
 *    thread-local message.
 *  - Pointers must be 16-byte aligned; leading dimensions are in elements.
 */
#ifndef XVA_HIP_H
#define XVA_HIP_H
#include <stdint.h>
#include "xva_gemm.h"
#ifdef __cplusplus
extern "C" {
#endif

const char* xva_last_error(void);
int xva_abi_version(void);
const char* xva_target_arch(void);

/* ------------------------------------------------------------------ mel front end ---- */
/* One config covers the reference's three mel variants:
 *   M1 TacotronSTFT.mel_spectrogram (fastpitch1_1/common/layers.py:121-138):
 *        pad = n_fft/2, mag_eps_add = 0,    mag_clamp_min = 0
 *   M2 mel_spectrogram (hifigan/meldataset.py:217-240):
 *        pad = (n_fft-hop)/2, mag_eps_add = 1e-9, mag_clamp_min = 0
 *   M3 TorchSTFT.__call__ (xvapitch/audio.py:138-181):
 *        pad = n_fft/2, mag_eps_add = 0,    mag_clamp_min = 1e-8
 * log_clamp is the dynamic-range-compression floor (1e-5 in all three). */
typedef struct xva_mel_config {
    int32_t n_fft;        /* 1024 */
    int32_t hop;          /* 256  */
    int32_t n_mel;        /* 80   */
    int32_t pad;          /* reflect padding on each side */
    float mag_eps_add;    /* magnitude = sqrt(max(re^2 + im^2 + mag_eps_add, mag_clamp_min)) */
    float mag_clamp_min;
    float log_clamp;      /* out = log(max(mel, log_clamp)) */
} xva_mel_config;

/* Number of frames T for clips of N samples, or -1 on bad arguments. */
int xva_mel_num_frames(const xva_mel_config* cfg, int N);
int64_t xva_mel_workspace_bytes(const xva_mel_config* cfg, int B, int N);
/* wav:  (B, N) fp32 in [-1, 1], row stride ld_wav.
 * dft_basis: (2*(n_fft/2+1), n_fft) fp32 = [real rows | imag rows] of the windowed DFT, the
 *            same matrix as STFT.forward_basis (common/stft.py:57-84).
 * mel_basis_padded: (n_mel, roundup(n_fft/2+1, 32)) fp32, librosa Slaney filterbank with the
 *            padded columns zero.
 * mel_out: (B, n_mel, T) fp32 — the reference layout. */
int xva_mel_spectrogram(const xva_mel_config* cfg, const float* wav, int B, int N, int64_t ld_wav,
                        const float* dft_basis, const float* mel_basis_padded, float* mel_out,
                        float* workspace, int64_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
