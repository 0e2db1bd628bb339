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
/* Optional per-launch GEMM timing with HIP events (used by bench.py's roofline leg only). */
void xva_prof_enable(int on);
int xva_prof_collect(double* out, int cap);

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
/* Diagnostics / test knob: 0 (default) = the windowed DFT of every frame is a 1024-point real FFT wherever n_fft == 1024 (all three reference
 * configurations) and the FORWARD mel (xva_mel_spectrogram, xva_mel_spectrogram_ragged) is one fused kernel — reflect-indexed frame, FFT,
 * magnitude, mel filterbank, log — whose intermediates stay in LDS; 2 = the FFT inside the four-launch pipeline (pad, FFT, magnitude, filterbank
 * GEMM; what the differentiable mel's backward always uses); 1 = always the dense GEMM against the windowed DFT basis (the reference's own
 * formulation).  Returns the previous mode. */
int xva_mel_set_dft(int mode);
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

/* Linear magnitude spectrogram (B, n_fft/2+1, T) fp32: TorchSTFT with use_mel=False (python/xvapitch/audio.py:138-171), the
 * 513-bin posterior-encoder input of xVAPitch (dataset side: AudioProcessor.spectrogram :632-652).  Workspace as above. */
int xva_linear_spectrogram(const xva_mel_config* cfg, const float* wav, int B, int N, int64_t ld_wav, const float* dft_basis,
                           float* lin_out, float* workspace, int64_t workspace_bytes, void* stream);
/* The same for a ragged batch of float clips (xVAPitch: python/xvapitch/dataset.py:251 computes AudioProcessor.spectrogram per clip — librosa.stft,
 * center, reflect padding at the clip's own ends — and the collate zero-pads, :470-475): row r of wav (B, ld_wav) holds n_samples[r] valid samples;
 * lin_out (B, n_fft / 2 + 1, 1 + Nmax / hop) gets the clip's own 1 + n_samples[r] / hop frames (written to n_frames_out[r]) and zeros after.
 * Every n_samples[r] must exceed n_fft / 2 (reflect padding, as torch.stft / librosa require); the values live on the device and are not checked. */
int xva_linear_spectrogram_ragged(const xva_mel_config* cfg, const float* wav, const int32_t* n_samples, int B, int Nmax, int64_t ld_wav,
                                  const float* dft_basis, float* lin_out, int32_t* n_frames_out, float* workspace, int64_t workspace_bytes, void* stream);

/* Differentiable mel for the generator's L1 mel loss (python/hifigan/xva_train.py:480,504; xVAPitch: VitsGeneratorLoss,
 * python/xvapitch/losses.py:187-193 with the M3 config):
 *   *loss_out += scale * mean|mel_tgt - mel(wav)| ; d_wav (+)= d loss / d wav ; mel_out = mel(wav).
 * Same config / basis arguments as xva_mel_spectrogram; any frame count. */
int64_t xva_mel_backward_workspace_bytes(const xva_mel_config* cfg, int B, int N);
int xva_mel_l1_loss_backward(const xva_mel_config* cfg, const float* wav, int B, int N, int64_t ld_wav, const float* mel_tgt,
                             const float* dft_basis, const float* mel_basis_padded, float scale, float* mel_out, float* loss_out,
                             float* d_wav, int64_t ld_dwav, int accumulate, float* workspace, int64_t workspace_bytes, void* stream);

/* --------------------------------------------------------- on-device batch preparation ---- */
/* What the reference does on the CPU in DataLoader workers, moved next to the mel kernels: the host hands over RAGGED
 * data as it comes off disk (int16 clips, symbol ids, cached pitch / durations) concatenated into flat buffers with an
 * offsets array (int64, elements, B entries) and per-item lengths (int32); everything below is one coalesced pass. */
#define XVA_DATA_I16 0
#define XVA_DATA_I32 1
#define XVA_DATA_I64 2
#define XVA_DATA_F32 3
#define XVA_DATA_F64 4
#define XVA_DATA_F32_TRUNC 5   /* destination only: fp32 holding trunc(x) — a value that passed through a LongTensor (see below) */
/* TTSCollate's descending sort by text length (python/fastpitch1_1/fastpitch/data_function.py:569-572):
 * order[r] = index of the r-th longest item, ties in ascending item order (stable).  B <= 4096. */
int xva_data_rank_desc(const int32_t* lens, int B, int32_t* order, void* stream);
/* TTSCollate's right zero-padding (data_function.py:574-660) of text ids / pitch / durations: item i is an
 * (inner, lens[i]) row-major slab at flat[offsets[i] * inner]; dst (B, inner, max_len) row r = item order[r] (order may be
 * NULL), truncated / zero-padded to max_len, converted src_dtype -> dst_dtype.  lens_out (B, may be NULL) = min(len, max_len).
 * Reference quirk kept for parity: TTSCollate allocates pitch_padded / energy_padded / durs_padded with the TEXT's dtype
 * (`dtype=batch[0][0].dtype`, :594-606,629-636), so pitch, energy and durations are truncated toward zero on the way into
 * the batch; XVA_DATA_F32_TRUNC / XVA_DATA_I32 destinations and `energy_trunc` below reproduce that. */
int xva_data_pad_gather(const void* flat, int src_dtype, const int64_t* offsets, const int32_t* lens, const int32_t* order, void* dst,
                        int dst_dtype, int B, int inner, int max_len, int32_t* lens_out, void* stream);
/* TTSDataset.get_mel + the mel part of TTSCollate (data_function.py:385-429,574-590) + energy (:327): clips are int16,
 * y = x / 32768, each reflect-padded by its own length; mel_out (B, n_mel, T(Nmax)) with frames past a clip's own
 * n_frames_out[r] zeroed; energy_out (B, T) = ||mel[:, t]||_2 (may be NULL).  Row r = clip order[r] (order may be NULL).
 * Workspace: xva_mel_workspace_bytes(cfg, B, Nmax). */
int xva_mel_spectrogram_ragged(const xva_mel_config* cfg, const int16_t* flat, const int64_t* offsets, const int32_t* n_samples,
                               const int32_t* order, int B, int Nmax, const float* dft_basis, const float* mel_basis_padded,
                               float* mel_out, float* energy_out, int energy_trunc, int32_t* n_frames_out, float* workspace,
                               int64_t workspace_bytes, void* stream);
int xva_mel_finish_ragged(float* mel, const int32_t* n_frames, float* energy, int B, int n_mel, int T, int energy_trunc, void* stream);
/* beta_binomial_prior_distribution + its collate (data_function.py:84-94,640-660): out (B, Tm, Tt) fp32,
 * out[r][m][k] = betabinom(n = P, a = m + 1, b = M - m).pmf(k) for m < M = mel_lens[r], k < P = text_lens[r], else 0. */
int xva_data_betabinom_prior(const int32_t* text_lens, const int32_t* mel_lens, float* out, int B, int Tm, int Tt, void* stream);
/* MelDataset.__getitem__ (python/hifigan/meldataset.py:340-373): peak[i] = max|x| of clip i (int16 magnitudes);
 * out (B, seg) fp32 = (x / 32768) / (peak / 32768) * gain over [starts[r], starts[r] + seg) of clip r, zero past its end
 * (normalize = 0: plain x / 32768).  Arithmetic in fp64 then rounded to fp32, as numpy + torch.FloatTensor do. */
int xva_wav_peak_i16(const int16_t* flat, const int64_t* offsets, const int32_t* lens, int B, int32_t* peak, void* stream);
int xva_wav_crop_norm(const int16_t* flat, const int64_t* offsets, const int32_t* lens, const int32_t* starts, const int32_t* peak,
                      float* out, int B, int seg, double gain, int normalize, void* stream);

/* ------------------------------------------------------------- FastPitch 1.1 engine ---- */
/* Replaces, for training stages 2-4, the PyTorch graph behind
 *   y_pred = FastPitch.forward(x)          python/fastpitch1_1/fastpitch/model.py:325-423
 *   loss   = FastPitchLoss.forward(...)    python/fastpitch1_1/fastpitch/loss_function.py:63-154
 *   loss.backward()                        python/fastpitch1_1/xva_train.py:810-813
 * called from FastPitchTrainer.iteration (xva_train.py:757-911).
 *
 * Sequence tensors are "padded token-major": (B, T+2, C) fp32, row 0 and row T+1 of each item structurally
 * zero, rows 1..len live.  Parameters / gradients are ONE flat fp32 buffer laid out by xva_fp_tensor_info
 * (conv k=3 weights stored tap-major [Cout][3][Cin]; `kind` = 1 marks them — the host permutes to/from the
 * checkpoint layout [Cout][Cin][3]).  The workspace must be zero-filled ONCE when allocated (guard rows).  Activation
 * slots are dtype-dependent (fp32 or bf16 per dims.compute): xva_fp_slot_offset returns BYTE offsets. */
typedef struct xva_fp_dims {
    int32_t B;        /* items in the micro-batch */
    int32_t Tt;       /* padded text length  (max_inp_lengths[0]) */
    int32_t Tm;       /* padded mel length   (max_mel_lengths[0]) */
    int32_t stage;    /* training stage 2, 3 or 4 (model.training_stage) */
    int32_t compute;  /* 0: fp32 activations + exact fp32 MFMA (parity mode); 1: bf16 activations + bf16-input MFMA (fp32 accumulate);
                       * 2 (round 6): fp16 OPERANDS over an fp32 residual stream — the storage plan and slots of mode 0 (fp32 activation slots), every MFMA operand
                       * (layer inputs, LayerNorm outputs, qkv, attention output, the feed-forward intermediate, their gradients, all weights) a single IEEE-half copy,
                       * products on v_mfma_f32_16x16x32_f16: the arithmetic width of the reference's own fp16 autocast path (python/fastpitch1_1/xva_train.py:350,787)
                       * and the cheapest format whose OUTPUTS stay within 1e-3 of the fp32 reference (profiles/r06_precision_probe.txt).  The activation-gradient
                       * buffers are fp16 too: pass a power-of-two loss scale through xva_fp_loss_grads' grad_scale and its inverse through xva_lamb_step's inv_scale
                       * (the reference's GradScaler, xva_train.py:856-859). */
    float p_dropout;  /* 0.1 in the reference's training mode (model.py:149-174: dropout, dropatt, predictor dropout); 0 = eval */
    uint64_t seed;    /* dropout seed of this micro-batch (forward and backward must pass the same value) */
} xva_fp_dims;

typedef struct xva_fp_batch {
    const int32_t* text;      /* (B, Tt) symbol ids, 0 = padding            (TTSCollate text_padded)  */
    const int32_t* in_lens;   /* (B)                                                                   */
    const int32_t* durs;      /* (B, Tt) integer durations, 0 on padding    (durs_padded)              */
    const float* pitch;       /* (B, Tm) frame-level pitch, 0 = unvoiced    (pitch_padded[:, 0])       */
    const float* energy;      /* (B, Tm) frame-level energy                 (energy_padded)            */
    const float* pos_table;   /* (>= max(Tt, Tm), 384) sinusoid table cat(sin, cos) (transformer.py:21-35) */
} xva_fp_batch;

enum {
    XVA_FP_SLOT_MEL_OUT = 0,   /* (B, Tm+2, 80)  */
    XVA_FP_SLOT_PITCH_PRED,    /* (B, Tt+2)      */
    XVA_FP_SLOT_ENERGY_PRED,   /* (B, Tt+2)      */
    XVA_FP_SLOT_LOG_DUR_PRED,  /* (B, Tt+2)      */
    XVA_FP_SLOT_DUR_PRED,      /* (B, Tt+2)      */
    XVA_FP_SLOT_PITCH_TGT,     /* (B, Tt+2)      */
    XVA_FP_SLOT_ENERGY_TGT,    /* (B, Tt+2)      */
    XVA_FP_SLOT_DEC_LENS,      /* (B) int32      */
    XVA_FP_SLOT_LOSS_ACC,      /* 8 floats: mel_num, mel_den, pitch_num, tok_den, energy_num, dur_num, -, - */
    XVA_FP_SLOT_LOSSES,        /* 8 floats: total, mel, dur, pitch, energy, -, -, - */
    XVA_FP_SLOT_D_MEL,         /* (B, Tm+2, 80)  */
    XVA_FP_SLOT_D_PITCH,       /* (B, Tt+2)      */
    XVA_FP_SLOT_D_ENERGY,      /* (B, Tt+2)      */
    XVA_FP_SLOT_D_LOGDUR,      /* (B, Tt+2)      */
    XVA_FP_SLOT_ENC_OUT,       /* (B, Tt+2, 384) */
    XVA_FP_SLOT_DEC_OUT,       /* (B, Tm+2, 384) */
    XVA_FP_SLOT_ENC_COND,      /* (B, Tt+2, 384) */
    XVA_FP_SLOT_COUNT
};

int64_t xva_fp_param_floats(void);
int xva_fp_num_tensors(void);
int xva_fp_tensor_info(int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim, int64_t* shape4,
                       int32_t* kind);
int xva_fp_trainable_ranges(int stage, int64_t* begins, int64_t* ends, int cap);
int64_t xva_fp_workspace_bytes(const xva_fp_dims* d);
int xva_fp_slot_offset(const xva_fp_dims* d, int slot, int64_t* off_bytes);
int xva_fp_forward(const xva_fp_dims* d, const float* params, const xva_fp_batch* batch, void* workspace,
                   int64_t workspace_bytes, void* stream);
int xva_fp_backward(const xva_fp_dims* d, const float* params, float* grads, const xva_fp_batch* batch, void* workspace,
                    int64_t workspace_bytes, void* stream);

/* Inference (FastPitch.infer, python/fastpitch1_1/fastpitch/model.py:426-481: predicted durations, pitch and energy; no speaker
 * embedding, no pitch transform).  Two calls because the mel length is data dependent: the host reads dec_lens, sizes the
 * decoder workspace for Tm = max(dec_lens) and calls the second half.
 *   encode: dims {B, Tt, Tm ignored, stage ignored, compute, dropout ignored}; batch needs text, in_lens, pos_table.
 *           enc_cond_out (B, Tt+2, 384) activation dtype; durs_out (B, Tt) int32 = (long)(dur_pred * pace + 0.5);
 *           dec_lens_out (B) int32; dur_pred / pitch_pred / energy_pred (B, Tt) fp32.
 *   decode: dims {B, Tt, Tm, ., compute, .}; mel_out (B, 80, Tm) fp32 (the reference's permuted layout; frames >= dec_lens hold proj(0) = bias
 *           exactly like the reference's masked decoder output). */
int xva_fp_infer_encode(const xva_fp_dims* d, const float* params, const xva_fp_batch* batch, float pace, float max_duration, void* workspace,
                        int64_t workspace_bytes, void* enc_cond_out, int32_t* durs_out, int32_t* dec_lens_out, float* dur_pred_out,
                        float* pitch_pred_out, float* energy_pred_out, void* stream);
int xva_fp_infer_decode(const xva_fp_dims* d, const float* params, const void* enc_cond, const int32_t* durs, const float* pos_table,
                        void* workspace, int64_t workspace_bytes, float* mel_out, void* stream);
int xva_fp_infer_finish(const float* dur_pad, const float* pitch_pad, const float* energy_pad, const int32_t* lens, float pace, int B, int Tt,
                        int32_t* durs, int32_t* dec_lens, float* dur_out, float* pitch_out, float* energy_out, void* stream);
int xva_fp_mel_to_bct(const void* in, int dt, float* out, int B, int Tm, int C, void* stream);

/* Training stage 1 — the aligner (python/fastpitch1_1/fastpitch/model.py:296-323,346-360; attention.py:171-220; alignment.py:76-118;
 * attn_loss_function.py:20-44; loss_function.py:73-81): text embeddings and the target mel go through ConvAttention's key / query
 * projections, attn = log_softmax(-0.0005 ||q - k||^2) + log(prior + 1e-8); soft = softmax over the valid keys; monotonic
 * alignment search gives the hard durations; loss = forward-sum (CTC, blank log-prob -1) / B.  fp32 storage; `compute` picks the
 * MFMA pipe of the GEMMs.  Gradients reach attention.* and encoder.word_emb only. */
typedef struct xva_fp_align_batch {
    const int32_t* text;      /* (B, Tt) */
    const int32_t* in_lens;   /* (B) */
    const float* mel;         /* (B, 80, Tm) zero padded */
    const int32_t* mel_lens;  /* (B) */
    const float* attn_prior;  /* (B, Tm, Tt) beta-binomial prior, zero padded (data_function.py:84-94,600-609) */
} xva_fp_align_batch;
int64_t xva_fp_align_workspace_bytes(const xva_fp_dims* d);
/* attn_soft_out / attn_logprob_out: (B, Tm, Tt) fp32 or NULL; durs_out (B, Tt) int32 hard durations; loss_out: 1 float (overwritten) */
int xva_fp_align_forward(const xva_fp_dims* d, const float* params, const xva_fp_align_batch* batch, void* workspace, int64_t workspace_bytes,
                         float* attn_soft_out, float* attn_logprob_out, int32_t* durs_out, float* loss_out, void* stream);
/* accumulates grad_scale * d(loss)/d(params) into grads; consumes the state xva_fp_align_forward left in the workspace (call once
 * per forward) */
int xva_fp_align_backward(const xva_fp_dims* d, const float* params, float* grads, const xva_fp_align_batch* batch, void* workspace,
                          int64_t workspace_bytes, float grad_scale, void* stream);
/* single kernels of the aligner (align_ops.hip) */
int xva_al_embed(const int32_t* ids, const float* emb, float* out, int B, int T, int C, void* stream);
int xva_al_mel_to_tm(const float* mel, float* out, int B, int C, int Tm, void* stream);
int xva_al_sqnorm(const float* X, float* out, int64_t rows, int C, void* stream);
int xva_al_attn_rows(const float* S, const float* qn, const float* kn, const float* prior, const int32_t* in_lens, float* logprob, float* soft,
                     float* lse1, float* lse2, int B, int Tm, int Tt, int ld, void* stream);
int xva_al_mas(const float* soft, const int32_t* in_lens, const int32_t* mel_lens, uint8_t* choice, int32_t* durs, int B, int Tm, int Tt, int ld,
               void* stream);
int xva_al_ctc(const float* logprob, const float* lse2, const int32_t* in_lens, const int32_t* mel_lens, float* alpha, float* beta,
               float* dlogprob, float* loss, int B, int Tm, int Tt, int ld, float gscale, void* stream);
int xva_al_logsoftmax_bwd(const float* S, const float* qn, const float* kn, const float* lse1, float* G, float* colsum, int B, int Tm, int Tt,
                          int ld, void* stream);
int xva_al_dk_fix(float* dk, const float* k, const float* colsum, int B, int Tt, int C, float scale, void* stream);

/* Data-parallel overlap: while a callback is registered (per host thread), xva_fp_backward_ex calls it right after it has recorded bucket i's event —
 * i.e. while the host is still issuing backward — so that the host can enqueue that bucket's wait + all-reduce on its exchange stream at once
 * (a wait issued after the whole backward has been issued resolves only when the recording lane has drained).  NULL unregisters. */
void xva_fp_set_bucket_callback(void (*cb)(int bucket, void* user), void* user);
/* the same for xva_hg_disc_backward_d_ex / xva_hg_generator_backward_ex (and their xVAPitch variants): bucket indices of the call in progress */
void xva_hg_set_bucket_callback(void (*cb)(int bucket, void* user), void* user);
/* Data-parallel overlap: gradient buckets (contiguous flat ranges, in backward completion order) and a backward
 * that records one hipEvent_t per bucket as it completes; the host starts that bucket's RCCL all-reduce on a side
 * stream (replaces nn.DataParallel's reduce_add_coalesced, python/fastpitch1_1/xva_train.py:48-53,465-466). */
int xva_fp_num_buckets(void);
int xva_fp_bucket_range(int i, int64_t* begin, int64_t* end);
int xva_fp_backward_ex(const xva_fp_dims* d, const float* params, float* grads, const xva_fp_batch* batch, void* workspace,
                       int64_t workspace_bytes, void* const* bucket_events, void* stream);
void* xva_event_create(void);
int xva_event_destroy(void* event);
int xva_event_record(void* event, void* stream);
int xva_stream_wait_event(void* stream, void* event);

/* The denominators of FastPitchLoss's masked means (python/fastpitch1_1/loss_function.py:70-73,98-100) from the targets alone: den2[0] =
 * #(mel_tgt != 0) (stages 3 / 4, else 0), den2[1] = sum of the token lengths.  Data-parallel ranks all-reduce these two floats UNDER the forward
 * pass and overwrite acc[1] / acc[3] of xva_fp_loss_partials with the global values before xva_fp_loss_grads: the losses a rank then reports are
 * its share of the global means (their SUM over ranks is the global loss — a reduction that is off the critical path). */
int xva_fp_loss_denominators(int stage, const float* mel_tgt, const int32_t* in_lens, float* den2, int B, int Tt, int Tm, void* stream);
/* FastPitchLoss in two phases (so data-parallel ranks can all-reduce `acc` in between: global normalisation).
 * mel_out / d_mel are activation-dtype tensors (dt); the token-level predictions and targets are fp32. */
int xva_fp_loss_partials(int stage, int dt, const void* mel_out, const float* mel_tgt, const float* pitch_pred, const float* pitch_tgt,
                         const float* energy_pred, const float* energy_tgt, const float* log_dur_pred, const int32_t* durs,
                         const int32_t* in_lens, float* acc, int B, int Tt, int Tm, void* stream);
int xva_fp_loss_grads(int stage, int dt, const void* mel_out, const float* mel_tgt, const float* pitch_pred, const float* pitch_tgt,
                      const float* energy_pred, const float* energy_tgt, const float* log_dur_pred, const int32_t* durs,
                      const int32_t* in_lens, const float* acc, float* losses_out, void* d_mel, float* d_pitch, float* d_energy,
                      float* d_logdur, int B, int Tt, int Tm, float grad_scale, float dur_w, float pitch_w, float energy_w,
                      void* stream);

/* Individual kernels of the path.  Activation tensors are `void*` of dtype dt (XVA_F32 / XVA_BF16; the element-wise ones — cast, colsum, add — also XVA_F16); parameters, statistics
 * and token-level scalars are fp32.  Dropout masks are a pure function of (seed, stream_id, element index). */
int xva_fp_embed_fwd(const int32_t* ids, const float* emb, const float* pos, void* out, int dt, int B, int T, int C, void* stream);
int xva_fp_embed_bwd(const int32_t* ids, const void* dX, int dt, float* dEmb, int B, int T, int C, void* stream);
int xva_fp_softmax_fwd(void* S, void* P_dropped, int dt, const int32_t* lens, int B, int Tp, int64_t Ts, float p_drop, uint64_t seed,
                       uint32_t stream_id, void* stream);
int xva_fp_softmax_bwd(const void* P, void* dP, int dt, int B, int Tp, int64_t Ts, float scale, float p_drop, uint64_t seed,
                       uint32_t stream_id, void* stream);
/* PAIR OUTPUTS AND THE fp16-OPERAND MODE (round 6).  Every `*_pair` / `*_pairs` / `*_planes` entry point below (and xva_split_bf16) takes the distance between
 * the hi and the lo plane in ELEMENTS.  A distance of 0 selects the single-plane form: the tensor is ONE IEEE-half tensor (XVA_F16) at the hi plane's address and
 * no lo plane is read or written — the operand format of FastPitch's compute mode 2, whose products are plain one-pass xva_gemm calls on XVA_F16 operands. */
/* LayerNorm on fp32 rows that ALSO leaves its output as a split-bf16 pair (hi plane at *_pair, lo plane pair_plane ELEMENTS after it; rows 0 .. rows - 1 of each
 * plane — guard rows are the caller's): the operand of the next `planes` product without a split launch.  Backward: the pair is the gradient entering the
 * dropout-ed branch (dX * m_out; dX itself when p_out == 0); dXm (fp32) may be null. */
int xva_fp_layernorm_fwd_pair(const void* X, const float* gamma, const float* beta, void* Y, void* y_pair, int64_t pair_plane, float* mean, float* rstd,
                              int64_t rows, int C, int mask_mode, const int32_t* lens, int Tp, void* stream);
int xva_fp_layernorm_bwd_pair(const void* dY, const void* X, const float* mean, const float* rstd, const float* gamma, void* dX, void* dXm, void* dx_pair,
                              int64_t pair_plane, float* dgamma, float* dbeta, int64_t rows, int C, int mask_mode, const int32_t* lens, int Tp, float p_out,
                              uint64_t seed_out, uint32_t stream_out, void* stream);
void xva_fp_set_ln_pairs(int on);
/* fp32 softmax rows with split-bf16 PAIR outputs (the operands of xva_gemm `planes`, fp32 mode with split products): forward writes P fp32 in place over S and
 * the (dropped) copy as a pair (hi plane at Pd_pair, lo plane pair_plane elements after it); backward reads P / dP fp32 and writes dS as a pair. */
int xva_fp_softmax_fwd_pairs(void* S, void* Pd_pair, int64_t pair_plane, const int32_t* lens, int B, int Tp, int64_t Ts, float p_drop, uint64_t seed,
                             uint32_t stream_id, void* stream);
int xva_fp_softmax_bwd_pairs(const void* P, const void* dP, void* dS_pair, int64_t pair_plane, int B, int Tp, int64_t Ts, float scale, float p_drop, uint64_t seed,
                             uint32_t stream_id, void* stream);
/* Fused single-head attention (d_head = 64) on bf16 tensors: qkv (B, Tp, 192) = [Q | K | V], keys 1..lens[b] valid.
 * Forward writes av (B, Tp, 64) and the per-row logsumexp; backward writes d_qkv (B, Tp, 192) (dscratch: B * Tp floats).
 * Same mathematics and dropout masks as the unfused xva_gemm + xva_fp_softmax_* chain (transformer.py:109-130). */
int xva_fp_attention_fwd(const void* qkv, const int32_t* lens, void* av, float* lse, int B, int Tp, float scale, float p_drop,
                         uint64_t seed, uint32_t stream_id, void* stream);
int xva_fp_attention_bwd(const void* qkv, const void* av, const void* d_av, const float* lse, float* dscratch, const int32_t* lens,
                         void* d_qkv, int B, int Tp, float scale, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
/* The same kernels on split-bf16 PAIRS (fp32 mode with split products; include/xva_gemm.h `planes`): qkv / av / d_av / d_qkv point at the hi planes, the lo planes
 * sit `*_plane` ELEMENTS after them; every product is hi.hi + hi.lo + lo.hi in fp32 and the outputs leave as pairs.  Replaces the unfused scores -> softmax ->
 * P V chain of that mode (two T x T fp32 tensors per layer and direction).  xva_fp_set_att_flash(0) restores the unfused chain. */
int xva_fp_attention_fwd_pairs(const void* qkv, int64_t qkv_plane, const int32_t* lens, void* av, int64_t av_plane, float* lse, int B, int Tp, float scale,
                               float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
int xva_fp_attention_bwd_pairs(const void* qkv, int64_t qkv_plane, const void* av, int64_t av_plane, const void* d_av, int64_t d_av_plane, const float* lse,
                               float* dscratch, const int32_t* lens, void* d_qkv, int64_t d_qkv_plane, int B, int Tp, float scale, float p_drop, uint64_t seed,
                               uint32_t stream_id, void* stream);
void xva_fp_set_att_flash(int on);
int xva_fp_layernorm_fwd(const void* X, const float* gamma, const float* beta, void* Y, int dt, float* mean, float* rstd, int64_t rows,
                         int C, int mask_mode, const int32_t* lens, int Tp, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
int xva_fp_layernorm_bwd(const void* dY, const void* X, const float* mean, const float* rstd, const float* gamma, void* dX, void* dX_masked,
                         int dt, float* dgamma, float* dbeta, int64_t rows, int C, int mask_mode, const int32_t* lens, int Tp, int relu_gate,
                         float p_in, uint64_t seed_in, uint32_t stream_in, float p_out, uint64_t seed_out, uint32_t stream_out,
                         const float* outer_d, const float* outer_w, void* stream);  /* outer_*: dY[r][c] = outer_d[r] * outer_w[c] (fp32), dY ignored */
int xva_fp_colsum(const void* X, int dt, float* out, int64_t rows, int C, int64_t ld, void* stream);
int xva_fp_avg_pitch(const float* dense, const int32_t* durs, float* avg_out, int B, int Tt, int Tm, int log1p_, void* stream);
int xva_fp_lenreg_map(const int32_t* durs, int32_t* tok, int32_t* tstart, int32_t* dec_lens, int B, int Tt, int Tm, float pace,
                      void* stream);
int xva_fp_cond_add_fwd(const void* in, const float* s, const float* w, const float* bias, void* out, int dt, const int32_t* lens,
                        int B, int Tp, int C, void* stream);
int xva_fp_cond_add_bwd(const void* dOut, int dt, const float* s, float* dw, float* db, const int32_t* lens, int B, int Tp, int C,
                        void* stream);
int xva_fp_lenreg_fwd(const void* enc, const int32_t* tok, const int32_t* dec_lens, const float* pos, void* out, int dt, int B, int Tt,
                      int Tm, int C, void* stream);
int xva_fp_lenreg_bwd(const void* dOut, const int32_t* tstart, const int32_t* dec_lens, void* dEnc, int dt, int B, int Tt, int Tm, int C,
                      int accumulate, void* stream);
int xva_fp_outer(const float* s, const float* w, void* out, int dt, int64_t rows, int C, void* stream);
int xva_fp_rowscale_colsum(const void* X, int dt, const float* s, float* out, int64_t rows, int C, void* stream);
int xva_fp_dur_from_log(const float* logd, float* out, int n, float max_dur, void* stream);
int xva_cast_f32(const float* src, void* dst, int dt, int64_t n, void* stream);
int xva_cast_to_f32(const void* src, int dt, float* dst, int64_t n, void* stream);
/* Transposed bf16 shadow of n (<= 16) tap-major k = 3 convolution weights: src = params + src_off[i], fp32 [Cout][3][Cin]; dst = out + dst_off[i]
 * (elements), bf16 [Cin][3][Cout] with dst[c][m][o] = src[o][2 - m][c] — the weight of the convolution that maps d(output) to d(input), so the
 * backward-data product of python/fastpitch1_1/fastpitch/transformer.py:59-77 (autograd of CoreNet's second Conv1d) runs the NT main loop. */
int xva_fp_wt_transpose3(const float* params, void* out, const int64_t* src_off, const int64_t* dst_off, int n, int Cout, int Cin, void* stream);
/* the same, each copy written as a split-bf16 pair (hi plane, lo plane `plane` elements after it): the weight operand of the backward-data products of the
 * fp32 mode's split-products feed-forward path (xva_gemm `planes`) */
int xva_fp_wt_transpose3_planes(const float* params, void* out, const int64_t* src_off, const int64_t* dst_off, int n, int Cout, int Cin, int64_t plane, void* stream);
/* fp32 tensor -> split-bf16 pair: dst[i] = bf16(src[i]), dst[plane + i] = bf16(src[i] - dst[i]) (n, plane multiples of 8): operands of xva_gemm's `planes`
 * products, which keep ~16 mantissa bits through the bf16 matrix pipe (include/xva_gemm.h). */
int xva_split_bf16(const float* src, void* dst, int64_t plane, int64_t n, void* stream);
/* zero n small byte spans (4-byte aligned, multiples of 4 bytes) in one launch per 40 spans: the guard rows of pairs that live in fp32 slots */
int xva_zero_spans(void* const* ptrs, const int64_t* bytes, int n, void* stream);
/* dst += src over n (even) elements of the activation dtype: joins the gradient contributions that the temporal predictors' backward
 * (python/fastpitch1_1/fastpitch/model.py:394-418) produces on its own stream into d(encoder output) */
int xva_fp_add_act(void* dst, const void* src, int dt, int64_t n, void* stream);

/* ------------------------------------------------------------------------ optimizers ---- */
/* (scal: >= 4 device floats: [1] the applied clip coefficient x inv_scale, [2] the pre-clip global gradient norm, [3] 1 when that norm was not finite and the step
 * was SKIPPED on the device — moments and parameters untouched, as torch.cuda.amp.GradScaler.step does for the reference's fp16 path — else 0.) */
/* Fused multi-tensor LAMB over the flat buffers = torch.nn.utils.clip_grad_norm_(.., max_grad_norm) followed by
 * Lamb.step (python/fastpitch1_1/lamb.py:40-106; call site xva_train.py:853-862).  Chunk descriptors come from
 * xva_opt_build_chunks (host) and list only the tensors that received a gradient this stage (Lamb skips
 * p.grad is None). */
int xva_opt_chunk_size(void);
int64_t xva_opt_build_chunks(const int64_t* offsets, const int64_t* numels, const int32_t* active, int n, int32_t* ctid,
                             int64_t* cstart, int32_t* clen, int64_t cap);
int xva_lamb_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t total_floats, const int32_t* ctid,
                  const int64_t* cstart, const int32_t* clen, int64_t n_chunks, int n_tensors, float* scal, float* norms, float lr,
                  float beta1, float beta2, float eps, float weight_decay, float max_grad_norm, float inv_scale, void* stream);
/* torch.optim.AdamW step over a flat buffer (python/hifigan/xva_train.py:298-300,498,515). */
int xva_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step, float lr,
                   float beta1, float beta2, float eps, float weight_decay, void* stream);

/* --------------------------------------------------------------------- HiFi-GAN v1 engine ---- */
/* Replaces the PyTorch graph of HiFiTrainer.iteration (python/hifigan/xva_train.py:451-567): Generator forward /
 * backward (python/hifigan/models.py:81-128), MultiPeriodDiscriminator + MultiScaleDiscriminator forward / backward
 * (:140-260), GAN + feature-matching losses (:263-294).  Parameters are two flat fp32 buffers (which = 0 generator,
 * 1 discriminators: "mpd.*" then "msd.*", trainable tensors first, then spectral-norm buffers) in the checkpoint's
 * own tensor layouts (weight_g / weight_v / weight_orig / weight_u / weight_v).  Activations are time-major sequences
 * of dtype dims.dt; the workspace must be zero-filled once when allocated. */
typedef struct xva_hg_dims {
    int32_t B;     /* items per micro-batch */
    int32_t seg;   /* samples per item (config_v1.json segment_size = 8192); multiple of 256 */
    int32_t dt;    /* activation / MFMA dtype: XVA_F32 (exact-fp32 parity mode) or XVA_BF16 */
} xva_hg_dims;
int64_t xva_hg_param_floats(int which);
int64_t xva_hg_trainable_floats(int which);
int xva_hg_num_tensors(int which);
int xva_hg_tensor_info(int which, int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim, int64_t* shape4,
                       int32_t* kind);
int64_t xva_hg_workspace_bytes(const xva_hg_dims* d);
/* mel: (B, 80, seg/256) fp32 -> wav_out: (B, seg) fp32 (may be NULL; the waveform also stays in the workspace) */
int xva_hg_generator_forward(const xva_hg_dims* d, const float* params_g, const float* mel, void* workspace, int64_t workspace_bytes,
                             float* wav_out, void* stream);
/* d_wav: (B, seg) fp32 gradient w.r.t. the generated waveform; accumulates into grads_g */
int xva_hg_generator_backward(const xva_hg_dims* d, const float* params_g, float* grads_g, const float* d_wav, void* workspace,
                              int64_t workspace_bytes, void* stream);
/* The VITS waveform decoder of xVAPitch — python/xvapitch/hifigan.py:156-262 (HifiganGenerator) as python/xvapitch/model.py:134-149 builds it:
 * the HiFi-GAN v1 generator above with `in_channels` latent channels in (192; 256 for the `big` model), conv_pre / conv_post WITHOUT weight norm
 * (tensors conv_pre.weight / .bias, conv_post.weight — no bias), and cond_layer = Conv1d(cond_channels, 512, 1) on the speaker vector added to
 * conv_pre's output (hifigan.py:247-248).  Parameters: one flat fp32 buffer, tensors named and shaped as in the reference state_dict
 * (xva_vits_dec_tensor_info).  z: (B, in_channels, seg / 256) fp32; g: (B, cond_channels) fp32 (NULL when cond_channels == 0);
 * wav_out: (B, seg) fp32.  backward accumulates into `grads` and writes d_z (B, in_channels, seg / 256) fp32 (may be NULL).
 * The workspace (zero-filled once when allocated) holds the activations between forward and backward. */
typedef struct xva_vits_dec_dims {
    int32_t B, seg, dt;               /* as xva_hg_dims */
    int32_t in_channels;              /* multiple of 8 */
    int32_t cond_channels;            /* multiple of 4; 0: no cond_layer */
} xva_vits_dec_dims;
int64_t xva_vits_dec_param_floats(const xva_vits_dec_dims* d);
int xva_vits_dec_num_tensors(const xva_vits_dec_dims* d);
int xva_vits_dec_tensor_info(const xva_vits_dec_dims* d, int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim,
                             int64_t* shape4);
int64_t xva_vits_dec_workspace_bytes(const xva_vits_dec_dims* d);
int xva_vits_dec_forward(const xva_vits_dec_dims* d, const float* params, const float* z, const float* g, void* workspace, int64_t workspace_bytes,
                         float* wav_out, void* stream);
int xva_vits_dec_backward(const xva_vits_dec_dims* d, const float* params, float* grads, const float* g, const float* d_wav, float* d_z,
                          void* workspace, int64_t workspace_bytes, void* stream);
/* VitsDiscriminator of xVAPitch — python/xvapitch/model.py:1590-1640: nets.0 = the scale discriminator of :1548-1587 (weight norm; grouped k = 41
 * convolutions with four input channels per group, run as dense products over block-diagonal effective weights), nets.1-5 = the period
 * discriminators of python/xvapitch/hifigan.py:301-367.  One flat fp32 parameter buffer with the reference's tensor names / shapes
 * (xva_vits_disc_tensor_info).  The three passes, their loss definitions (python/xvapitch/losses.py:62-84,306-351: LSGAN discriminator / generator
 * losses, feature loss x 2) and argument conventions are those of xva_hg_disc_forward_ex / _backward_d / _backward_g below; the workspace is its own
 * (xva_vits_disc_workspace_bytes, zero-filled once when allocated). */
int64_t xva_vits_disc_param_floats(void);
int xva_vits_disc_num_tensors(void);
int xva_vits_disc_tensor_info(int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim, int64_t* shape4);
int64_t xva_vits_disc_workspace_bytes(const xva_hg_dims* d);
int xva_vits_disc_forward(const xva_hg_dims* d, float* params_d, const float* y_real, const float* y_fake, void* workspace, int64_t workspace_bytes,
                          float* losses, int loss_mask, void* stream);
int xva_vits_disc_backward_d(const xva_hg_dims* d, float* params_d, float* grads_d, const float* y_real, const float* y_fake, void* workspace,
                             int64_t workspace_bytes, void* stream);
/* feature_grad: 1 = d(generator loss + feature loss) / d y_fake ; 0 = d(generator loss) / d y_fake only — what xVAPitch's trainer back-propagates:
 * python/xvapitch/model.py:345-347 passes (fake, real) features to feature_loss(feats_real, feats_generated) (python/xvapitch/losses.py:64-72),
 * whose .detach() therefore lands on the GENERATED features, so the feature term reaches no generator parameter. */
int xva_vits_disc_backward_g(const xva_hg_dims* d, float* params_d, const float* y_real, const float* y_fake, float* d_wav, int feature_grad,
                             void* workspace, int64_t workspace_bytes, void* stream);
/* Stream lanes.  Inside one call the engines issue independent chains on side streams they own (created once per host thread, forked
 * from and joined to the caller's stream with events inside the call): HiFi-GAN — period | scale discriminators, the generator's
 * weight gradients, the three parallel resblocks of a stage; FastPitch — the weight gradients of a layer and the temporal predictors.
 * n <= 1 puts everything back on the caller's stream (per-kernel measurements, debugging); returns the previous lane count.  Every
 * tensor is still produced by the same kernels in the same order, so activations do not depend on it; sums that end in fp32 atomics
 * (losses, bias / LayerNorm-parameter gradients, the spectral-norm power iteration) differ by the order of those atomics exactly as two
 * runs on one stream do (tests/test_stream_lanes_gpu.py), and FastPitch's d(encoder output) receives the predictors' contributions
 * from a separate add kernel instead of a GEMM epilogue (one more bf16 rounding in the throughput mode).
 * env XVA_HG_STREAMS / XVA_FP_STREAMS = 1 do the same at start-up. */
int xva_hg_set_streams(int n);
int xva_fp_set_streams(int n);
/* HiFi-GAN generator forward, bf16 mode: a ResBlock1 pair (python/hifigan/models.py:41-48) of the 32 / 64-channel stages as ONE launch (csrc/conv_pair.hip).
 * 0 = two convolution launches; 1 (default) = fused, operand = the stored activated copy: bit-identical; 2 / 3 = fused from the raw block input (one tensor
 * pass less; one rounding of the negative half differs: LeakyReLU applied to the bf16 value).  Returns the previous mode.  env XVA_HG_PAIR. */
int xva_hg_set_pair_mode(int mode);
/* FastPitch bf16 mode, backward-data of the feed-forward's second Conv1d (python/fastpitch1_1/fastpitch/transformer.py:59-77 under autograd):
 * 1 (default) = through a transposed, tap-reversed bf16 copy of the weight refreshed with the parameter shadow (NT main loop), 0 = on the weight as
 * stored (NN main loop).  Changes the workspace plan: set it before xva_fp_workspace_bytes.  Returns the previous mode.  */
int xva_fp_set_bwd_nt(int mode);
/* bf16 LayerNorm rows of 384 channels (python/fastpitch1_1/fastpitch/transformer.py:75,146): 1 (default) = the forward takes four rows per wavefront with 16-byte
 * accesses, 0 = one row per wavefront (the backward always does).  Same arithmetic per row; the row sums are taken over a different lane order (fp32 rounding only).  env XVA_FP_LN4. */
int xva_fp_set_ln4(int mode);
/* FastPitch fp32 mode with split-bf16 products (xva_gemm_set_fp32_products(1); outputs / losses within north_star's 1e-3): 1 (default) = the feed-forward
 * convolutions (python/fastpitch1_1/fastpitch/transformer.py:59-77, 94 % of the FLOPs) run the direct-to-LDS kernels on split-bf16 pairs the producers
 * write (include/xva_gemm.h `planes`), 0 = every product splits its operands while staging them (rounds 3 - 4).  Same arithmetic (hi hi + hi lo + lo hi);
 * changes the workspace plan: set before xva_fp_workspace_bytes.  Returns the previous mode.  env XVA_FP_FFN_PLANES. */
int xva_fp_set_ffn_planes(int mode);
/* The plan-affecting switches (xva_fp_set_ffn_planes, xva_fp_set_bwd_nt) as one word.  xva_fp_workspace_bytes and xva_fp_slot_offset depend on them: a caller that
 * caches either must re-query (and re-zero its workspace: the guard rows move) when this value changes. */
int xva_fp_plan_knobs(void);
/* bf16 mode, the tail of MultiHeadAttn.forward (python/fastpitch1_1/fastpitch/transformer.py:132-147): sum1 = x + dropout(AV Wo^T), y1 = LayerNorm(sum1) * rowmask
 * as ONE kernel (av (rows, 64), w_bf16 (384, 64), x / sum1 / y1 (rows, 384) bf16; mean / rstd fp32 per row; the dropout mask of the GEMM epilogue it replaces:
 * hash(seed, stream_id, row * 384 + col)).  xva_fp_set_onet_fused(0) / env XVA_FP_ONET_FUSED=0 put the engine back on the GEMM + LayerNorm pair. */
int xva_fp_onet_ln_fwd(const void* av, const void* w_bf16, const void* x, const float* gamma, const float* beta, void* sum1, void* y1, float* mean, float* rstd,
                       int64_t rows, int mask_mode, const int32_t* lens, int Tp, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
/* The same block in the fp16-operand mode (fp32 residual stream): av (rows, 64) and w (384, 64) IEEE half; x / sum1 / y1 (rows, 384) fp32, nothing rounded before the
 * row statistics; y1_f16 = the half copy of y1 that conv1 reads (what xva_fp_layernorm_fwd_pair stores at plane distance 0).  Same dropout mask. */
int xva_fp_onet_ln_fwd_f16(const void* av_f16, const void* w_f16, const float* x, const float* gamma, const float* beta, float* sum1, float* y1, void* y1_f16, float* mean,
                           float* rstd, int64_t rows, int mask_mode, const int32_t* lens, int Tp, float p_drop, uint64_t seed, uint32_t stream_id, void* stream);
int xva_fp_set_onet_fused(int mode);
/* Test / diagnostics: byte offset (into the caller's workspace) and geometry {nseq, T, C, padF, padB} of an activation tensor the last
 * forward stored, time-major (nseq, padF + T + padB, C) in the activation dtype.  kind: 0 mel input, 1 conv_pre output, 2 u[i0] (ups
 * output), 3 lrelu(u[i0]), 4 xt1[resblock i0][m i1] (= lrelu(c1(lrelu(x))), models.py:43-45), 5 / 6 x after block m and its lrelu copy,
 * 7 xs[i0] (mean of the three resblocks, :118-123), 8 the waveform, 9 MPD period i0 tensor i1 (0 = folded input), 10 MSD scale i0 set i1
 * tensor i2. */
int xva_hg_slot(const xva_hg_dims* d, int kind, int i0, int i1, int i2, int64_t* off_bytes, int32_t* geom5);
/* Data-parallel variants (the reference's nn.DataParallel reduce step, python/fastpitch1_1/xva_train.py:48-53; HiFi-GAN's
 * unused dist_config in python/hifigan/config_v1.json:32-36): the flat gradient buffer `which` splits into
 * xva_hg_num_buckets(which) contiguous buckets listed in backward-completion order; the *_ex backward records
 * bucket_events[i] (xva_event_create handles, entries or the array may be NULL) on `stream` as soon as bucket i's
 * gradients are final, so the caller can all-reduce it on a side stream under the rest of backward. */
int xva_hg_num_buckets(int which);
int xva_hg_bucket_range(int which, int i, int64_t* begin, int64_t* end);
int xva_hg_generator_backward_ex(const xva_hg_dims* d, const float* params_g, float* grads_g, const float* d_wav, void* workspace,
                                 int64_t workspace_bytes, void* const* bucket_events, void* stream);
int xva_hg_disc_backward_d_ex(const xva_hg_dims* d, float* params_d, float* grads_d, const float* y_real, const float* y_fake,
                              void* workspace, int64_t workspace_bytes, void* const* bucket_events, void* stream);
/* MPD + MSD on (real, fake) waveforms (B, seg) fp32 (xva_train.py:488-493 / 506-507).  `losses` (device, 4 floats or
 * NULL) receives {discriminator loss, generator LSGAN loss, feature-matching loss, -} (models.py:263-294).  params_d is
 * mutable: each pass of the spectral-norm discriminator advances weight_u / weight_v by one power iteration. */
int xva_hg_disc_forward(const xva_hg_dims* d, float* params_d, const float* y_real, const float* y_fake, void* workspace,
                        int64_t workspace_bytes, float* losses, void* stream);
/* Same with a loss selection: bit 0 = discriminator loss (all the D step needs, xva_train.py:488-493), bit 1 = generator LSGAN +
 * feature-matching losses (the G step, :506-512; the only part that reads every feature map).  Unselected entries stay 0.
 * Bit 2: the effective (weight-normalised, activation-dtype) weights in `workspace` were prepared from exactly these params_d by the previous forward on it
 * (nothing wrote params_d since; the spectral-norm buffers do not count): the reparametrisation pass is skipped.  A wrong promise gives stale weights. */
int xva_hg_disc_forward_ex(const xva_hg_dims* d, float* params_d, const float* y_real, const float* y_fake, void* workspace,
                           int64_t workspace_bytes, float* losses, int loss_mask, void* stream);
/* D step (xva_train.py:494-495): accumulates d(loss_disc_s + loss_disc_f)/d(params_d) into grads_d. */
int xva_hg_disc_backward_d(const xva_hg_dims* d, float* params_d, float* grads_d, const float* y_real, const float* y_fake,
                           void* workspace, int64_t workspace_bytes, void* stream);
/* G step (xva_train.py:506-513): d_wav (B, seg) = d(loss_gen_f + loss_gen_s + loss_fm_f + loss_fm_s)/d(y_fake). */
int xva_hg_disc_backward_g(const xva_hg_dims* d, float* params_d, const float* y_real, const float* y_fake, float* d_wav,
                           void* workspace, int64_t workspace_bytes, void* stream);

/* torch.nn.utils.weight_norm (old API, dim 0) of ONE conv weight: eff (tap-major [D0][k * D1], dtype dt) = g * v / ||v||, norm[D0] saved;
 * kind 0 = Conv1d v (Cout, Cin_g, k), kind 1 = ConvTranspose1d (also writes effB).  bwd: dv / dg += from the fp32 gradient of eff.
 * (HiFi-GAN python/hifigan/models.py:21-108, WaveNet python/xvapitch/wavenet.py:62-82.) */
int xva_hg_weight_norm_fwd(const float* v, const float* g, void* eff, void* effB, float* norm, int dt, int kind, int D0, int D1, int k, int s,
                           int pconv, void* stream);
int xva_hg_weight_norm_bwd(const float* dW, const float* v, const float* g, const float* norm, float* dv, float* dg, int kind, int D0, int D1,
                           int k, void* stream);
/* out[c] += scale * sum_r X[r][c] (bias gradients; X (rows, C) in dtype dt). */
int xva_hg_colsum(const void* X, int dt, float* out, int64_t rows, int C, float scale, void* stream);

/* ------------------------------------------------------------ xVAPitch-only blocks (first set) ---- */
/* Kernels of the VITS-style xVAPitch model that FastPitch / HiFi-GAN do not share.  Sequence tensors are
 * time-major (B, Tp = pad + T + pad, C) in dtype dt (XVA_F32 / XVA_BF16) with structurally zero pad rows; the WaveNet
 * convolutions themselves run on xva_gemm (host: xva-trainer_amd/xvapitch/wn.py). */
/* fused_add_tanh_sigmoid_multiply (python/xvapitch/wavenet.py:5-12): acts (B*Tp, H) = tanh(a[:, :H] + g) * sigmoid(a[:, H:] + g);
 * a (B*Tp, 2H); g (B, 2H; row stride g_ld) fp32 per-item conditioning broadcast over time, or NULL.  bwd: d_a from d_acts. */
int xva_wn_gate_fwd(const void* a, const float* g, int64_t g_ld, void* acts, int dt, int B, int Tp, int H, void* stream);
int xva_wn_gate_bwd(const void* a, const float* g, int64_t g_ld, const void* d_acts, void* d_a, int dt, int B, int Tp, int H, void* stream);
/* WN.forward's residual / skip split (wavenet.py:103-108): x_next = (x + rs[:, :H]) * mask, out += rs[:, H:] (last layer: out += rs, rs
 * (rows, H)); mask = rows pad <= t' < pad + lens[b].  bwd: d_rs from (d_x, d_out). */
int xva_wn_res_skip_fwd(const void* rs, const void* x, void* x_next, void* out, int dt, int B, int Tp, int pad, int H, int last, const int32_t* lens,
                        void* stream);
int xva_wn_res_skip_bwd(const void* d_x, const void* d_out, void* d_rs, int dt, int B, int Tp, int pad, int H, int last, const int32_t* lens,
                        void* stream);
/* The layer loop of WN (python/xvapitch/wavenet.py:84-109) as two engine calls over xva_gemm and the four kernels above.  Sequences are the time-major stores of
 * xva-trainer_amd/xvapitch/wn.py:Seq: (2 * 32 guard rows + B * (8 + T + 8)) x C elements of dtype dt, pad / guard rows zero.  x: the stack's (masked) input, H columns;
 * out: H columns, ZERO on entry (the skip sum accumulates into it); gc: (B, 2 H L) fp32 = cond_layer(g) or NULL; tab: host array of L * XVA_XVP_WN_PER_LAYER device
 * pointers — per layer the in_layer's EFFECTIVE weight (2H, k H) tap-major in dt, its bias (fp32), the res_skip layer's effective weight (2H | H for the last, H), bias,
 * then the fp32 buffers the backward ACCUMULATES into: d(in weight) (2H, k H), d(in bias), d(res_skip weight), d(res_skip bias).  backward: d_out (H columns; rows
 * t >= len may hold anything), d_x: ZERO on entry, returns d(input); d_gc (B, 2 H L) fp32 accumulated (with gc).  With XVA_XVP_WN_LANE=1 the weight-gradient products, their bias sums and
 * the conditioning gradient run on an engine-owned side stream, joined to `stream` before the call returns (default: everything on `stream`).
 * workspace: xva_xvp_wn_workspace_bytes(), ZERO-FILLED before forward, handed unchanged to backward. */
typedef struct xva_xvp_wn_dims {
    int32_t B, T, H, k, rate, L;   /* hidden channels, kernel size, dilation rate, layers (<= 32) */
    int32_t dt;                    /* storage of the sequences and effective weights: 0 fp32, 1 bf16 */
    int32_t compute;               /* 0 exact fp32 products, 1 bf16 MFMA */
} xva_xvp_wn_dims;
#define XVA_XVP_WN_PER_LAYER 8
int64_t xva_xvp_wn_workspace_bytes(const xva_xvp_wn_dims* d);
int xva_xvp_wn_forward(const xva_xvp_wn_dims* d, const void* const* tab, const void* x, void* out, const float* gc, const int32_t* lens, void* workspace,
                       int64_t workspace_bytes, void* stream);
int xva_xvp_wn_backward(const xva_xvp_wn_dims* d, void* const* tab, const void* x, const void* d_out, void* d_x, const float* gc, float* d_gc, const int32_t* lens,
                        void* workspace, int64_t workspace_bytes, void* sk_ws, int64_t sk_ws_bytes, void* stream);
/* maximum_path (python/xvapitch/util.py:14-53): value (B, t_x, t_y) fp32, x_lens / y_lens (B) -> path (B, t_x, t_y) fp32 of 0 / 1,
 * on the device (the reference runs it in numpy on the CPU each step).  workspace: xva_maximum_path_workspace_bytes. */
int64_t xva_maximum_path_workspace_bytes(int B, int t_x, int t_y);
int xva_maximum_path(const float* value, const int32_t* x_lens, const int32_t* y_lens, float* path, void* workspace, int64_t workspace_bytes,
                     int B, int t_x, int t_y, void* stream);
/* segment (util.py:166-178): out (B, C, S) = x[b, :, idx[b] : idx[b] + S] of x (B, C, T); bwd scatters d_out back (zero elsewhere). */
int xva_segment_fwd(const float* x, const int64_t* idx, float* out, int B, int C, int T, int S, void* stream);
int xva_segment_bwd(const float* d_out, const int64_t* idx, float* d_x, int B, int C, int T, int S, void* stream);
/* VitsGeneratorLoss.kl_loss (python/xvapitch/losses.py:87-104) on (B, H, T) fp32 tensors, mask (B, 1, T): acc2[0] += sum(kl * mask),
 * acc2[1] += sum(mask) (zero acc2 first; loss = acc2[0] / acc2[1]); kl_sample_wise (B, H, T) optional.  bwd: gradients of
 * gscale * loss (any output may be NULL). */
int xva_kl_loss_fwd(const float* z_p, const float* logs_q, const float* m_p, const float* logs_p, const float* mask, float* kl_sample_wise,
                    float* acc2, int B, int H, int T, void* stream);
int xva_kl_loss_bwd(const float* z_p, const float* m_p, const float* logs_p, const float* mask, const float* acc2, float gscale, float* d_z_p,
                    float* d_logs_q, float* d_m_p, float* d_logs_p, int B, int H, int T, void* stream);
/* Zero the rows of a time-major sequence outside [pad, pad + lens[b]) (`* x_mask` after a biased convolution). */
int xva_seq_mask(void* x, int dt, int B, int Tp, int pad, int C, const int32_t* lens, void* stream);
/* Per-item column sums of a time-major sequence (B, Tp, C): out[b * out_stride + c] += sum_t x[b][t][c] — the gradient of WN's conditioning
 * term (python/xvapitch/wavenet.py:91-97), one launch for the batch. */
int xva_seq_item_colsum(const void* x, int dt, float* out, int B, int Tp, int C, int64_t out_stride, void* stream);
/* ResidualCouplingBlock with mean_only=True (python/xvapitch/model.py:1519-1535): out (B, Ch, T) = stats + x1 * mask (forward) or
 * (x1 - stats) * mask (reverse); stats = the masked `post` output as a time-major sequence (B, pad + T + pad, Ch). */
int xva_coupling_mean_only(const void* stats, const float* x1, float* out, int dt, int B, int Ch, int T, int pad, const int32_t* lens, int reverse,
                           void* stream);
int xva_coupling_mean_only_bwd(const float* d_out, float* d_x1, void* d_stats, int dt, int B, int Ch, int T, int pad, const int32_t* lens, int reverse,
                               void* stream);
/* PosteriorEncoder's sampling (python/xvapitch/model.py:1470-1475): stats = masked `proj` output (B, pad + T + pad, 2 Co) = [mean | log_scale];
 * z = (mean + eps * exp(log_scale)) * mask with eps (B, Co, T) the caller's N(0, 1) draw; mean / logs (B, Co, T).  bwd: d_stats from
 * d_z / d_mean / d_logs (each may be NULL); pad rows of d_stats must already be zero. */
int xva_posterior_sample(const void* stats, const float* eps, float* z, float* mean, float* logs, int dt, int B, int Co, int T, int pad,
                         const int32_t* lens, void* stream);
int xva_posterior_sample_bwd(const void* stats, const float* eps, const float* d_z, const float* d_mean, const float* d_logs, void* d_stats, int dt, int B,
                             int Co, int T, int pad, const int32_t* lens, void* stream);
/* (B, C, T) fp32 <-> time-major sequence (B, pad + T + pad, C) in dt; to_seq zeroes pads and positions t >= lens[b] (lens may be NULL);
 * to_bct overwrites or (accumulate) adds into x. */
int xva_bct_to_seq(const float* x, void* seq, int dt, int B, int C, int T, int pad, const int32_t* lens, void* stream);
int xva_seq_to_bct(const void* seq, float* x, int dt, int B, int C, int T, int pad, int accumulate, void* stream);

/* RelativePositionMultiHeadAttention.attention of the xVAPitch text encoder (python/xvapitch/glow_tts.py:173-292): scaled dot-product
 * scores + the relative-key term (window w), masked_fill(-1e4) outside the item's length, softmax, P V + the relative-value term.
 * q / k / v / out: fp32 activation matrices, item b's token t in row b * Tp + pad + t, head h at columns h*dk ..; emb_*: (Hr, 2w + 1, dk) with
 * Hr = 1 (heads share) or H; P: (B, H, T, T), the softmax, kept for the backward.  drop_p > 0: nn.Dropout on the attention weights (:204) —
 * both products use dropout(P); element (b, h, i, j) is index ((b H + h) T + i) T + j of dropout site drop_stream under drop_seed (the keyed
 * hash of csrc/xva_common.h xva_dropout_scale; the backward is given the same three values). */
int xva_relattn_fwd(const float* q, const float* k, const float* v, int64_t ld, const float* emb_k, const float* emb_v, const int32_t* lens, float* P,
                    float* out, int64_t ld_out, int B, int T, int H, int dk, int w, int Hr, int Tp, int pad, float drop_p, uint64_t drop_seed,
                    uint32_t drop_stream, void* stream);
/* nn.Dropout on a contiguous tensor of n elements (dt: 0 fp32, 1 bf16): y[i] = x[i] * (0 | 1 / (1 - p)), decided by the keyed hash of
 * (seed, site, i); applying it to a gradient with the same (p, seed, site) is the backward.  y == x allowed. */
int xva_dropout_apply(const void* x, void* y, int dt, int64_t n, float p, uint64_t seed, uint32_t site, void* stream);
/* dS: (B, H, T, T) scratch; dq / dk / dv are written, d_emb_k / d_emb_v accumulated into */
int xva_relattn_bwd(const float* dO, int64_t ld_do, const float* q, const float* k, const float* v, int64_t ld, const float* emb_k, const float* emb_v,
                    const int32_t* lens, const float* P, float* dS, float* dq, float* dk_out, float* dv_out, int64_t ld_d, float* d_emb_k, float* d_emb_v,
                    int B, int T, int H, int dk, int w, int Hr, int Tp, int pad, float drop_p, uint64_t drop_seed, uint32_t drop_stream, void* stream);
/* LayerNorm2 (glow_tts.py:34-56): layer_norm over the C channels of each row, any C; the backward accumulates into dgamma / dbeta */
int xva_ln_rows_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows, int C, float eps, void* stream);
int xva_ln_rows_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx, float* dgamma, float* dbeta,
                    int64_t rows, int C, void* stream);

/* RelativePositionTransformer (python/xvapitch/glow_tts.py:59-485; the text encoder of model.py:1125-1136 and the pitch predictor of :1283-1305:
 * layer_norm_type "2", relative window w, heads share the relative embeddings, in = hidden channels) as two engine calls over the primitives above.
 * x / out / d_out / d_x: (B, C, T) / (B, Co, T) fp32 like the reference's tensors; lens (B,) = the rows of x_mask.  params / grads: host arrays of
 * L * XVA_XVP_TR_PER_LAYER (+ 2 with has_proj: proj weight (Co, C, 1), bias) DEVICE pointers in the order, per layer, conv_q w, b, conv_k w, b, conv_v w, b,
 * conv_o w, b (nn.Conv1d (C, C, 1)), emb_rel_k, emb_rel_v (1, 2w + 1, C / H), ffn conv_1 w (F, C, k), b, conv_2 w (Co_l, F, k), b, norm1 gamma, beta, norm2 gamma,
 * beta — the reference's own layouts; gradients are ACCUMULATED into `grads`.  Co == 1 (has_proj): the stack returns proj(x) of the last layer's attention
 * block (glow_tts.py:479-482); that layer's feed-forward network and second LayerNorm are not evaluated and get no gradient.  Dropout (p_drop > 0): the
 * reference's four sites per layer, site0 + 4 * layer + {0, 1, 2, 3} under `seed` (same masks as the per-primitive path; backward takes the same values).
 * workspace: xva_xvp_tr_workspace_bytes() bytes, 16-byte aligned, ZERO-FILLED before forward (structural pad rows) and handed unchanged to backward;
 * sk_ws: split-K slab scratch of the weight-gradient products (64 MiB is plenty), contents irrelevant. */
typedef struct xva_xvp_tr_dims {
    int32_t B, T;
    int32_t C, F, H, L;   /* hidden channels, feed-forward channels, heads, layers (<= 64) */
    int32_t k, w;         /* feed-forward kernel size (odd), relative attention window */
    int32_t Co;           /* out_channels: C without proj; 1 or a multiple of 4 with it */
    int32_t has_proj;
    int32_t compute;      /* 0 = exact fp32 products, 1 = bf16 MFMA on the fp32-stored operands of the projections / feed-forward convolutions */
    float p_drop;
    uint64_t seed;
    uint32_t site0;
} xva_xvp_tr_dims;
#define XVA_XVP_TR_PER_LAYER 18
int64_t xva_xvp_tr_workspace_bytes(const xva_xvp_tr_dims* d);
int xva_xvp_tr_forward(const xva_xvp_tr_dims* d, const float* const* params, const float* x, const int32_t* lens, float* out, void* workspace,
                       int64_t workspace_bytes, void* stream);
int xva_xvp_tr_backward(const xva_xvp_tr_dims* d, const float* const* params, float* const* grads, const float* d_out, const int32_t* lens, float* d_x,
                        void* workspace, int64_t workspace_bytes, void* sk_ws, int64_t sk_ws_bytes, void* stream);

/* Pieces of the stochastic duration predictor (python/xvapitch/sdp.py).  fp32, time-major (B, T, C) tensors without pad rows; `lens` = x_mask.
 * xva_dwconv_*: the depthwise dilated Conv1d of DilatedDepthSeparableConv (sdp.py:66-69,85) on x * x_mask with zero padding (k odd <= 7);
 * the backward writes dx and accumulates into dw (C, k) / db (C).  xva_gelu_*: torch's exact (erf) F.gelu (sdp.py:87,90). */
int xva_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, const int32_t* lens, int B, int T, int C, int k, int d, void* stream);
int xva_dwconv_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db, const int32_t* lens, int B, int T, int C, int k, int d,
                   void* stream);
int xva_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int xva_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);

/* DilatedDepthSeparableConv.forward (python/xvapitch/sdp.py:70-93) and its backward as two engine calls over the primitives above: (x [+ g]) -> L x {depthwise
 * conv (dilation k^i) -> LayerNorm2 -> GELU -> 1x1 conv -> LayerNorm2 -> GELU -> [Dropout, site0 + i under seed] -> + x} -> * x_mask.  x / g (may be NULL) / out / dy /
 * dx: (B, T, C) fp32; params / grads: host arrays of 8 L device pointers — per layer convs_sep weight (C, 1, k), bias, convs_1x1 weight (C, C, 1), bias, norms_1
 * gamma, beta, norms_2 gamma, beta; gradients are ACCUMULATED; dx = d x = d g.  workspace: xva_xvp_dds_workspace_bytes(), no initialisation needed, handed
 * unchanged to the backward; sk_ws: split-K slab scratch of the 1x1 weight gradients. */
typedef struct xva_xvp_dds_dims {
    int32_t B, T, C, k, L;
    float p_drop;
    uint64_t seed;
    uint32_t site0;
} xva_xvp_dds_dims;
int64_t xva_xvp_dds_workspace_bytes(const xva_xvp_dds_dims* d);
int xva_xvp_dds_forward(const xva_xvp_dds_dims* d, const float* const* params, const float* x, const float* g, const int32_t* lens, float* out, void* workspace,
                        int64_t workspace_bytes, void* stream);
int xva_xvp_dds_backward(const xva_xvp_dds_dims* d, const float* const* params, float* const* grads, const float* dy, const int32_t* lens, float* dx, void* workspace,
                         int64_t workspace_bytes, void* sk_ws, int64_t sk_ws_bytes, void* stream);

/* Two pieces of ConvFlow's backward (python/xvapitch/sdp.py:116-176).  xva_small_wgrad: d W (M, N) += dy^T x and d b (M) += column sums of dy for a 1x1 convolution
 * with small M, N (dy (rows, M), x (rows, N), fp32 FMAs, atomics) — the `proj` layer's gradients in one launch.  xva_cf_pre_bwd: the backward of `pre`
 * (Conv1d(1, H, 1) on x0) joined with the assembly of d z (rows, 2): d z[r] = (sum_c dh[r, c] w[c] + d_x0_pass[r], d_x1[r]); d_pre_w / d_pre_b (H) accumulated. H <= 256. */
/* The rest of ConvFlow's glue, one launch each (z / out / d_out / dz: (B, T, 2); x0, x1, y1, ld: (B, T); h: (B, T, H); hp: (B, T, NPp = 3K - 1 rounded up to 4) the
 * proj output, hs (B, T, NP = 3K - 1) the spline's parameters; dm (2, B, T) = [d x0 passed through | d y1]; lens = the rows of x_mask):
 * pre_fwd: x0, x1 = z[..., 0], z[..., 1]; h = b + x0 w (sdp.py:149-150).  mask_slice: hs = hp[..., :NP] * x_mask (:153-154).  post_fwd: out = [x0, y1] * x_mask,
 * ld *= x_mask (in place), ldsum[b] = sum_t ld (:170-175).  bwd_head / pad_mask: the mirror images in the backward pass. */
int xva_cf_pre_fwd(const float* z, const float* pre_w, const float* pre_b, float* x0, float* x1, float* h, int64_t rows, int H, void* stream);
int xva_cf_mask_slice(const float* hp, float* hs, int B, int T, int NPp, int NP, const int32_t* lens, void* stream);
int xva_cf_post_fwd(const float* x0, const float* y1, float* ld, float* out, float* ldsum, int B, int T, const int32_t* lens, void* stream);
int xva_cf_bwd_head(const float* d_out, const float* d_logdet, float* dm, float* d_ld, int B, int T, const int32_t* lens, void* stream);
int xva_cf_pad_mask(const float* dhs, float* dhp, int B, int T, int NPp, int NP, const int32_t* lens, void* stream);
int xva_small_wgrad(const float* dy, const float* x, float* dW, float* db, int64_t rows, int M, int N, void* stream);
int xva_cf_pre_bwd(const float* dh, const float* pre_w, const float* x0, const float* d_x0_pass, const float* d_x1, float* dz, float* d_pre_w, float* d_pre_b, int64_t rows,
                   int H, void* stream);

/* Rational-quadratic spline with linear tails, forward direction: piecewise_rational_quadratic_transform(inverse=False, tails="linear") of
 * python/xvapitch/util.py:203-391 as ConvFlow uses it (sdp.py:151-167).  x, y, logdet: n elements; h: (n, 3K - 1) raw parameters
 * [K widths | K heights | K - 1 derivatives], widths / heights multiplied by wh_scale (= 1 / sqrt(hidden)) before their softmax.  K <= 16.
 * The backward returns d x and d h from the gradients of y and of log|det|. */
int xva_rq_spline_fwd(const float* x, const float* h, float* y, float* logdet, int64_t n, int K, float wh_scale, float bound, void* stream);
int xva_rq_spline_bwd(const float* x, const float* h, const float* dy, const float* dlogdet, float* dx, float* dh, int64_t n, int K, float wh_scale,
                      float bound, void* stream);

/* The inverse direction of the same spline (util.py:322-350; ConvFlow with reverse=True, the duration predictor's sampling path sdp.py:311-321):
 * x = spline^-1(y) per element; outside [-bound, bound] the identity.  The sampling path discards log|det|, so none is returned. */
int xva_rq_spline_inv(const float* y, const float* h, float* x, int64_t n, int K, float wh_scale, float bound, void* stream);

/* ElementwiseAffine (sdp.py:95-114) on (B, T, C): y = (x * exp(log_scale) + translation) * mask, logdet[b] = len_b * sum(log_scale); the backward
 * accumulates into d_log_scale / d_translation.  xva_sdp_dequant_*: the variational-dequantisation step of StochasticDurationPredictor.forward
 * (sdp.py:283-296) per token: z0_log = log(max(dr - sigmoid(z_u), 1e-5)) * mask, logsig = (logsigmoid(z_u) + logsigmoid(-z_u)) * mask. */
int xva_affine_fwd(const float* x, const float* log_scale, const float* translation, float* y, float* logdet, const int32_t* lens, int B, int T, int C, void* stream);
int xva_affine_bwd(const float* x, const float* log_scale, const float* dy, const float* dlogdet, float* dx, float* d_log_scale, float* d_translation,
                   const int32_t* lens, int B, int T, int C, void* stream);
int xva_sdp_dequant_fwd(const float* z_u, const float* dr, float* z0_log, float* logsig, const int32_t* lens, int B, int T, void* stream);
int xva_sdp_dequant_bwd(const float* z_u, const float* dr, const float* d_z0_log, const float* d_logsig, float* d_z_u, const int32_t* lens, int B, int T,
                        void* stream);

#ifdef __cplusplus
}
#endif
#endif
