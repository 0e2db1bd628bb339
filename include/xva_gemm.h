/* xva_gemm.h — C-ABI descriptor of the MFMA "implicit-convolution GEMM" that carries every dense contraction of the
 * path: nn.Linear / nn.Conv1d(k=3) / torch.bmm of FastPitch (python/fastpitch1_1/fastpitch/transformer.py:59-152,
 * model.py:103-122,261), the DFT / mel-filterbank products of the mel front end (common/stft.py:86-103), and the
 * Conv1d / ConvTranspose1d / Conv2d(k,1) / grouped Conv1d stacks of HiFi-GAN (python/hifigan/models.py:17-260)
 * in forward, backward-data and backward-weight form.
 *
 * Operands live in HBM as fp32, bf16 or IEEE half (`a_dtype`, `b_dtype`, `c_dtype`; 0 = fp32, 1 = bf16, 2 = fp16), 16-byte aligned, leading
 * dimensions in ELEMENTS and multiples of 4 (fp32) / 8 (bf16).  `compute` selects the matrix pipe:
 *   0 = exact fp32 (v_mfma_f32_16x16x4_f32; operands must be stored fp32) — the parity mode;
 *   1 = bf16 inputs, fp32 accumulation (v_mfma_f32_16x16x32_bf16; fp32-stored operands are rounded while staged); fp16-stored operands: v_mfma_f32_16x16x32_f16;
 *   2 = fp32-stored operands, every element split into hi + lo bf16 while staged and three bf16 MFMAs per product (~1e-5 relative per product:
 *       what xva_gemm_set_fp32_products(1) makes of compute 0, chosen per call).
 *
 * Convolution over a time-major (rows = time, columns = channels) activation is expressed by K SEGMENTS: the reduction
 * index kk = j * seglen + c (tap j, channel c) reads the A row shifted by tap j:
 *      A(r, kk) = A[r * lda + kk + (kk / a_seglen) * a_segadj]                 (NT, NN)
 * so a conv with dilation d over C channels has a_seglen = C_in, a_segadj = d * row_stride - C_in and a strided conv
 * simply has lda = stride * row_stride.  (k = 3, d = 1, lda = C is the plain overlapping-row case: a_segadj = 0.)
 */
#ifndef XVA_GEMM_H
#define XVA_GEMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define XVA_GEMM_NT 0 /* C[M,N] = A[M,K] * B[N,K]^T   (A, B k-contiguous)                          */
#define XVA_GEMM_NN 1 /* C[M,N] = A[M,K] * B[K,N]     (B n-contiguous; row kk may be segment-mapped) */
#define XVA_GEMM_TN 2 /* C[M,N] = A[K,M]^T * B[K,N]   (A m-contiguous, B n-contiguous, col-segmented)*/

#define XVA_F32 0
#define XVA_BF16 1
#define XVA_F16 2  /* IEEE half (round 6): direct-to-LDS kernels only.  Wherever this header says "bf16 storage" a problem may use XVA_F16 instead — for ALL its
                    * 16-bit tensors at once (A, B and whichever of C / R / G / F are not fp32): compute 1 then issues v_mfma_f32_16x16x32_f16 (the same rate, 11
                    * mantissa bits instead of 8; the width the reference's own GPU path computes in under autocast, python/fastpitch1_1/xva_train.py:350,787).
                    * No split planes (a pair of halves is pointless), no resident-operand weight-gradient kernel. */

#define XVA_ACT_NONE 0
#define XVA_ACT_RELU 1
#define XVA_ACT_LRELU 2   /* slope = act_slope */
#define XVA_ACT_TANH 3
#define XVA_ACT_LOGCLAMP 4 /* log(max(v, act_slope)) */

typedef struct xva_gemm_params {
    const void* A;
    const void* B;
    void* C;
    int32_t M, N, K;
    int64_t lda, ldb, ldc;
    int32_t batch;          /* >= 1 (blockIdx batch); optional second level: z = z1 * batch2 + z2 */
    int64_t sA, sB, sC;     /* batch strides (elements) */
    int32_t batch2;         /* 0/1 = none */
    int64_t sA2, sB2, sC2;
    /* A tap segments (NT / NN): see header comment. a_seglen == 0 disables.
     * TN: A column m lives at A + k * lda + m + (m / a_seglen) * a_segadj (column segments, like B's). */
    int32_t a_seglen;
    int64_t a_segadj;
    /* B segments.  NN: row kk lives at B + seg0 + (kk / seglen) * segstride + (kk % seglen) * ldb.
     *              TN: column n lives at B + seg0 + k * ldb + n + (n / seglen) * segstride.
     * seglen == 0 disables. */
    int32_t seglen;
    int64_t seg0, segstride;
    /* operand transforms applied while staging (LeakyReLU fused into the consumer): x -> x > 0 ? x : slope * x */
    int32_t a_lrelu, b_lrelu;
    float a_slope, b_slope;
    /* epilogue: v = alpha * (acc + bias[col]) ; dropout ; [v += fm_c * sign(G - F)] ; v *= (G > 0 ? 1 : gate_slope) ; v += beta * R ; act ; row-mask ; store */
    float alpha, beta;
    const float* bias;      /* [N] fp32 or NULL; second-level batch z2 reads bias + z2 * sbias2 */
    int64_t sbias2;
    int32_t act;            /* XVA_ACT_* */
    float act_slope;
    const void* R;          /* residual (dtype r_dtype), or NULL */
    int64_t ldr, sR, sR2;
    int32_t r_dtype;
    const void* G;          /* gate tensor (dtype g_dtype): backward of ReLU / LeakyReLU, or NULL */
    int64_t ldg, sG, sG2;
    int32_t g_dtype;
    float gate_slope;
    /* row mask on the (mapped) global row index r' = r * mask_mul + mask_add (batch must be 1): item b = r' / Tp,
     * t' = r' % Tp; rows outside mask_pad <= t' < mask_pad + mask_len are structural zeros (mask_len == 0 means
     * Tp - 2 * mask_pad, mask_mul == 0 means 1); XVA_MASK_LEN additionally zeroes t' - mask_pad >= lens[b]. */
    int32_t mask_mode;
    const int32_t* lens;
    int32_t Tp, mask_pad, mask_len, mask_mul, mask_add;
    int32_t accumulate;     /* 0: C = v ; 1: C += v (atomic when splitk > 1) ; 2: C += v always atomically (batches that
                             * reduce into one C); fp32 C only for atomics */
    int32_t splitk;         /* >= 1; > 1 requires accumulate = 1 and a linear epilogue; 0 = let xva_gemm choose (it splits only
                             * when accumulate != 0, C is fp32 and the epilogue is linear) */
    int32_t compute;        /* 0 fp32, 1 bf16, 2 split-bf16 products of fp32 operands */
    int32_t layout;         /* XVA_GEMM_* */
    int32_t a_dtype, b_dtype, c_dtype;
    int32_t c_trans;        /* 1: store C transposed: element (row, col) at C[col * ldc + row] */
    /* TN only: the reduction index runs over `K / kb_len` blocks of kb_len rows (e.g. the items of a batch): row k of A / B
     * lives at base + (k / kb_len) * kb_sA|kb_sB + (k % kb_len) * lda|ldb.  kb_len == 0 disables. */
    int32_t kb_len;
    int64_t kb_sA, kb_sB;
    /* epilogue dropout (nn.Dropout on the GEMM output before the residual add): v *= 0 or 1/(1-p), mask = hash(seed, stream, row*N+col) */
    float drop_p;
    uint64_t drop_seed;
    uint32_t drop_stream;
    /* optional split-K scratch (device, 16-byte aligned): when splitk > 1 and sk_ws holds >= splitk * batch * batch2 * M * N floats
     * each K split writes its partial tile there and a second launch reduces them into C (no atomics; deterministic order).
     * Without it split-K accumulates with fp32 atomics. */
    void* sk_ws;
    int64_t sk_ws_bytes;
    /* optional second output with C's dtype and indexing (C2[i] next to every C[i] written): the LeakyReLU of the stored value,
     * lrelu(C[i], c2_slope) — the producer writes the activated copy its consumers would otherwise recompute per tap
     * (HiFi-GAN ResBlock1: the residual stream stays raw, the convolutions read the activated copy; models.py:41-48).
     * Needs splitk == 1 and accumulate == 0. */
    void* C2;
    float c2_slope;
    /* convolution hint (NT / NN with A tap segments): elements between consecutive INPUT rows of A (the tensor's channel count).  A
     * strided convolution has lda = stride * a_rowpitch, a grouped one a_seglen < a_rowpitch.  0 = lda.  Only used to recognise
     * problems the resident-input kernel can take; the product it describes is the same. */
    int64_t a_rowpitch;
    /* optional feature-matching term of a backward-data product (HiFi-GAN generator step: d/dx of 2 * mean|x_real - x_fake| through the fake
     * feature map, python/hifigan/models.py:263-269, added to the gradient that arrives from the layer above BEFORE the LeakyReLU gate):
     *     v = alpha * (acc + bias) ; dropout ; v += fm_c * sign(G - F) ; gate ; + beta * R ; act ; ...
     * F (dtype g_dtype) is indexed exactly like G (ldg, sG, sG2) and needs G; NULL disables.  Direct-to-LDS kernels (bf16 operands) only. */
    const void* F;
    float fm_c;
    /* SPLIT-bf16 PLANES (round 5; direct-to-LDS kernels, bf16 storage, no tap segments / K blocks, K a multiple of 32).  A value x that has to keep ~16
     * mantissa bits through the bf16 matrix pipe is stored as two bf16 tensors of the same shape, hi = bf16(x) and lo = bf16(x - hi):
     *   planes != 0: A and B are such pairs (hi plane at the given pointer, lo plane `a_plane` / `b_plane` ELEMENTS after it) and the product is
     *                A_hi B_hi + A_hi B_lo + A_lo B_hi with fp32 accumulation (lo x lo, <= 2^-16 of the term, is dropped) — ONE launch whose K loop runs
     *                three passes over the same tiles: the arithmetic of compute 2, with the split done once by the producer instead of at every staging;
     *   c_plane != 0 (c_dtype bf16, splitk == 1, no accumulate / second output / transposed store): C is written as such a pair (lo plane c_plane elements
     *                after C).  A gate tensor may be the hi plane of a pair (sign and zero-ness of x survive the rounding). */
    int32_t planes;
    int64_t a_plane, b_plane, c_plane;
    /* Convolution weight gradients on the resident-operand kernel (TN with column segments, bf16 operands, fp32 C accumulated through the slabs: csrc/wgrad_res.h):
     * colsum_out[z2 * M + m] += alpha * sum_k A[k][m] — the BIAS gradient of the same layer, from the dY tiles the kernel already holds in LDS (one extra MFMA
     * against a fragment of ones per k-step in one wave of the first column group; fp32 atomics across the row-range splits).  xva_gemm refuses the field when
     * another kernel would take the product: ask xva_gemm_takes_colsum first. */
    float* colsum_out;
} xva_gemm_params;

/* Launches on `stream` (a hipStream_t); returns 0 or a negative XVA_ERR_* code. */
int xva_gemm(const xva_gemm_params* p, void* stream);
/* 1 when xva_gemm would run this product on the resident-operand weight-gradient kernel, i.e. honours colsum_out (p->colsum_out itself is not looked at). */
int xva_gemm_takes_colsum(const xva_gemm_params* p);

/* Diagnostics / test knob: main-loop selection for bf16-stored operands. -1 automatic (default), 0 general register-staged kernel, 8 the
 * 256x128 tile with a 32-deep K tile (two workgroups per CU),
 * 1..5 direct-to-LDS 128x128 / 256x256 / 128x64 / 64x64 / 128x32 tiles wherever eligible, 6 automatic without the resident-input
 * convolution kernel, 7 the 384x128 tile (NT / NN). Returns the previous mode. Results are the same up to fp32 summation order. */
int xva_gemm_set_mainloop(int mode);
/* How products of fp32-stored operands with compute == 0 are formed: 0 (default) = the exact fp32 MFMA (k-ordered fmaf chain: the parity mode),
 * 1 = every operand element split into two bf16 while staged (x = hi + lo, 16 mantissa bits) and three bf16 MFMAs per product with fp32
 * accumulation (the lo * lo term, <= 2^-16 of the product, is dropped): ~1e-5 relative per product at a fraction of the matrix-pipe time.
 * Returns the previous mode. */
int xva_gemm_set_fp32_products(int mode);
int xva_gemm_get_fp32_products(void);
/* Diagnostics / test knob: K loop of the 256x256 direct-to-LDS tile. 0 = all waves in one phase (two barriers per 64-deep K tile),
 * 1 (default) = two wave groups one barrier apart over a ring of four 32-deep K tiles ({12 LDS reads + DMA | 32 MFMAs} phases), 2 = 1 for
 * the NT layout, 0 for NN / TN. Returns the previous mode. Same results up to fp32 summation order. */
int xva_gemm_set_kloop(int mode);
/* Diagnostics / test knob: K loop of the 384x128 direct-to-LDS tile (NT / NN). 1 (default) = the staggered loop above with {10 LDS reads + DMA |
 * 24 MFMAs} phases, 0 = all waves in one phase. Returns the previous mode. Same results up to fp32 summation order. */
int xva_gemm_set_kloop384(int mode);
/* NT products on the 256 x 256 tile (K % 64 == 0, tap segments a multiple of 64): 1 (default) = DMA pieces of 8 rows x 128 bytes (whole cache lines) into a ring of
 * five 32 KiB operand units, 0 = the 16-row x 64-byte pieces of the 32-deep tiles.  Same products in the same order: results are bit-identical. */
int xva_gemm_set_wholeline(int mode);
/* Diagnostics / test knob: 1 (default) = convolution weight gradients (TN, column segments, fp32 C accumulated through the caller's
 * split-K slabs) run on the resident-operand kernel (csrc/wgrad_res.h: the chunk's dY and X rows loaded once, all taps from LDS);
 * 0 = they stay on the general TN tiles.  Returns the previous mode.  Same results up to fp32 summation order. */
int xva_gemm_set_wgrad(int mode);
/* Tuning overrides of that kernel's plan (tools/wgrad_bench.py): rows per chunk (0 = automatic), DMA instructions per wave per chunk
 * (2 / 4 / 6: 16 / 32 / 48 KiB stages; 0 = automatic), workgroups in flight (0 = automatic).  xva_gemm_set_wgrad(2) additionally skips the
 * slab reduction (kernel timing only: C is NOT updated). */
void xva_gemm_wgrad_tune(int rows_per_chunk, int dma_per_wave, int workgroups);

#ifdef __cplusplus
}
#endif
#endif
