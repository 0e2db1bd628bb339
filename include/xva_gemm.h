/* xva_gemm.h — C-ABI descriptor of the MFMA GEMM that carries every dense contraction of
 * the FastPitch hot path (reference: the cuBLAS/cuDNN calls behind nn.Linear / nn.Conv1d(k=3)
 * / torch.bmm in python/fastpitch1_1/fastpitch/transformer.py:59-152 and model.py:103-122,261).
 *
 * All operands are fp32 in HBM, 16-byte aligned, leading dimensions multiples of 4 elements.
 * `compute` selects the MFMA path: 0 = exact fp32 (v_mfma_f32_16x16x4_f32), 1 = bf16 inputs
 * with fp32 accumulation (v_mfma_f32_16x16x32_bf16; operands are rounded to bf16 while they
 * are staged into LDS).
 *
 * A k=3 "same" Conv1d over a (B, T+2, C) padded token-major tensor is expressed with an
 * OVERLAPPING-row A operand: lda = C, K = 3*C, A = x - C  (row r reads rows r-1, r, r+1).
 */
#ifndef XVA_GEMM_H
#define XVA_GEMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define XVA_GEMM_NT 0 /* C[M,N] = A[M,K] * B[N,K]^T   (A, B k-contiguous)            */
#define XVA_GEMM_NN 1 /* C[M,N] = A[M,K] * B[K,N]     (B n-contiguous, optional segs) */
#define XVA_GEMM_TN 2 /* C[M,N] = A[K,M]^T * B[K,N]   (A m-contiguous, B n-contiguous)*/

typedef struct xva_gemm_params {
    const float* A;
    const float* B;
    float* C;
    int32_t M, N, K;
    int64_t lda, ldb, ldc;
    int32_t batch;          /* >= 1 */
    int64_t sA, sB, sC;     /* batch strides (elements) */
    /* NN only: B row kk lives at B + seg0 + (kk / seglen) * segstride + (kk % seglen) * ldb.
     * seglen == 0 disables segmentation (row kk at B + kk * ldb). */
    int32_t seglen;
    int64_t seg0, segstride;
    /* epilogue: v = alpha*acc (+bias[col]) ; relu ; (+R) ; (*[G>0]) ; row-mask ; store */
    float alpha;
    const float* bias;      /* [N] or NULL */
    int32_t relu;
    float log_clamp;        /* > 0: v = logf(max(v, log_clamp)) after bias/relu (mel dynamic-range compression) */
    const float* R;         /* residual, same batch index, or NULL */
    int64_t ldr, sR;
    const float* G;         /* gate tensor (ReLU backward: keep where G > 0) or NULL */
    int64_t ldg, sG;
    int32_t mask_mode;      /* XVA_MASK_* applied on global row index (batch must be 1) */
    const int32_t* lens;    /* [rows / Tp] */
    int32_t Tp;
    int32_t accumulate;     /* 0: C = v ; 1: C += v (atomic when splitk > 1) */
    int32_t splitk;         /* >= 1; > 1 requires accumulate = 1 */
    int32_t compute;        /* 0 fp32, 1 bf16 */
    int32_t layout;         /* XVA_GEMM_* */
} xva_gemm_params;

/* Launches on `stream` (a hipStream_t); returns 0 or a negative XVA_ERR_* code. */
int xva_gemm(const xva_gemm_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif
