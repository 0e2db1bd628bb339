// hg_wn.h — descriptors of the batched weight-norm launches (hg_ops.hip <-> hifigan_engine.hip; internal).
// HiFi-GAN re-parametrises ~130 conv weights per forward (torch.nn.utils.weight_norm, models.py:21-108): one launch per tensor is
// launch-latency bound (~8 us each), so the engine hands the kernel up to XVA_WN_BATCH layer descriptors as kernel arguments.
#pragma once
#include <stdint.h>
#define XVA_WN_BATCH 40
struct xva_wn_desc {
    const float* v; const float* g; float* norm;
    void* eff; void* effB;                 // forward outputs
    const float* dW; float* dv; float* dg;  // backward
    int32_t dt, kind, D0, D1, k, s, pconv, block0;
};
struct xva_wn_batch { int32_t n; xva_wn_desc d[XVA_WN_BATCH]; };
extern "C" int xva_hg_weight_norm_batch(const xva_wn_desc* descs, int n, int backward, void* stream);

// Batched loss reductions (hg_reduce): one launch for all feature maps of a discriminator pass.
#define XVA_RED_BATCH 48
struct xva_red_desc {
    const void* a; const void* b; float* out;
    float scale;
    int32_t dt, nseq, Hp, padF, T, C, mode, gx, vec, block0;
};
struct xva_red_batch { int32_t n; xva_red_desc d[XVA_RED_BATCH]; };
extern "C" int xva_hg_reduce_batch(const xva_red_desc* descs, int n, void* stream);

// Batched bias-gradient column sums (hg_colsum2): all layers of a discriminator backward in one launch.
#define XVA_CS_BATCH 64
struct xva_cs_desc { const void* X; float* out; int64_t rows; float scale; int32_t dt, C, Creal, rpb, cb, block0; };
struct xva_cs_batch { int32_t n; xva_cs_desc d[XVA_CS_BATCH]; };
extern "C" int xva_hg_colsum_batch(const xva_cs_desc* descs, int n, void* stream);
