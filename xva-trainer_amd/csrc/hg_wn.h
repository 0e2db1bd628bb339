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

// Batched spectral norm (the 8 layers of MSD discriminator 0, one power iteration each per pass): every phase of the iteration is ONE
// launch over all layers instead of one per layer (7 tiny launches x 8 layers x 4 passes per iteration were launch-latency bound).
#define XVA_SN_BATCH 8
struct xva_sn_desc {
    const float* W; float* u; float* v;      // weight_orig (D0, inner), buffers weight_u (D0) / weight_v (inner): advanced in place
    float* su; float* sv;                    // copies of the advanced buffers kept for this pass's backward
    void* eff; void* eff2;                   // effective weight W / sigma, tap-major, dtype dt (and an optional fp32 copy)
    float* sigma; float* tmp;                // sigma (1 float) and inner + D0 floats of scratch
    int32_t dt, D0, D1, k;
    int32_t b_wtu, b_wv, b_scale;            // first block of this layer in the three multi-block phases
};
struct xva_sn_batch { int32_t n; int32_t nb_wtu, nb_wv, nb_scale; xva_sn_desc d[XVA_SN_BATCH]; };
extern "C" int xva_hg_spectral_norm_fwd_batch(xva_sn_desc* descs, int n, void* stream);
