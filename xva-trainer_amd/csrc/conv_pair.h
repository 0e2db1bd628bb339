// conv_pair.h — one ResBlock1 pair of HiFi-GAN's generator (python/hifigan/models.py:41-48:  xt = c1(leaky_relu(x)); xt = c2(leaky_relu(xt)); x = xt + x)
// as ONE launch: the dilated convolution's activated output stays in LDS and feeds the second convolution from there (conv_pair.hip).
#pragma once
#include <stdint.h>
#include "../../include/xva_gemm.h"

// The SECOND convolution (k2 taps, dilation 1) is described by an xva_gemm_params in its per-item NT form (hg_conv.h: B = weights [C][k2 * C], bias, C / C2 / R,
// alpha / beta / accumulate, M = T rows per item, N = C, K = k2 * C, batch = items; A is not used).  The FIRST one by this struct.
struct xva_conv_pair {
    const void* X;        // bf16 input of the first convolution: row 0 of item 0 = valid row -(h1 + h2), h1 = d1 (k1 - 1) / 2, h2 = (k2 - 1) / 2 (pad rows: zeros)
    int64_t sX, ldx;      // elements between items / rows
    const void* W1;       // bf16 [C][k1 * C], tap-major
    const float* bias1;
    int k1, d1;
    float slope1;         // T1 = leaky_relu(conv1 + bias1, slope1)
    void* T1;             // bf16 output (valid row 0 of item 0): the activated intermediate, kept for the backward pass
    int64_t sT1, ldt;
    int x_raw;            // 0: X = the stored activated copy.  1: X = the RAW block input, LeakyReLU(x_slope) applied to the operand fragments, the residual of the
    float x_slope;        //    second convolution taken from the resident tile (its R must be null, alpha == beta).  2: raw input activated in place in LDS, R as given
};

// 0 = launched; -1 = this pair is not taken (the caller runs the two convolutions)
int xva_conv_pair_fwd(const xva_gemm_params* conv2, const xva_conv_pair* conv1, void* stream);
