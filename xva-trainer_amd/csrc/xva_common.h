// xva_common.h — shared device/host helpers for the gfx950 (MI355X / CDNA4) kernels.
// Wave = 64 lanes everywhere in this tree; nothing here is written for 32-wide warps.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define XVA_OK 0
#define XVA_ERR_ARG -1
#define XVA_ERR_HIP -2
#define XVA_ERR_WS -3

// last-error string (thread-local), exported through xva_last_error()
void xva_set_error(const char* fmt, ...);

#define XVA_CHECK_ARG(cond, ...)                                                      \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            xva_set_error(__VA_ARGS__);                                                \
            return XVA_ERR_ARG;                                                        \
        }                                                                              \
    } while (0)

#define XVA_LAUNCH_CHECK()                                                            \
    do {                                                                               \
        hipError_t e__ = hipGetLastError();                                            \
        if (e__ != hipSuccess) {                                                       \
            xva_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,           \
                          hipGetErrorString(e__));                                     \
            return XVA_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define XVA_TRY(expr)                                                                 \
    do {                                                                               \
        int rc__ = (expr);                                                             \
        if (rc__ != XVA_OK) return rc__;                                               \
    } while (0)

static inline int xva_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Row-mask modes for sequence tensors stored in the padded token-major layout
// (B, Tp = T + 2, C): row 0 and row T+1 of every item are structural zero rows (the
// zero padding of the k=3 convolutions), rows 1..len are live, rows len+1..T are the
// batch padding the reference multiplies away with `mask`.
#define XVA_MASK_NONE 0
#define XVA_MASK_PAD 1  // zero only the two structural rows  (1 <= t' <= Tp-2 kept)
#define XVA_MASK_LEN 2  // zero everything outside 1 <= t' <= len[b]

#ifdef __HIPCC__
__device__ __forceinline__ bool xva_row_live(int mode, const int* __restrict__ lens, int Tp, long r) {
    if (mode == XVA_MASK_NONE) return true;
    int t, b;
    if (r >= 0 && r < (1l << 31)) {   // 32-bit division (a 64-bit one is ~100 VALU instructions per row: LayerNorm rows are issue-bound)
        b = (int)((unsigned)r / (unsigned)Tp); t = (int)((unsigned)r - (unsigned)b * (unsigned)Tp);
    } else { b = (int)(r / Tp); t = (int)(r - (long)b * Tp); }
    if (t == 0 || t == Tp - 1) return false;
    if (mode == XVA_MASK_PAD) return true;
    return t <= lens[b];
}

__device__ __forceinline__ float xva_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float xva_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blockDim.x <= 1024 (multiple of 64). `sh` needs 16 floats.
__device__ __forceinline__ float xva_block_sum(float v, float* sh) {
    v = xva_wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = 0.f;
    for (int i = 0; i < nw; ++i) r += sh[i];
    return r;
}
__device__ __forceinline__ float xva_block_max(float v, float* sh) {
    v = xva_wave_max(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    float r = -INFINITY;
    for (int i = 0; i < nw; ++i) r = fmaxf(r, sh[i]);
    return r;
}

// Stateless counter-based RNG for dropout: the keep-mask of element `idx` of stream `stream_id` under `seed` is a pure function,
// so backward regenerates it instead of storing it.  A keyed 32-bit mixer (two rounds of xorshift-multiply, the "lowbias32"
// constants) on the element counter: 2 integer multiplies per element.  (The first version ran a 64-bit murmur finaliser per
// element — 8 quarter-rate 32-bit multiplies — and made LayerNorm backward and the dropout epilogues VALU-bound.)  The two keys
// are functions of (seed, stream) only: uniform, computed on the scalar unit.
__device__ __forceinline__ uint32_t xva_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t xva_hash(uint64_t seed, uint32_t stream_id, uint64_t idx) {
    const uint32_t k1 = xva_mix32((uint32_t)seed ^ xva_mix32((uint32_t)(seed >> 32) + 0x9E3779B9u * (stream_id + 1u)));
    const uint32_t k2 = xva_mix32(k1 + 0x85ebca6bu);
    uint32_t x = ((uint32_t)idx + (uint32_t)(idx >> 32)) ^ k1;
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= k2;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// returns the multiplier to apply: 0 or 1/(1-p).  p<=0 -> 1.
__device__ __forceinline__ float xva_dropout_scale(float p, uint64_t seed, uint32_t stream_id, uint64_t idx) {
    if (p <= 0.f) return 1.f;
    uint32_t h = xva_hash(seed, stream_id, idx);
    float u = (float)(h >> 8) * (1.0f / 16777216.0f);
    return u < p ? 0.f : 1.f / (1.f - p);
}
#endif
