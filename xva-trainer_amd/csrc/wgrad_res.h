// wgrad_res.h — convolution weight gradients with RESIDENT operand tiles (gfx950).
//
//      dW[g][co][j * CI + ci] (+)= alpha * sum_{block b, row t} dY[b][t][g * CO + co] * X[b][s * t + j * d][g * CI + ci]
//
// (python/hifigan/models.py:17-260: the backward-weight product of every Conv1d of the generator's resblocks and of the grouped /
// strided scale-discriminator convolutions.)  As a GEMM this is a TN product with M = CO, N = k * CI, K = all rows: the output is tiny
// (CO x k*CI, e.g. 32 x 352) and K is 10^5 - 10^6.  The general TN kernel (gemm_glds.h) gives every 128 x 128 output tile its own
// workgroups, so X is streamed once per column tile — k times through L2 -> LDS — and hundreds of split-K slabs are needed to fill the
// chip (measured: 1.0 TB/s of algorithmic traffic, 100 - 450 TFLOP/s).  Here a workgroup owns the WHOLE (CO_W x column-range) output
// block in registers and walks down the rows in chunks of R: the chunk's R dY rows and its s*R + halo X rows travel HBM -> LDS once
// (global_load_lds), and every tap's MFMA operand is gathered from that one resident image by LDS transpose reads
// (ds_read_b64_tr_b16: both operands are row-major [t][channel], i.e. k-strided — a lane supplies the address of 4 channels of ONE row,
// and receives ONE channel of 4 rows).  A B fragment's 16 columns are four independent 4-channel units, each addressed by its own
// lanes: a column tile may span several taps (CI = 8: two taps per tile).
// Strided convolutions (s = 2, 3, 4) keep the X rows of a chunk de-interleaved by row phase (x mod s): a tap then walks consecutive
// rows of ONE phase region, exactly like a stride-1 tap, and the LDS bank pattern of a fragment read does not depend on s.
// Bank swizzle (16-byte chunk c of row idx stored at chunk c ^ swz(idx), applied to the DMA source addresses because the DMA image is
// lane-linear): a half-wave transpose read touches rows {r .. r+3, r+8 .. r+11} x 32 bytes; the XOR spreads those 8 pieces over the
// 8 32-byte windows of a bank row for 64 / 128 / 256-byte rows; 32-byte rows are permuted instead (bits 2 / 3 of the row swapped).
// Pipeline: a ring of D stages of 8 * NIW KiB; all 8 waves issue exactly NIW DMA instructions per chunk (the tail of a stage is padded
// with loads of a zero page), so "chunk c has landed" is the compile-time wait vmcnt((D - 2) * NIW) with chunks c + 1 .. c + D - 2 still in
// flight — inside a training step the operands come from HBM and one chunk of lookahead leaves the latency exposed (first version:
// 50 - 130 us where 13 - 40 were expected).  The 8 waves of a workgroup own DISTINCT column tiles (round-robin) and every wave walks all
// k-steps of a chunk: per barrier a wave has (R / 32) * MI * tiles MFMAs, and nothing is reduced across waves.  The DMA source offsets of a
// wave's NIW instructions are computed once (per-instruction row / column / region arithmetic costs ~100 VALU instructions: as much as
// the chunk's MFMAs when it was redone per chunk).
// Partial sums go to the caller's split-K slabs ([split][CO][k*CI] fp32); xva_gemm_splitk_reduce_kernel adds them up (deterministic).
#pragma once
#include "gemm_glds.h"

namespace xva_wgrad {
using namespace xva_glds;

struct Plan {
    int CO, CI, k, s, d;          // per-group channel counts, taps, stride, dilation (rows)
    int co_tiles;                 // CO / (MI * 16)
    int col_groups, tiles_per_cg; // column (tap x channel) groups of 16-column tiles; tiles per group (<= 8 * NBW)
    int R;                        // dY rows per chunk
    int RGN;                      // rows of one phase region of the X tile (multiple of 16)
    int cpi;                      // chunks per block
    int nsplit;                   // row-range splits (slabs)
    int a_bytes, b_bytes;         // LDS bytes of one dY / X tile (multiples of 1024; a_bytes + b_bytes <= 8 * NIW KiB)
    int64_t x_pitch;              // elements between consecutive X rows
    int ablate;                   // tools/wgrad_bench.py only: bit 0 = no fragment reads / MFMAs, bit 1 = no DMA inside the loop (results are garbage)
};

template <int CPR> __device__ __forceinline__ int swz(int idx) {
    if constexpr (CPR >= 16) return (((idx & 3) | (((idx >> 3) & 1) << 2)) << 1) & (CPR - 1);
    else if constexpr (CPR == 8) return ((((idx >> 1) & 1) | (((idx >> 3) & 1) << 1)) << 1);
    else if constexpr (CPR == 4) return ((idx >> 3) & 1) << 1;
    else return 0;
}
__device__ __forceinline__ int swz_rt(int cpr, int idx) {
    return cpr >= 16 ? ((((idx & 3) | (((idx >> 3) & 1) << 2)) << 1) & (cpr - 1)) : (cpr == 8 ? swz<8>(idx) : (cpr == 4 ? swz<4>(idx) : 0));
}
// position of row idx inside its region: 32-byte rows (2 chunks) are permuted so that rows r and r + 8 fall into different 32-byte windows mod 8
__device__ __forceinline__ int rowpos_rt(int cpr, int idx) { return cpr <= 2 ? swap23(idx) : idx; }

// LDS transpose reads as inline asm.  Through the builtin the compiler cannot tell the read from the LDS bytes an in-flight
// global_load_lds is still writing and puts `s_waitcnt vmcnt(0)` in front of every ds_read_b64_tr_b16 — the whole DMA ring drained once
// per k-step (first versions of this kernel: DMA time + compute time instead of their maximum).  The price: the compiler no longer
// tracks lgkmcnt for these reads; lds_wait<N>() is the explicit wait and ties the fragment registers to it.
__device__ __forceinline__ s16x4 ds_tr16(const XVA_LDS uint8_t* a) {
    s16x4 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"((uint32_t)(uintptr_t)a));
    return r;
}
struct TrFrag { s16x4 lo, hi; };
__device__ __forceinline__ TrFrag tr_issue(const XVA_LDS uint8_t* a0, const XVA_LDS uint8_t* a1) { TrFrag f; f.lo = ds_tr16(a0); f.hi = ds_tr16(a1); return f; }
// at most N LDS operations may still be outstanding (they return in order); the named fragments are complete afterwards
template <int N> __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ bf16x8 tr_value(TrFrag& f) {
    asm volatile("" : "+v"(f.lo), "+v"(f.hi));                     // ordered after the preceding lds_wait: nothing reads the registers before it
    s16x8 v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}

// MI: 16-row tiles of the output block (CO_W = 16 * MI); NBW: 16-column tiles per wave (the 8 waves share a column group round-robin);
// NIW: DMA instructions per wave per chunk (stage = 8 * NIW KiB); D: ring stages
template <int MI, int NBW, int NIW, int D>
__global__ __launch_bounds__(512, (8 * NIW * D <= 80 && MI * NBW <= 12 && NBW <= 6) ? 4 : 2) void xva_wgrad_res_kernel(xva_gemm_params p, Plan w) {
    extern __shared__ __attribute__((aligned(1024))) uint8_t smem_raw[];
    XVA_LDS uint8_t* smem = (XVA_LDS uint8_t*)smem_raw;
    constexpr int CO_W = MI * 16, CPA = CO_W / 8, STG = NIW * 8 * 1024;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int CPB = w.CI / 8;

    // ---- which block of the output, which row range -------------------------------------------------------------------------------
    int Lg;
    {
        const unsigned total = gridDim.x, id = blockIdx.x;
        const unsigned xcd = id & 7u, slot = id >> 3, q = total >> 3, r = total & 7u;
        Lg = (int)((xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot);
    }
    const int ntile = w.co_tiles * w.col_groups;
    const int tile = Lg % ntile, zz = Lg / ntile;
    const int ks = zz % w.nsplit, z2 = zz / w.nsplit;              // z2: group
    const int ct = tile % w.co_tiles, cg = tile / w.co_tiles;
    const int N = p.N;                                             // k * CI
    const int n_begin = cg * w.tiles_per_cg * 16;
    const int n_end = min(N, n_begin + w.tiles_per_cg * 16);
    const int j_lo = n_begin / w.CI;                               // first tap of this column group
    const int nblk = p.kb_len > 0 ? p.K / p.kb_len : 1;
    const int rows_blk = p.kb_len > 0 ? p.kb_len : p.K;
    const int chunks_total = nblk * w.cpi;
    const int per = (chunks_total + w.nsplit - 1) / w.nsplit;
    const int c_begin = ks * per, c_end = min(chunks_total, c_begin + per);

    const uint16_t* Ag = reinterpret_cast<const uint16_t*>(p.A) + (int64_t)z2 * p.sA2 + ct * CO_W;
    const uint16_t* Bg = reinterpret_cast<const uint16_t*>(p.B) + p.seg0 + (int64_t)z2 * p.sB2 + (int64_t)j_lo * w.d * w.x_pitch;
    const uint16_t* zero = reinterpret_cast<const uint16_t*>(g_zero_page);
    // rows of X (relative to Bg, inside a block) a valid dY row can touch: 0 .. s * (rows_blk - 1) + (taps of this group - 1) * d
    const int j_hi = (n_end - 1) / w.CI;
    const int x_rows_valid = w.s * (rows_blk - 1) + (j_hi - j_lo) * w.d + 1;
    const int nia = w.a_bytes >> 10, nib = w.b_bytes >> 10;        // DMA wave instructions per tile

    // Every wave issues exactly NIW instructions per chunk: instruction q = wave + 8 * i fills LDS bytes [q * 1024, q * 1024 + 1024) of the
    // stage.  What this lane fetches for instruction i (chunk-independent): dmo = element offset from the chunk's tile base, dmr = the
    // row it belongs to (dY row / X row of the tile; compared against the block's end), kind 0 = dY, 1 = X, 2 = padding (zero page).
    int dmo[NIW], dmr[NIW];
#pragma unroll
    for (int it = 0; it < NIW; ++it) {
        const int q = wave + 8 * it;
        dmo[it] = 0; dmr[it] = 0x7fffffff;
        if (q < nia) {
            const int pos = q * 64 + lane;
            const int rp = pos / CPA, cpos = pos - rp * CPA;
            const int idx = CPA <= 2 ? swap23(rp) : rp;
            dmo[it] = idx * (int)p.lda + (cpos ^ swz<CPA>(idx)) * 8;
            dmr[it] = idx;
        } else if (q < nia + nib) {
            const int pos = (q - nia) * 64 + lane;
            const int rowp = pos / CPB, cpos = pos - rowp * CPB;
            const int region = rowp / w.RGN, rp = rowp - region * w.RGN;
            const int idx = rowpos_rt(CPB, rp);                    // swap23 is an involution: position -> row
            const int xr = idx * w.s + region;                     // X row relative to the tile's first row
            if (region < w.s) { dmo[it] = xr * (int)w.x_pitch + (cpos ^ swz_rt(CPB, idx)) * 8; dmr[it] = xr; }
        }
    }
    auto issue = [&](int c, int slot) {
        XVA_LDS uint8_t* St = smem + slot * STG;
        const bool live = c < c_end;
        const int blk = live ? c / w.cpi : 0, tc = live ? (c - blk * w.cpi) * w.R : 0;
        const uint16_t* Ab = Ag + (int64_t)blk * p.kb_sA + (int64_t)tc * p.lda;
        const uint16_t* Bb = Bg + (int64_t)blk * p.kb_sB + (int64_t)w.s * tc * w.x_pitch;
        const int a_lim = live ? rows_blk - tc : 0, b_lim = live ? x_rows_valid - w.s * tc : 0;
#pragma unroll
        for (int it = 0; it < NIW; ++it) {
            const int q = wave + 8 * it;                           // q < nia is wave-uniform
            const uint16_t* src = q < nia ? (dmr[it] < a_lim ? Ab + dmo[it] : zero) : (dmr[it] < b_lim ? Bb + dmo[it] : zero);
            __builtin_amdgcn_global_load_lds((const XVA_GLB void*)src, (XVA_LDS void*)(St + q * 1024), 16, 0, 0);
        }
    };

    // ---- per-lane fragment offsets (k-step 0; a k-step adds 32 rows: the swizzles only use row bits below 5) --------------------------
    const int g = lane >> 4, i = lane & 15, q4 = i >> 2, u = i & 3;
    uint32_t oA[MI][2], oB[NBW][2];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = 8 * g + q4 + 4 * h;
            const int cch = mi * 2 + (u >> 1);
            const int rp = CPA <= 2 ? swap23(idx) : idx;
            oA[mi][h] = (uint32_t)((rp * CPA + (cch ^ swz<CPA>(idx))) * 16 + (u & 1) * 8);
        }
    int cnt = 0;                                                   // column tiles of this wave
    const int tiles_here = (n_end - n_begin + 15) / 16;
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int nt = nb * 8 + wave;
        if (nt < tiles_here) cnt = nb + 1;
        int n = n_begin + nt * 16 + u * 4;
        if (n > N - 4) n = N - 4;                                  // units past the last column read a valid place; never stored
        const int tap = n / w.CI - j_lo, ci = n - (tap + j_lo) * w.CI;
        const int xoff = tap * w.d;                                // X rows from the tile's first row at t = 0
        const int region = xoff % w.s, sh = xoff / w.s;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int idx = 8 * g + q4 + 4 * h + sh;
            const int rp = rowpos_rt(CPB, idx);
            oB[nb][h] = (uint32_t)(w.a_bytes + ((region * w.RGN + rp) * CPB + ((ci >> 3) ^ swz_rt(CPB, idx))) * 16 + ((ci >> 2) & 1) * 8);
        }
    }
    const uint32_t stepA = 32 * CPA * 16, stepB = 32 * CPB * 16;

    f32x4 acc[MI][NBW];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[mi][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // The bias gradient of the same layer (p.colsum_out): column sums of dY = dY^T x ones — one more MFMA per (k-step, 16-row tile) against a
    // fragment of ones, in wave 0 of the FIRST column group only (each (output-row tile, row range, group) has exactly one).
    const bool bias_wave = p.colsum_out != nullptr && cg == 0 && wave == 0;
    f32x4 accb[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) accb[mi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const bf16x8 ones = __builtin_bit_cast(bf16x8, (u32x4){0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u});

    const int ksteps = w.R / 32;
    constexpr int WAIT_N = (D - 2) * NIW;
    constexpr int WAIT_IMM = 0x0F70 | (WAIT_N & 15) | ((WAIT_N >> 4) << 14);
    static_assert(WAIT_N < 64, "vmcnt immediate");
#pragma unroll
    for (int j = 0; j < D - 1; ++j) issue(c_begin + j, j);
    int slot = 0;
    for (int c = c_begin; c < c_end; ++c) {
        __builtin_amdgcn_s_waitcnt(WAIT_IMM);                      // this wave's part of chunk c has landed (c + 1 .. c + D - 2 may be in flight)
        __builtin_amdgcn_s_barrier();                              // ... everyone's; and every wave is done with chunk c - 1's stage
        if (!(w.ablate & 2)) issue(c + D - 1, slot == 0 ? D - 1 : slot - 1);
        const XVA_LDS uint8_t* St = smem + slot * STG;
        // One kind of wait (lgkmcnt(0): scalar loads share the counter and return out of order, so counted waits are not safe): all fragment
        // reads of a k-step are issued, waited for, then its MFMAs run back to back; the CU's other waves fill the read latency.
        // (double-buffered fragments — reads of step q + 1 under the MFMAs of step q — spill: 2 x (MI + NBW) x 4 registers on top of the accumulators)
#pragma unroll 1
        for (int kq = 0; kq < ((w.ablate & 1) ? 0 : ksteps); ++kq) {
            TrFrag fa[MI], fb[NBW];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[mi] = tr_issue(St + oA[mi][0] + kq * stepA, St + oA[mi][1] + kq * stepA);
            static_for<NBW>([&](auto nbc) {
                constexpr int nb = decltype(nbc)::value;
                if (nb < cnt) fb[nb] = tr_issue(St + oB[nb][0] + kq * stepB, St + oB[nb][1] + kq * stepB);
            });
            lds_wait<0>();
            bf16x8 af[MI];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) af[mi] = tr_value(fa[mi]);
            if (bias_wave) {
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) accb[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, af[mi], accb[mi], 0, 0, 0);
            }
            static_for<NBW>([&](auto nbc) {
                constexpr int nb = decltype(nbc)::value;
                if (nb < cnt) {
                    const bf16x8 bfr = tr_value(fb[nb]);
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        acc[mi][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr, af[mi], acc[mi][nb], 0, 0, 0);   // lane: 4 consecutive columns of row (lane & 15)
                }
            });
        }
        slot = slot == D - 1 ? 0 : slot + 1;
    }
    // ---- partial sums -> slab [z2][ks][CO][N] ------------------------------------------------------------------------------------------
    float* slab = reinterpret_cast<float*>(p.sk_ws) + ((int64_t)z2 * w.nsplit + ks) * (int64_t)p.M * N;
    static_for<NBW>([&](auto nbc) {
        constexpr int nb = decltype(nbc)::value;
        const int n = n_begin + (nb * 8 + wave) * 16 + g * 4;
        if (nb < cnt && n < n_end) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const int co = ct * CO_W + mi * 16 + i;
                *reinterpret_cast<float4*>(slab + (int64_t)co * N + n) = make_float4(acc[mi][nb][0], acc[mi][nb][1], acc[mi][nb][2], acc[mi][nb][3]);
            }
        }
    });
    if (bias_wave && g == 0) {       // every column of the ones product holds the sum: lanes 0 .. 15 own rows (output channels) i
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) atomicAdd(p.colsum_out + (int64_t)z2 * p.M + ct * CO_W + mi * 16 + i, p.alpha * accb[mi][0]);
    }
}

}  // namespace xva_wgrad
