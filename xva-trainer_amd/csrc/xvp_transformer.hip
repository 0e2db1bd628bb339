// xvp_transformer.hip — xVAPitch's RelativePositionTransformer (python/xvapitch/glow_tts.py:59-485 as TextEncoder / the pitch predictor build it:
// layer_norm_type "2", relative window, heads share the relative embeddings; model.py:1125-1136,1283-1305) as TWO engine calls, forward and backward.
//
// The kernels are the ones xva-trainer_amd/xvapitch/transformer.py sequenced from Python one ctypes call at a time (xva_gemm in implicit-conv form,
// xva_relattn_*, xva_ln_rows_*, xva_dropout_apply, xva_hg_colsum): the text encoder's 10 layers and the pitch predictor's 3 are ~600 launches of
// 5 - 40 us per iteration, and issuing them from Python cost ~17 us of host time each — more than the device needs to run them.  Here the host side is a
// C++ loop (one hipLaunch per kernel, nothing else), the per-layer weight re-layouts / masked copies / gradient scatters are one launch each instead of
// three to six torch operators, and everything the backward needs stays in ONE caller-owned workspace.
//
// Sequence layout (the same as xvapitch/wn.py:Seq): time-major (B, Tp = PAD + T + PAD, C) fp32 with GUARD spare rows before and after; pad and guard rows
// are structural zeros — the workspace must be ZERO-FILLED by the caller before xva_xvp_tr_forward (the convolution taps of an item's first / last token
// read them; several kernels only write token rows).
#include "xva_common.h"
#include "xva_gemm.h"
#include "xva_hip.h"

namespace {
constexpr int PAD = 8, GUARD = 32;

struct Geo {
    int B, T, Tp, C, F, H, L, k, w, Co, Cp, proj, cmp;
    float pd; uint64_t seed; uint32_t site0;
    int64_t rows, rtot;     // B * Tp ; 2 * GUARD + B * Tp
};

int geo_of(const xva_xvp_tr_dims* d, Geo& g) {
    XVA_CHECK_ARG(d, "xvp_tr: null dims");
    g.B = d->B; g.T = d->T; g.Tp = d->T + 2 * PAD; g.C = d->C; g.F = d->F; g.H = d->H; g.L = d->L; g.k = d->k; g.w = d->w; g.Co = d->Co;
    g.proj = d->has_proj; g.cmp = d->compute; g.pd = d->p_drop; g.seed = d->seed; g.site0 = d->site0;
    g.Cp = (g.Co + 3) / 4 * 4;
    XVA_CHECK_ARG(g.B > 0 && g.T > 0 && g.L > 0 && g.C > 0 && g.C % 4 == 0 && g.F % 4 == 0 && g.H > 0 && g.C % g.H == 0 && (g.k & 1) && g.k / 2 <= PAD,
                  "xvp_tr: B, T, L > 0; C, F multiples of 4; C %% H == 0; k odd <= %d", 2 * PAD + 1);
    XVA_CHECK_ARG(g.proj ? (g.Co == 1 || g.Co % 4 == 0) : g.Co == g.C, "xvp_tr: out_channels must equal C (no proj) or be 1 / a multiple of 4 (proj)");
    XVA_CHECK_ARG(g.pd >= 0.f && g.pd < 1.f && (g.cmp >= 0 && g.cmp <= 2), "xvp_tr: p_drop in [0, 1), compute 0 / 1 / 2");
    g.rows = (int64_t)g.B * g.Tp; g.rtot = g.rows + 2 * GUARD;
    return XVA_OK;
}

// ---- workspace carving: the same walk sizes it (base == nullptr) and hands out the pointers ------------------------------------------------------
struct Layer {
    float *xm, *qkv, *P, *att, *s1, *m1, *r1, *x1, *x1m, *h, *y2, *s2, *m2, *r2, *x2, *res;       // forward (kept for the backward)
    float *wqkv, *bqkv, *w1t, *w2t;                                                            // re-laid-out weights of this pass
    float *scr;                                                                                // backward: dW2 | dW1 | dWqkv | dbqkv (zero on entry)
    int CoL; bool ffn;
};
struct Ws {
    float *x0, *out, *wp, *bp;
    // backward sequences, one buffer per ROLE (a role's kernels write the same rows in every layer, so the structural zeros survive the reuse)
    float *dsq, *dx0, *ds2, *dy2, *dh, *dx1m, *dx1, *ds1, *dyo, *datt, *dqkv, *dS, *dxm[2], *wide, *dprj, *dWp, *dbp;
    Layer l[64];
    int64_t floats;
};
struct Carver {
    float* base; int64_t off;
    float* take(int64_t n) { n = (n + 63) / 64 * 64; float* p = base ? base + off : nullptr; off += n; return p; }
};
void carve(const Geo& g, float* base, Ws& w) {
    Carver c{base, 0};
    const int64_t R = g.rtot, C = g.C, F = g.F;
    const int64_t PP = (int64_t)g.B * g.H * g.T * g.T;
    w.x0 = c.take(R * C);
    for (int i = 0; i < g.L; i++) {
        Layer& l = w.l[i];
        const bool last = i == g.L - 1;
        l.CoL = last ? g.Co : g.C;
        l.ffn = !(last && g.Co == 1);
        l.xm = c.take(R * C); l.qkv = c.take(R * 3 * C); l.P = c.take(PP); l.att = c.take(R * C); l.s1 = c.take(R * C);
        l.m1 = c.take(g.rows); l.r1 = c.take(g.rows); l.x1 = c.take(R * C);
        l.res = (last && g.proj) ? c.take(R * g.Cp) : nullptr;
        l.wqkv = c.take(3 * C * C); l.bqkv = c.take(3 * C);
        if (l.ffn) {
            l.x1m = c.take(R * C); l.h = c.take(R * F); l.y2 = c.take(R * l.CoL); l.s2 = c.take(R * l.CoL); l.m2 = c.take(g.rows); l.r2 = c.take(g.rows);
            l.x2 = c.take(R * l.CoL); l.w1t = c.take(F * g.k * C); l.w2t = c.take((int64_t)l.CoL * g.k * F);
            l.scr = c.take((int64_t)l.CoL * g.k * F + F * g.k * C + 3 * C * C + 3 * C);
        } else {
            l.x1m = l.h = l.y2 = l.s2 = l.m2 = l.r2 = l.x2 = l.w1t = l.w2t = nullptr;
            l.scr = c.take(3 * C * C + 3 * C);
        }
    }
    const int64_t Cx = g.Cp > g.C ? g.Cp : g.C;
    w.out = c.take(R * g.Co);
    w.wp = g.proj ? c.take((int64_t)g.Cp * C) : nullptr; w.bp = g.proj ? c.take(g.Cp) : nullptr;
    w.dsq = c.take(R * g.Co); w.dx0 = c.take(R * g.Co);
    w.ds2 = c.take(R * Cx); w.dy2 = c.take(R * Cx); w.dh = c.take(R * F); w.dx1m = c.take(R * C); w.dx1 = c.take(R * C); w.ds1 = c.take(R * C);
    w.dyo = c.take(R * C); w.datt = c.take(R * C); w.dqkv = c.take(R * 3 * C); w.dS = c.take(PP); w.dxm[0] = c.take(R * C); w.dxm[1] = c.take(R * C);
    w.wide = g.proj ? c.take(R * g.Cp) : nullptr; w.dprj = g.proj ? c.take(R * C) : nullptr;
    w.dWp = g.proj ? c.take((int64_t)g.Cp * C) : nullptr; w.dbp = g.proj ? c.take(g.Cp) : nullptr;
    w.floats = c.off;
}
inline float* view(float* store, int64_t C) { return store + (int64_t)GUARD * C; }

// ---- small kernels ---------------------------------------------------------------------------------------------------------------------------------
// dst = src * x_mask over the B * Tp view rows (float4; C % 4 == 0): the copy + xva_seq_mask pair of the Python sequencing in one pass
__global__ void copy_mask_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4, int C4, int Tp, const int32_t* __restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int64_t r = i / C4;
    const int b = (int)(r / Tp), t = (int)(r - (int64_t)b * Tp) - PAD;
    dst[i] = (t >= 0 && t < lens[b]) ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
}
__global__ void add_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, int64_t n4) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 x = a[i], y = b[i];
    o[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
}
// dxm = (dxm + ds1) * x_mask
__global__ void add_mask_kernel(float4* __restrict__ x, const float4* __restrict__ y, int64_t n4, int C4, int Tp, const int32_t* __restrict__ lens) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int64_t r = i / C4;
    const int b = (int)(r / Tp), t = (int)(r - (int64_t)b * Tp) - PAD;
    float4 a = x[i];
    const float4 c = y[i];
    a = (t >= 0 && t < lens[b]) ? make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    x[i] = a;
}
__global__ void axpy_kernel(float* __restrict__ y, const float* __restrict__ x, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += x[i];
}
// columns: dst (rows, Cd) [:, :n] = src (rows, Cs) [:, :n]
__global__ void cols_kernel(const float* __restrict__ src, int Cs, float* __restrict__ dst, int Cd, int n, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * n) return;
    const int64_t r = i / n; const int c = (int)(i - r * n);
    dst[r * Cd + c] = src[r * Cs + c];
}
// the pass's weight layouts of one layer: wqkv (3C, C) / bqkv (3C) = the stacked q, k, v projections; w1t (F, k C), w2t (Co, k F) = the feed-forward
// convolutions tap-major (nn.Conv1d (Cout, Cin, k) -> (Cout, k * Cin))
struct PackArgs { const float *wq, *wk, *wv, *bq, *bk, *bv, *w1, *w2; float *wqkv, *bqkv, *w1t, *w2t; int C, F, k, Co; };
__global__ void pack_kernel(PackArgs a) {
    const int64_t CC = (int64_t)a.C * a.C, n_qkv = 3 * CC, n_b = 3 * a.C, n1 = a.w1 ? (int64_t)a.F * a.k * a.C : 0, n2 = a.w2 ? (int64_t)a.Co * a.k * a.F : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_qkv + n_b + n1 + n2; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_qkv) {
            const int j = (int)(i / CC);
            a.wqkv[i] = (j == 0 ? a.wq : j == 1 ? a.wk : a.wv)[i - j * CC];
        } else if (i < n_qkv + n_b) {
            const int64_t e = i - n_qkv; const int j = (int)(e / a.C);
            a.bqkv[e] = (j == 0 ? a.bq : j == 1 ? a.bk : a.bv)[e - (int64_t)j * a.C];
        } else if (i < n_qkv + n_b + n1) {
            const int64_t e = i - n_qkv - n_b;                       // e = (f, j, c) of w1t ; source w1[f][c][j]
            const int c = (int)(e % a.C); const int64_t fj = e / a.C; const int j = (int)(fj % a.k); const int64_t f = fj / a.k;
            a.w1t[e] = a.w1[(f * a.C + c) * a.k + j];
        } else {
            const int64_t e = i - n_qkv - n_b - n1;                  // e = (co, j, f) of w2t ; source w2[co][f][j]
            const int f = (int)(e % a.F); const int64_t cj = e / a.F; const int j = (int)(cj % a.k); const int64_t co = cj / a.k;
            a.w2t[e] = a.w2[(co * a.F + f) * a.k + j];
        }
    }
}
// the reverse walk for the gradients: parameter gradients += the layer's scratch (dW2 | dW1 tap-major, dWqkv | dbqkv stacked)
struct UnpackArgs { float *gq, *gk, *gv, *gbq, *gbk, *gbv, *g1, *g2; const float *dqkv, *dbqkv, *d1, *d2; int C, F, k, Co; };
__global__ void unpack_kernel(UnpackArgs a) {
    const int64_t CC = (int64_t)a.C * a.C, n_qkv = 3 * CC, n_b = 3 * a.C, n1 = a.g1 ? (int64_t)a.F * a.k * a.C : 0, n2 = a.g2 ? (int64_t)a.Co * a.k * a.F : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_qkv + n_b + n1 + n2; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_qkv) {
            const int j = (int)(i / CC);
            (j == 0 ? a.gq : j == 1 ? a.gk : a.gv)[i - j * CC] += a.dqkv[i];
        } else if (i < n_qkv + n_b) {
            const int64_t e = i - n_qkv; const int j = (int)(e / a.C);
            (j == 0 ? a.gbq : j == 1 ? a.gbk : a.gbv)[e - (int64_t)j * a.C] += a.dbqkv[e];
        } else if (i < n_qkv + n_b + n1) {
            const int64_t e = i - n_qkv - n_b;                       // destination-major: e = (f, c, j) of g1 ; source d1[f][j * C + c]
            const int j = (int)(e % a.k); const int64_t fc = e / a.k; const int c = (int)(fc % a.C); const int64_t f = fc / a.C;
            a.g1[e] += a.d1[(f * a.k + j) * a.C + c];
        } else {
            const int64_t e = i - n_qkv - n_b - n1;                  // e = (co, f, j) of g2 ; source d2[co][j * F + f]
            const int j = (int)(e % a.k); const int64_t cf = e / a.k; const int f = (int)(cf % a.F); const int64_t co = cf / a.F;
            a.g2[e] += a.d2[(co * a.k + j) * a.F + f];
        }
    }
}

inline unsigned grid_of(int64_t n, int cap = 4096) { int64_t g = (n + 255) / 256; return (unsigned)(g > cap ? cap : (g < 1 ? 1 : g)); }

// ---- the products, in the parameterisation of xvapitch/wn.py conv_fwd / conv_bwd_data / conv_bwd_weight ------------------------------------------
struct Drop { float p; uint64_t seed; uint32_t site; };
xva_gemm_params base_params(const Geo& g) {
    xva_gemm_params p;
    memset(&p, 0, sizeof(p));
    p.batch = 1; p.batch2 = 1; p.alpha = 1.f; p.beta = 1.f; p.splitk = 1; p.mask_mul = 1; p.compute = g.cmp;
    p.mask_mode = XVA_MASK_PAD; p.Tp = g.Tp; p.mask_pad = PAD; p.mask_len = g.T;
    return p;
}
// y = conv1d(x; k taps) + bias [+ relu] [dropout] [+ R] on every view row; pad rows zeroed by the epilogue mask
int conv_fwd(const Geo& g, const float* x, int Cin, const float* w, const float* bias, float* y, int Cout, int k, bool relu, const float* R, const Drop* dr, void* st) {
    xva_gemm_params p = base_params(g);
    const int P = (k - 1) / 2;
    p.A = x + (int64_t)(GUARD - P) * Cin; p.B = w; p.C = y + (int64_t)GUARD * Cout;
    p.M = (int32_t)g.rows; p.N = Cout; p.K = k * Cin; p.lda = Cin; p.ldb = (int64_t)k * Cin; p.ldc = Cout;
    p.layout = XVA_GEMM_NT; p.bias = bias; p.act = relu ? XVA_ACT_RELU : XVA_ACT_NONE;
    p.a_seglen = k > 1 ? Cin : 0; p.a_segadj = 0;
    if (R) { p.R = R; p.ldr = Cout; }
    if (dr && dr->p > 0.f) { p.drop_p = dr->p; p.drop_seed = dr->seed; p.drop_stream = dr->site; }
    return xva_gemm(&p, st);
}
// dx = dy (*) w^T (tap-major w (Cout, k Cin)) [gated by G > 0] [dropout]
int conv_bwd_data(const Geo& g, const float* dy, int Cout, const float* w, float* dx, int Cin, int k, const float* G, const Drop* dr, void* st) {
    xva_gemm_params p = base_params(g);
    const int P = (k - 1) / 2;
    p.A = dy + (int64_t)(GUARD + P) * Cout; p.B = w; p.C = dx + (int64_t)GUARD * Cin;
    p.M = (int32_t)g.rows; p.N = Cin; p.K = k * Cout; p.lda = Cout; p.ldb = (int64_t)k * Cin; p.ldc = Cin;
    p.layout = XVA_GEMM_NN;
    if (k > 1) { p.a_seglen = Cout; p.a_segadj = -2 * (int64_t)Cout; p.seglen = Cout; p.seg0 = 0; p.segstride = Cin; }
    if (G) { p.G = G; p.ldg = Cin; p.gate_slope = 0.f; }
    if (dr && dr->p > 0.f) { p.drop_p = dr->p; p.drop_seed = dr->seed; p.drop_stream = dr->site; }
    return xva_gemm(&p, st);
}
// dw (Cout, k Cin) += dy^T xcat ; db (Cout) += column sums of dy
int conv_bwd_weight(const Geo& g, const float* dy, int Cout, const float* x, int Cin, float* dw, float* db, int k, void* sk, int64_t sk_bytes, void* st) {
    xva_gemm_params p = base_params(g);
    const int P = (k - 1) / 2;
    p.mask_mode = XVA_MASK_NONE; p.Tp = 0; p.mask_pad = 1; p.mask_len = 0;
    p.A = dy + (int64_t)GUARD * Cout; p.B = x + (int64_t)(GUARD - P) * Cin; p.C = dw;
    p.M = Cout; p.N = k * Cin; p.K = (int32_t)g.rows; p.lda = Cout; p.ldb = Cin; p.ldc = (int64_t)k * Cin;
    p.layout = XVA_GEMM_TN; p.accumulate = 1; p.splitk = 0;
    if (k > 1) { p.seglen = Cin; p.seg0 = 0; p.segstride = 0; }
    p.sk_ws = sk; p.sk_ws_bytes = sk_bytes;
    XVA_TRY(xva_gemm(&p, st));
    return xva_hg_colsum(view((float*)dy, Cout), 0, db, g.rows, Cout, 1.f, st);
}
int copy_mask(const Geo& g, const float* src, float* dst, int C, const int32_t* lens, hipStream_t s) {
    const int64_t n4 = g.rows * C / 4;
    hipLaunchKernelGGL(copy_mask_kernel, dim3((unsigned)xva_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)view((float*)src, C), (float4*)view(dst, C), n4, C / 4, g.Tp, lens);
    XVA_LAUNCH_CHECK();
    return XVA_OK;
}
enum { P_WQ = 0, P_BQ, P_WK, P_BK, P_WV, P_BV, P_WO, P_BO, P_EK, P_EV, P_W1, P_B1, P_W2, P_B2, P_G1, P_BE1, P_G2, P_BE2 };
}  // namespace

extern "C" int64_t xva_xvp_tr_workspace_bytes(const xva_xvp_tr_dims* d) {
    Geo g;
    if (geo_of(d, g) != XVA_OK || g.L > 64) return -1;
    Ws w;
    carve(g, nullptr, w);
    return w.floats * 4;
}

extern "C" int xva_xvp_tr_forward(const xva_xvp_tr_dims* d, const float* const* prm, const float* x_bct, const int32_t* lens, float* out_bct, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    Geo g;
    XVA_TRY(geo_of(d, g));
    XVA_CHECK_ARG(g.L <= 64 && prm && x_bct && lens && out_bct && workspace, "xvp_tr_forward: null argument / more than 64 layers");
    Ws w;
    carve(g, (float*)workspace, w);
    XVA_CHECK_ARG(workspace_bytes >= w.floats * 4 && ((uintptr_t)workspace % 16) == 0, "xvp_tr_forward: workspace too small (%lld < %lld) or unaligned", (long long)workspace_bytes,
                  (long long)(w.floats * 4));
    hipStream_t s = (hipStream_t)stream;
    const int C = g.C, F = g.F, k = g.k, dk = C / g.H;
    XVA_TRY(xva_bct_to_seq(x_bct, view(w.x0, C), 0, g.B, C, g.T, PAD, nullptr, stream));
    if (g.proj) {                                                                                 // proj rows zero-padded to a multiple of 4 (out_channels == 1)
        const float* const* pp = prm + (int64_t)g.L * XVA_XVP_TR_PER_LAYER;
        if (hipMemcpyAsync(w.wp, pp[0], (size_t)g.Co * C * 4, hipMemcpyDeviceToDevice, s) != hipSuccess ||
            hipMemcpyAsync(w.bp, pp[1], (size_t)g.Co * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) { xva_set_error("xvp_tr_forward: proj copy failed"); return XVA_ERR_HIP; }
    }
    const float* x = w.x0;
    int Cx = C;
    for (int li = 0; li < g.L; li++) {
        Layer& l = w.l[li];
        const float* const* p = prm + (int64_t)li * XVA_XVP_TR_PER_LAYER;
        const bool last = li == g.L - 1;
        const uint32_t site = g.site0 + 4u * li;
        PackArgs pa{p[P_WQ], p[P_WK], p[P_WV], p[P_BQ], p[P_BK], p[P_BV], l.ffn ? p[P_W1] : nullptr, l.ffn ? p[P_W2] : nullptr, l.wqkv, l.bqkv, l.w1t, l.w2t, C, F, k, l.CoL};
        hipLaunchKernelGGL(pack_kernel, dim3(grid_of(3ll * C * C + (l.ffn ? (int64_t)F * k * C * 2 : 0), 1024)), dim3(256), 0, s, pa);
        XVA_LAUNCH_CHECK();
        XVA_TRY(copy_mask(g, x, l.xm, C, lens, s));                                               // x = x * x_mask            (glow_tts.py:471)
        XVA_TRY(conv_fwd(g, l.xm, C, l.wqkv, l.bqkv, l.qkv, 3 * C, 1, false, nullptr, nullptr, stream));   // conv_q / conv_k / conv_v  (:166-168)
        float* qv = view(l.qkv, 3 * C);
        XVA_TRY(xva_relattn_fwd(qv, qv + C, qv + 2 * C, 3 * C, p[P_EK], p[P_EV], lens, l.P, view(l.att, C), C, g.B, g.T, g.H, dk, g.w, 1, g.Tp, PAD, g.pd, g.seed,
                                site, stream));                                                   // attention, dropout(p_attn) (:173-214)
        Drop d1{g.pd, g.seed, site + 1};
        XVA_TRY(conv_fwd(g, l.att, C, p[P_WO], p[P_BO], l.s1, C, 1, false, view(l.xm, C), &d1, stream));   // x + dropout(conv_o(..))   (:170,473-474)
        XVA_TRY(xva_ln_rows_fwd(view(l.s1, C), p[P_G1], p[P_BE1], view(l.x1, C), l.m1, l.r1, g.rows, C, 1e-5f, stream));   // norm_layers_1 (:474)
        const float* res = l.x1;
        if (last && g.proj) {                                                                      // x = proj(x)               (:479-480)
            XVA_TRY(conv_fwd(g, l.x1, C, w.wp, w.bp, l.res, g.Cp, 1, false, nullptr, nullptr, stream));
            res = l.res;
        }
        if (!l.ffn) { x = res; Cx = g.Cp; continue; }                                              // out_channels == 1: the stack returns proj(x) (:482)
        XVA_TRY(copy_mask(g, l.x1, l.x1m, C, lens, s));                                            // FFN: conv_1(pad(x * x_mask)), relu (:342-343)
        Drop d2{g.pd, g.seed, site + 2};
        XVA_TRY(conv_fwd(g, l.x1m, C, l.w1t, p[P_B1], l.h, F, k, true, nullptr, &d2, stream));      // dropout(relu(.)) = relu(dropout(.)) (:343-344)
        XVA_TRY(xva_seq_mask(view(l.h, F), 0, g.B, g.Tp, PAD, F, lens, stream));                   // conv_2(pad(x * x_mask)) * x_mask (:345-346)
        XVA_TRY(conv_fwd(g, l.h, F, l.w2t, p[P_B2], l.y2, l.CoL, k, false, nullptr, nullptr, stream));
        XVA_TRY(xva_seq_mask(view(l.y2, l.CoL), 0, g.B, g.Tp, PAD, l.CoL, lens, stream));
        if (g.pd > 0.f)                                                                            // y = dropout(ffn(x))       (:477)
            XVA_TRY(xva_dropout_apply(view(l.y2, l.CoL), view(l.y2, l.CoL), 0, g.rows * l.CoL, g.pd, g.seed, site + 3, stream));
        const int64_t n4 = g.rtot * l.CoL / 4;
        hipLaunchKernelGGL(add_kernel, dim3((unsigned)xva_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)res, (const float4*)l.y2, (float4*)l.s2, n4);   // norm_layers_2(x + y) (:482)
        XVA_LAUNCH_CHECK();
        XVA_TRY(xva_ln_rows_fwd(view(l.s2, l.CoL), p[P_G2], p[P_BE2], view(l.x2, l.CoL), l.m2, l.r2, g.rows, l.CoL, 1e-5f, stream));
        x = l.x2; Cx = l.CoL;
    }
    if (Cx == g.Co) {
        XVA_TRY(copy_mask(g, x, w.out, g.Co, lens, s));                                            // x * x_mask                (:483)
    } else {
        hipLaunchKernelGGL(cols_kernel, dim3((unsigned)xva_cdiv(g.rows * g.Co, 256)), dim3(256), 0, s, view((float*)x, Cx), Cx, view(w.out, g.Co), g.Co, g.Co, g.rows);
        XVA_LAUNCH_CHECK();
        XVA_TRY(xva_seq_mask(view(w.out, g.Co), 0, g.B, g.Tp, PAD, g.Co, lens, stream));
    }
    return xva_seq_to_bct(view(w.out, g.Co), out_bct, 0, g.B, g.Co, g.T, PAD, 0, stream);
}

extern "C" int xva_xvp_tr_backward(const xva_xvp_tr_dims* d, const float* const* prm, float* const* grd, const float* d_out_bct, const int32_t* lens, float* d_x_bct,
                                   void* workspace, int64_t workspace_bytes, void* sk_ws, int64_t sk_ws_bytes, void* stream) {
    Geo g;
    XVA_TRY(geo_of(d, g));
    XVA_CHECK_ARG(g.L <= 64 && prm && grd && d_out_bct && lens && d_x_bct && workspace && sk_ws, "xvp_tr_backward: null argument / more than 64 layers");
    Ws w;
    carve(g, (float*)workspace, w);
    XVA_CHECK_ARG(workspace_bytes >= w.floats * 4, "xvp_tr_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int C = g.C, F = g.F, k = g.k, dk = C / g.H;
    XVA_TRY(xva_bct_to_seq(d_out_bct, view(w.dsq, g.Co), 0, g.B, g.Co, g.T, PAD, nullptr, stream));
    {   // dx = d_out * x_mask (Co need not be a multiple of 4: scalar path through xva_seq_mask)
        if (g.Co % 4 == 0) XVA_TRY(copy_mask(g, w.dsq, w.dx0, g.Co, lens, s));
        else {
            if (hipMemcpyAsync(view(w.dx0, g.Co), view(w.dsq, g.Co), (size_t)g.rows * g.Co * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) { xva_set_error("xvp_tr_backward: copy failed"); return XVA_ERR_HIP; }
            XVA_TRY(xva_seq_mask(view(w.dx0, g.Co), 0, g.B, g.Tp, PAD, g.Co, lens, stream));
        }
    }
    // d proj(x1): accumulates the proj gradients, returns d x1 in w.dprj
    auto proj_bwd = [&](const float* dres /* (rtot, Co) */, const float* x1, float* const* pg) -> int {
        const float* dr = dres;
        if (g.Cp != g.Co) {                                                                        // one output channel rides in a 4-wide sequence (GEMM leading dimensions)
            hipLaunchKernelGGL(cols_kernel, dim3((unsigned)xva_cdiv(g.rtot * g.Co, 256)), dim3(256), 0, s, dres, g.Co, w.wide, g.Cp, g.Co, g.rtot);
            XVA_LAUNCH_CHECK();
            dr = w.wide;
        }
        XVA_TRY(conv_bwd_weight(g, dr, g.Cp, x1, C, w.dWp, w.dbp, 1, sk_ws, sk_ws_bytes, stream));
        hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)xva_cdiv((int64_t)g.Co * C, 256)), dim3(256), 0, s, pg[0], w.dWp, (int64_t)g.Co * C);
        hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)xva_cdiv(g.Co, 256)), dim3(256), 0, s, pg[1], w.dbp, (int64_t)g.Co);
        XVA_LAUNCH_CHECK();
        return conv_bwd_data(g, dr, g.Cp, w.wp, w.dprj, C, 1, nullptr, nullptr, stream);
    };
    const float* dx = w.dx0;                                                                       // (rtot, Co) into the last layer, (rtot, C) below it
    float* const* pgr = grd + (int64_t)g.L * XVA_XVP_TR_PER_LAYER;
    for (int li = g.L - 1; li >= 0; li--) {
        Layer& l = w.l[li];
        const float* const* p = prm + (int64_t)li * XVA_XVP_TR_PER_LAYER;
        float* const* gr = grd + (int64_t)li * XVA_XVP_TR_PER_LAYER;
        const bool last = li == g.L - 1;
        const uint32_t site = g.site0 + 4u * li;
        const int Co = l.CoL;
        const float* dx1;
        float* dWqkv; float* dbqkv;
        if (!l.ffn) {
            XVA_TRY(proj_bwd(dx, l.x1, pgr));
            dx1 = w.dprj;
            dWqkv = l.scr; dbqkv = l.scr + 3ll * C * C;
        } else {
            float* dW2 = l.scr; float* dW1 = dW2 + (int64_t)Co * k * F;
            dWqkv = dW1 + (int64_t)F * k * C; dbqkv = dWqkv + 3ll * C * C;
            XVA_TRY(xva_ln_rows_bwd(view((float*)dx, Co), view(l.s2, Co), l.m2, l.r2, p[P_G2], view(w.ds2, Co), gr[P_G2], gr[P_BE2], g.rows, Co, stream));
            XVA_TRY(copy_mask(g, w.ds2, w.dy2, Co, lens, s));                                      // y2 = dropout(conv_2(..) * x_mask)
            if (g.pd > 0.f) XVA_TRY(xva_dropout_apply(view(w.dy2, Co), view(w.dy2, Co), 0, g.rows * Co, g.pd, g.seed, site + 3, stream));
            XVA_TRY(conv_bwd_weight(g, w.dy2, Co, l.h, F, dW2, gr[P_B2], k, sk_ws, sk_ws_bytes, stream));
            // d(conv_1 output) = (dy2 (*) W2) gated by relu (h is stored masked and post-ReLU: h > 0 is both the gate and the mask), then the site's dropout
            Drop d2{g.pd, g.seed, site + 2};
            XVA_TRY(conv_bwd_data(g, w.dy2, Co, l.w2t, w.dh, F, k, view(l.h, F), &d2, stream));
            XVA_TRY(conv_bwd_weight(g, w.dh, F, l.x1m, C, dW1, gr[P_B1], k, sk_ws, sk_ws_bytes, stream));
            XVA_TRY(conv_bwd_data(g, w.dh, F, l.w1t, w.dx1m, C, k, nullptr, nullptr, stream));
            XVA_TRY(xva_seq_mask(view(w.dx1m, C), 0, g.B, g.Tp, PAD, C, lens, stream));            // x1m = x1 * x_mask
            const float* dres = w.ds2;                                                             // residual branch: x or proj(x)
            if (last && g.proj) { XVA_TRY(proj_bwd(w.ds2, l.x1, pgr)); dres = w.dprj; }
            const int64_t n4 = g.rtot * C / 4;
            hipLaunchKernelGGL(add_kernel, dim3((unsigned)xva_cdiv(n4, 256)), dim3(256), 0, s, (const float4*)dres, (const float4*)w.dx1m, (float4*)w.dx1, n4);
            XVA_LAUNCH_CHECK();
            dx1 = w.dx1;
        }
        XVA_TRY(xva_ln_rows_bwd(view((float*)dx1, C), view(l.s1, C), l.m1, l.r1, p[P_G1], view(w.ds1, C), gr[P_G1], gr[P_BE1], g.rows, C, stream));
        const float* dyo = w.ds1;                                                                  // s1 = x + dropout(conv_o(att))
        if (g.pd > 0.f) { XVA_TRY(xva_dropout_apply(view(w.ds1, C), view(w.dyo, C), 0, g.rows * C, g.pd, g.seed, site + 1, stream)); dyo = w.dyo; }
        XVA_TRY(conv_bwd_weight(g, dyo, C, l.att, C, gr[P_WO], gr[P_BO], 1, sk_ws, sk_ws_bytes, stream));   // a 1x1 conv: the gradient buffer IS the GEMM's C
        XVA_TRY(conv_bwd_data(g, dyo, C, p[P_WO], w.datt, C, 1, nullptr, nullptr, stream));
        float* qv = view(l.qkv, 3 * C); float* dq = view(w.dqkv, 3 * C);
        XVA_TRY(xva_relattn_bwd(view(w.datt, C), C, qv, qv + C, qv + 2 * C, 3 * C, p[P_EK], p[P_EV], lens, l.P, w.dS, dq, dq + C, dq + 2 * C, 3 * C, gr[P_EK], gr[P_EV], g.B,
                                g.T, g.H, dk, g.w, 1, g.Tp, PAD, g.pd, g.seed, site, stream));
        XVA_TRY(conv_bwd_weight(g, w.dqkv, 3 * C, l.xm, C, dWqkv, dbqkv, 1, sk_ws, sk_ws_bytes, stream));
        UnpackArgs ua{gr[P_WQ], gr[P_WK], gr[P_WV], gr[P_BQ], gr[P_BK], gr[P_BV], l.ffn ? gr[P_W1] : nullptr, l.ffn ? gr[P_W2] : nullptr, dWqkv, dbqkv,
                      l.ffn ? l.scr + (int64_t)Co * k * F : nullptr, l.ffn ? l.scr : nullptr, C, F, k, Co};
        hipLaunchKernelGGL(unpack_kernel, dim3(grid_of(3ll * C * C + (l.ffn ? (int64_t)F * k * C * 2 : 0), 1024)), dim3(256), 0, s, ua);
        XVA_LAUNCH_CHECK();
        float* dxm = w.dxm[li & 1];
        XVA_TRY(conv_bwd_data(g, w.dqkv, 3 * C, l.wqkv, dxm, C, 1, nullptr, nullptr, stream));
        const int64_t n4 = g.rows * C / 4;
        hipLaunchKernelGGL(add_mask_kernel, dim3((unsigned)xva_cdiv(n4, 256)), dim3(256), 0, s, (float4*)view(dxm, C), (const float4*)view(w.ds1, C), n4, C / 4, g.Tp, lens);   // residual x + y ; xm = x * x_mask
        XVA_LAUNCH_CHECK();
        dx = dxm;
    }
    return xva_seq_to_bct(view((float*)dx, C), d_x_bct, 0, g.B, C, g.T, PAD, 0, stream);
}
