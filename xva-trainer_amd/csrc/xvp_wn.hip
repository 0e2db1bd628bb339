// xvp_wn.hip — the layer loop of xVAPitch's WaveNet stack `WN` (python/xvapitch/wavenet.py:84-109: dilated conv -> fused tanh / sigmoid gate [+ conditioning] ->
// 1x1 conv -> residual / skip split), forward and backward, as two engine calls over xva_gemm (implicit-conv form) and the xva_wn_* kernels.
//
// The posterior encoder (16 layers) and the four coupling blocks of the flow (4 layers each) run through it; their backward passes, one after the other, are the
// tail of the iteration's backward pass: ~12 launches per layer of which only five depend on each other (res / skip backward -> d(1x1 input) -> gate backward ->
// mask -> d(dilated-conv input)).  The host side is a C++ loop.  XVA_XVP_WN_LANE=1 sends the other seven — both weight-gradient products with their slab reduces
// and bias sums, the conditioning gradient — to a side stream ("lane"), ordered by events and joined before the call returns, so that the dependent chain is five
// launches per layer: measured SLOWER inside the iteration (27.9 ms against 26.8; tools/c5_ab.py) — the iteration's other four streams already fill the device, and
// the lane's products then compete with the chain they were meant to unblock — so it is off by default.
// Sequence layout as xva-trainer_amd/xvapitch/wn.py:Seq (time-major (B, PAD + T + PAD, C) in fp32 or bf16, GUARD rows either side, structural zeros);
// weight norm, the conditioning layer and the buffers' allocation stay with the caller.
#include "xva_common.h"
#include "xva_gemm.h"
#include "xva_hip.h"

namespace {
constexpr int PAD = 8, GUARD = 32;

struct G {
    int B, T, Tp, H, k, rate, L, dt, cmp, es;
    int64_t rows, rtot;
};
int geo(const xva_xvp_wn_dims* d, G& g) {
    XVA_CHECK_ARG(d && d->B > 0 && d->T > 0 && d->H > 0 && d->H % 8 == 0 && d->L >= 1 && d->L <= 32 && (d->k & 1) && d->rate >= 1 && (d->dt == 0 || d->dt == 1) &&
                      (d->compute == 0 || d->compute == 1),
                  "xvp_wn: B, T > 0; H a multiple of 8; 1 <= L <= 32; k odd; dt / compute 0 | 1");
    g.B = d->B; g.T = d->T; g.Tp = d->T + 2 * PAD; g.H = d->H; g.k = d->k; g.rate = d->rate; g.L = d->L; g.dt = d->dt; g.cmp = d->compute; g.es = d->dt ? 2 : 4;
    g.rows = (int64_t)g.B * g.Tp; g.rtot = g.rows + 2 * GUARD;
    int64_t dil = 1;
    for (int i = 1; i < g.L; i++) dil *= g.rate;
    XVA_CHECK_ARG(dil * (g.k - 1) / 2 <= PAD, "xvp_wn: dilation beyond %d rows of structural padding", PAD);
    return XVA_OK;
}
struct Lay { char *a, *acts, *xn; };          // kept for the backward: dilated-conv output, gate output, the NEXT layer's input
struct W {
    Lay l[32];
    char *rs, *d_rs[2], *d_acts, *d_a[2];
    int64_t bytes;
};
struct Carver {
    char* base; int64_t off;
    char* take(int64_t n) { n = (n + 255) / 256 * 256; char* p = base ? base + off : nullptr; off += n; return p; }
};
void carve(const G& g, char* base, W& w) {
    Carver c{base, 0};
    const int64_t e = g.es, R = g.rtot, H = g.H;
    for (int i = 0; i < g.L; i++) {
        w.l[i].a = c.take(R * 2 * H * e); w.l[i].acts = c.take(R * H * e);
        w.l[i].xn = i < g.L - 1 ? c.take(R * H * e) : nullptr;
    }
    w.rs = c.take(R * 2 * H * e);
    w.d_rs[0] = c.take(R * 2 * H * e); w.d_rs[1] = c.take(R * 2 * H * e); w.d_acts = c.take(R * H * e); w.d_a[0] = c.take(R * 2 * H * e); w.d_a[1] = c.take(R * 2 * H * e);
    w.bytes = c.off;
}
inline char* vw(const G& g, const void* store, int C) { return (char*)store + (int64_t)GUARD * C * g.es; }

xva_gemm_params base(const G& g) {
    xva_gemm_params p;
    memset(&p, 0, sizeof(p));
    p.batch = 1; p.batch2 = 1; p.alpha = 1.f; p.beta = 1.f; p.splitk = 1; p.mask_mul = 1; p.compute = g.cmp;
    p.mask_mode = XVA_MASK_PAD; p.Tp = g.Tp; p.mask_pad = PAD; p.mask_len = g.T;
    p.a_dtype = p.b_dtype = p.c_dtype = g.dt;
    return p;
}
// the three products in the parameterisation of xvapitch/wn.py conv_fwd / conv_bwd_data / conv_bwd_weight
int conv_fwd(const G& g, const void* x, int Cin, const void* w, const float* bias, void* y, int Cout, int k, int d, void* st) {
    xva_gemm_params p = base(g);
    const int P = d * (k - 1) / 2;
    p.A = (const char*)x + (int64_t)(GUARD - P) * Cin * g.es; p.B = w; p.C = (char*)y + (int64_t)GUARD * Cout * g.es;
    p.M = (int32_t)g.rows; p.N = Cout; p.K = k * Cin; p.lda = Cin; p.ldb = (int64_t)k * Cin; p.ldc = Cout; p.layout = XVA_GEMM_NT; p.bias = bias;
    if (k > 1) { p.a_seglen = Cin; p.a_segadj = (int64_t)d * Cin - Cin; }
    return xva_gemm(&p, st);
}
int conv_bwd_data(const G& g, const void* dy, int Cout, const void* w, void* dx, int Cin, int k, int d, int accumulate, void* st) {
    xva_gemm_params p = base(g);
    const int P = d * (k - 1) / 2;
    p.A = (const char*)dy + (int64_t)(GUARD + P) * Cout * g.es; p.B = w; p.C = (char*)dx + (int64_t)GUARD * Cin * g.es;
    p.M = (int32_t)g.rows; p.N = Cin; p.K = k * Cout; p.lda = Cout; p.ldb = (int64_t)k * Cin; p.ldc = Cin; p.layout = XVA_GEMM_NN; p.accumulate = accumulate;
    if (k > 1) { p.a_seglen = Cout; p.a_segadj = -(int64_t)d * Cout - Cout; p.seglen = Cout; p.seg0 = 0; p.segstride = Cin; }
    return xva_gemm(&p, st);
}
int conv_bwd_weight(const G& g, const void* dy, int Cout, const void* x, int Cin, float* dw, float* db, int k, int d, void* sk, int64_t sk_bytes, void* st) {
    xva_gemm_params p = base(g);
    const int P = d * (k - 1) / 2;
    p.mask_mode = XVA_MASK_NONE; p.Tp = 0; p.mask_pad = 1; p.mask_len = 0; p.c_dtype = 0;
    p.A = (const char*)dy + (int64_t)GUARD * Cout * g.es; p.B = (const char*)x + (int64_t)(GUARD - P) * Cin * g.es; p.C = dw;
    p.M = Cout; p.N = k * Cin; p.K = (int32_t)g.rows; p.lda = Cout; p.ldb = Cin; p.ldc = (int64_t)k * Cin; p.layout = XVA_GEMM_TN; p.accumulate = 1; p.splitk = 0;
    if (k > 1) { p.seglen = Cin; p.seg0 = 0; p.segstride = (int64_t)d * Cin - Cin; }
    p.sk_ws = sk; p.sk_ws_bytes = sk_bytes;
    XVA_TRY(xva_gemm(&p, st));
    return xva_hg_colsum(vw(g, dy, Cout), g.dt, db, g.rows, Cout, 1.f, st);
}

// the weight-gradient lane: one side stream and a few events per host thread, created on first use
struct Lane { hipStream_t s = nullptr; hipEvent_t ev[4]; bool ok = false, tried = false; };
Lane& lane() {
    static thread_local Lane l;
    if (!l.tried) {
        l.tried = true;
        const char* e = getenv("XVA_XVP_WN_LANE");
        if (e && atoi(e) != 0) {                // off by default: see the header comment (measured 27.9 ms with the lane against 26.8 without)
            l.ok = hipStreamCreateWithFlags(&l.s, hipStreamNonBlocking) == hipSuccess;
            for (int i = 0; l.ok && i < 4; ++i) l.ok = hipEventCreateWithFlags(&l.ev[i], hipEventDisableTiming) == hipSuccess;
        }
    }
    return l;
}
#define HIP_OK(x) do { if ((x) != hipSuccess) { xva_set_error("xvp_wn: %s failed", #x); return XVA_ERR_HIP; } } while (0)
}  // namespace

extern "C" int64_t xva_xvp_wn_workspace_bytes(const xva_xvp_wn_dims* d) {
    G g;
    if (geo(d, g) != XVA_OK) return -1;
    W w;
    carve(g, nullptr, w);
    return w.bytes;
}

extern "C" int xva_xvp_wn_forward(const xva_xvp_wn_dims* d, const void* const* tab, const void* x, void* out, const float* gc, const int32_t* lens, void* workspace,
                                  int64_t workspace_bytes, void* stream) {
    G g;
    XVA_TRY(geo(d, g));
    XVA_CHECK_ARG(tab && x && out && lens && workspace && ((uintptr_t)workspace % 16) == 0, "xvp_wn_forward: null / unaligned argument");
    W w;
    carve(g, (char*)workspace, w);
    XVA_CHECK_ARG(workspace_bytes >= w.bytes, "xvp_wn_forward: workspace too small");
    const int H = g.H;
    const void* cur = x;
    int dil = 1;
    for (int i = 0; i < g.L; i++, dil *= g.rate) {
        const void* const* t = tab + (int64_t)i * XVA_XVP_WN_PER_LAYER;       // in_layer eff (2H, k H), bias ; res_skip eff, bias ; their gradient buffers
        const bool last = i == g.L - 1;
        const Lay& l = w.l[i];
        XVA_TRY(conv_fwd(g, cur, H, t[0], (const float*)t[1], l.a, 2 * H, g.k, dil, stream));                              // wavenet.py:91
        XVA_TRY(xva_wn_gate_fwd(vw(g, l.a, 2 * H), gc ? gc + (int64_t)i * 2 * H : nullptr, (int64_t)2 * H * g.L, vw(g, l.acts, H), g.dt, g.B, g.Tp, H, stream));   // :92-99
        // the gate maps the zero pad rows of `a` to tanh(g) * sigmoid(g) != 0 when conditioned: `acts` needs its pads cleared for the weight gradient
        if (gc) XVA_TRY(xva_seq_mask(vw(g, l.acts, H), g.dt, g.B, g.Tp, PAD, H, lens, stream));
        XVA_TRY(conv_fwd(g, l.acts, H, t[2], (const float*)t[3], w.rs, last ? H : 2 * H, 1, 1, stream));                   // :101
        XVA_TRY(xva_wn_res_skip_fwd(vw(g, w.rs, last ? H : 2 * H), vw(g, cur, H), last ? nullptr : vw(g, l.xn, H), vw(g, out, H), g.dt, g.B, g.Tp, PAD, H, last ? 1 : 0, lens,
                                    stream));                                                                              // :103-108
        cur = l.xn;
    }
    return XVA_OK;
}

extern "C" int xva_xvp_wn_backward(const xva_xvp_wn_dims* d, void* const* tab, const void* x, const void* d_out, void* d_x, const float* gc, float* d_gc, const int32_t* lens,
                                   void* workspace, int64_t workspace_bytes, void* sk_ws, int64_t sk_ws_bytes, void* stream) {
    G g;
    XVA_TRY(geo(d, g));
    XVA_CHECK_ARG(tab && x && d_out && d_x && lens && workspace && sk_ws && (!gc == !d_gc), "xvp_wn_backward: null argument (gc and d_gc come together)");
    W w;
    carve(g, (char*)workspace, w);
    XVA_CHECK_ARG(workspace_bytes >= w.bytes, "xvp_wn_backward: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    Lane& ln = lane();
    const bool side = ln.ok;
    void* ls = side ? (void*)ln.s : stream;
    const int H = g.H;
    int dil = 1;
    for (int i = 1; i < g.L; i++) dil *= g.rate;
    if (side) { HIP_OK(hipEventRecord(ln.ev[0], s)); HIP_OK(hipStreamWaitEvent(ln.s, ln.ev[0], 0)); }      // the lane starts behind everything issued so far
    for (int i = g.L - 1; i >= 0; i--, dil /= g.rate) {
        void* const* t = tab + (int64_t)i * XVA_XVP_WN_PER_LAYER;
        const bool last = i == g.L - 1;
        const Lay& l = w.l[i];
        const int Crs = last ? H : 2 * H;
        char* d_rs = w.d_rs[i & 1]; char* d_a = w.d_a[i & 1];
        // the lane finished with this pair of buffers two layers ago?  (event 2 + (i & 1) was recorded behind its work on them)
        if (side && i + 2 <= g.L - 1) HIP_OK(hipStreamWaitEvent(s, ln.ev[2 + (i & 1)], 0));
        XVA_TRY(xva_wn_res_skip_bwd(vw(g, d_x, H), vw(g, d_out, H), vw(g, d_rs, Crs), g.dt, g.B, g.Tp, PAD, H, last ? 1 : 0, lens, stream));
        if (side) { HIP_OK(hipEventRecord(ln.ev[0], s)); HIP_OK(hipStreamWaitEvent(ln.s, ln.ev[0], 0)); }
        XVA_TRY(conv_bwd_weight(g, d_rs, Crs, l.acts, H, (float*)t[6], (float*)t[7], 1, 1, sk_ws, sk_ws_bytes, ls));
        XVA_TRY(conv_bwd_data(g, d_rs, Crs, t[2], w.d_acts, H, 1, 1, 0, stream));
        XVA_TRY(xva_wn_gate_bwd(vw(g, l.a, 2 * H), gc ? gc + (int64_t)i * 2 * H : nullptr, (int64_t)2 * H * g.L, vw(g, w.d_acts, H), vw(g, d_a, 2 * H), g.dt, g.B, g.Tp, H, stream));
        if (side) { HIP_OK(hipEventRecord(ln.ev[1], s)); HIP_OK(hipStreamWaitEvent(ln.s, ln.ev[1], 0)); }
        if (d_gc)                                                   // d(cond)[b] = sum over the item's rows of d_a (pad / dead rows carry zeros)
            XVA_TRY(xva_seq_item_colsum(vw(g, d_a, 2 * H), g.dt, d_gc + (int64_t)i * 2 * H, g.B, g.Tp, 2 * H, (int64_t)2 * H * g.L, ls));
        XVA_TRY(conv_bwd_weight(g, d_a, 2 * H, i == 0 ? x : w.l[i - 1].xn, H, (float*)t[4], (float*)t[5], g.k, dil, sk_ws, sk_ws_bytes, ls));
        if (side) HIP_OK(hipEventRecord(ln.ev[2 + (i & 1)], ln.s));
        // d_x (residual path, already masked by res_skip_bwd's first half when not last) += conv^T(d_a)
        if (!last) XVA_TRY(xva_seq_mask(vw(g, d_x, H), g.dt, g.B, g.Tp, PAD, H, lens, stream));
        XVA_TRY(conv_bwd_data(g, d_a, 2 * H, t[0], d_x, H, g.k, dil, 1, stream));
    }
    if (side) { HIP_OK(hipEventRecord(ln.ev[0], ln.s)); HIP_OK(hipStreamWaitEvent(s, ln.ev[0], 0)); }
    return XVA_OK;
}
