// hg_conv.h — 1-D convolutions over time-major sequence tensors expressed as implicit-convolution GEMMs (xva_gemm):
// forward, backward-data and backward-weight for Conv1d with stride / dilation / groups and for ConvTranspose1d.
// Used by the HiFi-GAN engine (python/hifigan/models.py:17-260); see include/xva_gemm.h for the operand model.
//
// A sequence tensor (Seq) holds nseq items of Hp = padF + T + padB rows x C channels (fp32 or bf16), pad rows zero.
// Two launch modes:
//   merged   — one GEMM over ALL rows of all items (pad rows are computed and masked back to zero).  Needs an affine
//              row map between input and output rows: same geometry for stride 1; Hp_in = s * Hp_out and
//              padF_in = s * padF_out for stride s.  Best when T is small (discriminator tails).
//   per-item — batch = nseq GEMMs of T_out rows each; any geometry; nothing is masked because only valid rows are written.
#pragma once
#include "xva_common.h"
#include "../../include/xva_gemm.h"

struct Seq {
    char* base = nullptr;   // workspace base (bytes)
    int64_t off = 0;        // byte offset of item 0, row 0
    int nseq = 0, T = 0, C = 0, padF = 0, padB = 0, dt = 0;
    int es() const { return dt == XVA_BF16 ? 2 : 4; }
    int Hp() const { return padF + T + padB; }
    int64_t item() const { return (int64_t)Hp() * C; }            // elements
    int64_t rows() const { return (int64_t)nseq * Hp(); }
    void* ptr(int64_t elem = 0) const { return base + off + elem * es(); }
    void* valid(int64_t elem = 0) const { return ptr((int64_t)padF * C + elem); }
    // view of items [i0, i0 + n)
    Seq slice(int i0, int n) const { Seq s = *this; s.off += (int64_t)i0 * item() * es(); s.nseq = n; return s; }
    bool same_geom(const Seq& o) const { return nseq == o.nseq && T == o.T && padF == o.padF && padB == o.padB; }
};

struct ConvW {                 // effective (reparametrised) weights of one conv layer, tap-major [G][Cout_g][k][Cin_g]
    const void* eff = nullptr; // activation dtype
    const float* bias = nullptr;
    float* dweff = nullptr;    // fp32 gradient of eff (same layout)
    float* dbias = nullptr;
    int Cin = 0, Cout = 0, k = 1, s = 1, d = 1, P = 0, groups = 1;
};

struct ConvEpi {
    int a_lrelu = 0; float a_slope = 0.f;          // LeakyReLU applied to the conv INPUT while staging
    int act = XVA_ACT_NONE; float act_slope = 0.f;  // activation on the output
    const Seq* R = nullptr; float alpha = 1.f, beta = 1.f;   // y = alpha * (conv + bias) + beta * R
    int accumulate = 0;
    const Seq* Y2 = nullptr; float y2_slope = 0.f;  // also store lrelu(y, y2_slope) there (Y's geometry)
};

// split-K slab scratch of the weight-gradient GEMMs: set by the engine entry points from their workspace plan
struct HgSplitKScratch { void* ptr = nullptr; int64_t bytes = 0; };
inline HgSplitKScratch& hg_skws() { static thread_local HgSplitKScratch s; return s; }

inline xva_gemm_params hg_gp(int compute, int dt) {
    xva_gemm_params g;
    memset(&g, 0, sizeof(g));
    g.batch = 1; g.batch2 = 1; g.alpha = 1.f; g.beta = 1.f; g.splitk = 1; g.compute = compute; g.mask_mul = 1;
    g.a_dtype = g.b_dtype = g.c_dtype = dt;
    g.sk_ws = hg_skws().ptr; g.sk_ws_bytes = hg_skws().bytes;
    return g;
}
bool xva_gemm_wgrad_res_ok(const xva_gemm_params& p);   // wgrad_res.hip: does the resident-operand weight-gradient kernel take this TN problem?
inline int hg_conv_out_len(int T, const ConvW& w) { return (T + 2 * w.P - w.d * (w.k - 1) - 1) / w.s + 1; }

enum { HG_MERGED = 0, HG_PERITEM = 1 };
inline bool hg_affine(const Seq& X, const Seq& Y, const ConvW& w) { return X.nseq == Y.nseq && X.Hp() == w.s * Y.Hp() && X.padF == w.s * Y.padF; }
// (round-3 A/B, knob removed; kept on): long stride-1 sequences whose length is a whole number of 128-row tiles run per item even where the merged form exists.
// The merged form convolves (and masks) the pad rows and its tile count is whatever nseq * Hp / 128 gives: HiFi-GAN's 128-channel stage is
// 64 x 2112 rows = 1056 tiles = 2.06 rounds of the 512 resident-input workgroups a chip holds (a third, nearly empty round), per item it is
// 64 x 16 = 1024 tiles = exactly two; the 256-channel stage (T = 256, 32 + 32 pad rows) spends 20 % of its merged rows on pads.
inline int hg_peritem_pref() { return 1; }
// Forward only: the merged form also re-zeroes the pad rows of its output, which backward relies on (gradient tensors share workspace slots).
inline int hg_mode(const Seq& X, const Seq& Y, const ConvW& w, bool forward = false) {
    if (!hg_affine(X, Y, w)) return HG_PERITEM;
    if (forward && hg_peritem_pref() && w.s == 1 && Y.T % 128 == 0 && Y.T >= 256) return HG_PERITEM;
    return HG_MERGED;
}

// ---- forward: Y = act(alpha * (conv(lrelu?(X)) + bias) + beta * R) ------------------------------------------------
inline int hg_conv_fwd(const Seq& X, const Seq& Y, const ConvW& w, const ConvEpi& e, int compute, void* st) {
    XVA_CHECK_ARG(X.C == w.Cin && Y.C == w.Cout && X.nseq == Y.nseq, "conv_fwd: channel/batch mismatch");
    XVA_CHECK_ARG(Y.T == hg_conv_out_len(X.T, w), "conv_fwd: output length %d != %d", Y.T, hg_conv_out_len(X.T, w));
    const int Cig = w.Cin / w.groups, Cog = w.Cout / w.groups;
    xva_gemm_params g = hg_gp(compute, X.dt);
    g.layout = XVA_GEMM_NT;
    g.N = Cog; g.K = w.k * Cig;
    g.lda = (int64_t)w.s * X.C; g.ldb = g.K; g.ldc = Y.C;
    g.a_seglen = Cig; g.a_segadj = (int64_t)w.d * X.C - Cig; g.a_rowpitch = X.C;
    g.B = w.eff; g.bias = w.bias;
    g.a_lrelu = e.a_lrelu; g.a_slope = e.a_slope; g.act = e.act; g.act_slope = e.act_slope;
    g.alpha = e.alpha; g.beta = e.beta; g.accumulate = e.accumulate;
    g.batch2 = w.groups; g.sA2 = Cig; g.sB2 = (int64_t)Cog * g.K; g.sC2 = Cog; g.sR2 = Cog;
    if (e.R) { XVA_CHECK_ARG(e.R->same_geom(Y) && e.R->C == Y.C, "conv_fwd: residual geometry"); g.ldr = Y.C; g.r_dtype = e.R->dt; }
    if (e.Y2) { XVA_CHECK_ARG(e.Y2->same_geom(Y) && e.Y2->C == Y.C && e.Y2->dt == Y.dt, "conv_fwd: second output geometry"); g.c2_slope = e.y2_slope; }
    if (hg_mode(X, Y, w, true) == HG_MERGED) {
        g.A = (const char*)X.ptr() - (int64_t)w.P * X.C * X.es();
        g.C = Y.ptr(); g.M = (int)Y.rows();
        if (e.R) g.R = e.R->ptr();
        if (e.Y2) g.C2 = e.Y2->ptr();
        g.mask_mode = XVA_MASK_PAD; g.Tp = Y.Hp(); g.mask_pad = Y.padF; g.mask_len = Y.T;
    } else {
        g.batch = X.nseq; g.sA = X.item(); g.sC = Y.item(); g.sR = Y.item();
        g.A = (const char*)X.valid() - (int64_t)w.P * X.C * X.es();
        g.C = Y.valid(); g.M = Y.T;
        if (e.R) g.R = e.R->valid();
        if (e.Y2) g.C2 = e.Y2->valid();
    }
    g.sbias2 = Cog;
    return xva_gemm(&g, st);
}

// ---- forward of a ResBlock1 pair in one launch (conv_pair.hip): T1 = lrelu(conv1(Xin) + b1, slope1) ; Y = alpha * (conv2(T1) + b2) + beta * R  [; Y2 = lrelu(Y)]
// Xin: the activated block input (x_raw = 0; e.R = the raw input as for hg_conv_fwd) or the raw one (x_raw = 1: LeakyReLU(x_slope) on the operand
// fragments, the residual from the resident tile, e.R is ignored and alpha must equal beta; x_raw = 2: activated in place in LDS, e.R as given).  Returns 1 = launched, 0 = not taken (run the two
// convolutions), < 0 = error.
#include "conv_pair.h"
inline int hg_conv_pair_fwd(const Seq& Xin, int x_raw, float x_slope, const Seq& T1, const Seq& Y, const ConvW& w1, const ConvW& w2, float slope1, const ConvEpi& e,
                            int compute, void* st) {
    const int C = Xin.C;
    if (compute == 0 || Xin.dt != XVA_BF16 || !(C == 32 || C == 64)) return 0;
    if (w1.Cin != C || w1.Cout != C || w2.Cin != C || w2.Cout != C || w1.groups != 1 || w2.groups != 1 || w1.s != 1 || w2.s != 1 || w2.d != 1) return 0;
    const int h1 = w1.d * (w1.k - 1) / 2, h2 = (w2.k - 1) / 2;
    if (!(w1.k & 1) || !(w2.k & 1) || w1.P != h1 || w2.P != h2 || Xin.padF < h1 + h2 || Xin.padB < h1 + h2) return 0;
    if (!T1.same_geom(Xin) || !Y.same_geom(Xin) || T1.C != C || Y.C != C || T1.dt != Xin.dt || Y.dt != Xin.dt || e.a_lrelu || e.act != XVA_ACT_NONE) return 0;
    if (x_raw == 1 && e.alpha != e.beta) return 0;
    xva_gemm_params g = hg_gp(compute, Xin.dt);
    g.layout = XVA_GEMM_NT;
    g.N = C; g.K = w2.k * C; g.ldb = g.K; g.ldc = C; g.M = Y.T;
    g.B = w2.eff; g.bias = w2.bias; g.sbias2 = C;
    g.alpha = e.alpha; g.beta = e.beta; g.accumulate = e.accumulate;
    g.batch = Xin.nseq; g.sC = Y.item(); g.sR = Y.item();
    g.C = Y.valid();
    if (x_raw != 1 && e.R) { if (!e.R->same_geom(Y) || e.R->C != C) return 0; g.R = e.R->valid(); g.ldr = C; g.r_dtype = e.R->dt; }
    if (e.Y2) { if (!e.Y2->same_geom(Y) || e.Y2->C != C || e.Y2->dt != Y.dt) return 0; g.C2 = e.Y2->valid(); g.c2_slope = e.y2_slope; }
    xva_conv_pair q;
    memset(&q, 0, sizeof(q));
    q.X = (const char*)Xin.valid() - (int64_t)(h1 + h2) * C * Xin.es(); q.sX = Xin.item(); q.ldx = C;
    q.W1 = w1.eff; q.bias1 = w1.bias; q.k1 = w1.k; q.d1 = w1.d; q.slope1 = slope1;
    q.T1 = T1.valid(); q.sT1 = T1.item(); q.ldt = C;
    q.x_raw = x_raw; q.x_slope = x_slope;
    return xva_conv_pair_fwd(&g, &q, st) == 0 ? 1 : 0;
}

// ---- backward-data: dX = gate(X) * sum_taps dY (*) W  (+ beta * R) -----------------------------------------------
// stride 1: one GEMM.  stride s: one GEMM per input phase psi (polyphase), each using the taps j = j0 + m*s.
struct BwdEpi {
    const Seq* gate = nullptr; float gate_slope = 0.f;   // multiply by lrelu'(gate) (gate has dX's geometry)
    const Seq* fm = nullptr; float fm_c = 0.f;           // before the gate: + fm_c * sign(gate - fm) (feature matching; fm has gate's geometry and dtype)
    const Seq* R = nullptr; float alpha = 1.f, beta = 1.f;
    int accumulate = 0;
};
inline int hg_conv_bwd_data(const Seq& dY, const Seq& dX, const ConvW& w, const BwdEpi& e, int compute, void* st) {
    XVA_CHECK_ARG(dX.C == w.Cin && dY.C == w.Cout && dX.nseq == dY.nseq, "conv_bwd_data: channel/batch mismatch");
    XVA_CHECK_ARG(w.s == 1 || w.d == 1, "conv_bwd_data: strided + dilated is not used by the path");
    const int Cig = w.Cin / w.groups, Cog = w.Cout / w.groups;
    const int mode = hg_mode(dX, dY, w);
    for (int psi = 0; psi < w.s; ++psi) {
        const int j0 = (psi + w.P) % w.s, c0 = (psi + w.P) / w.s;
        const int ntap = (w.k - j0 + w.s - 1) / w.s;
        if (ntap <= 0) continue;   // (cannot happen for k >= s)
        xva_gemm_params g = hg_gp(compute, dY.dt);
        g.layout = XVA_GEMM_NN;
        g.N = Cig; g.K = ntap * Cog;
        g.lda = dY.C; g.ldb = (int64_t)w.k * Cig; g.ldc = (int64_t)w.s * dX.C;
        // A(q, (m, co)) = dY[(q + c0 - m * d') * C + co], d' = dilation (stride-1 case) or 1
        const int64_t step = (w.s == 1) ? (int64_t)w.d : 1;
        g.a_seglen = Cog; g.a_segadj = -step * dY.C - Cog;
        // B row (m, co) -> eff[co][j0 + m*s][:]
        g.seglen = Cog; g.seg0 = (int64_t)j0 * Cig; g.segstride = (int64_t)w.s * Cig;
        g.B = w.eff;
        g.alpha = e.alpha; g.beta = e.beta; g.accumulate = e.accumulate;
        g.batch2 = w.groups; g.sA2 = Cog; g.sB2 = (int64_t)Cog * w.k * Cig; g.sC2 = Cig; g.sR2 = Cig; g.sG2 = Cig;
        if (e.R) { g.ldr = (int64_t)w.s * dX.C; g.r_dtype = e.R->dt; }
        if (e.gate) { g.ldg = (int64_t)w.s * dX.C; g.g_dtype = e.gate->dt; g.gate_slope = e.gate_slope; }
        if (e.fm) { XVA_CHECK_ARG(e.gate && e.fm->same_geom(*e.gate) && e.fm->C == e.gate->C && e.fm->dt == e.gate->dt, "conv_bwd_data: feature-map geometry"); g.fm_c = e.fm_c; }
        // stride 1: dX[t] = sum_j dY[t + P - j*d] W_j  -> c0 = P (row offset), taps step by d
        const int64_t a_row0 = (w.s == 1) ? (int64_t)w.P : (int64_t)c0;
        if (mode == HG_MERGED) {
            g.A = (const char*)dY.ptr() + a_row0 * dY.C * dY.es();
            g.C = (char*)dX.ptr() + (int64_t)psi * dX.C * dX.es();
            g.M = (int)dY.rows();
            if (e.R) g.R = (const char*)e.R->ptr() + (int64_t)psi * dX.C * e.R->es();
            if (e.gate) g.G = (const char*)e.gate->ptr() + (int64_t)psi * dX.C * e.gate->es();
            if (e.fm) g.F = (const char*)e.fm->ptr() + (int64_t)psi * dX.C * e.fm->es();
            g.mask_mode = XVA_MASK_PAD; g.Tp = dX.Hp(); g.mask_pad = dX.padF; g.mask_len = dX.T; g.mask_mul = w.s; g.mask_add = psi;
        } else {
            const int Q = (dX.T - psi + w.s - 1) / w.s;
            if (Q <= 0) continue;
            g.batch = dX.nseq; g.sA = dY.item(); g.sC = dX.item(); g.sR = dX.item(); g.sG = dX.item();
            g.A = (const char*)dY.valid() + a_row0 * dY.C * dY.es();
            g.C = (char*)dX.valid() + (int64_t)psi * dX.C * dX.es();
            g.M = Q;
            if (e.R) g.R = (const char*)e.R->valid() + (int64_t)psi * dX.C * e.R->es();
            if (e.gate) g.G = (const char*)e.gate->valid() + (int64_t)psi * dX.C * e.gate->es();
            if (e.fm) g.F = (const char*)e.fm->valid() + (int64_t)psi * dX.C * e.fm->es();
        }
        XVA_TRY(xva_gemm(&g, st));
    }
    return XVA_OK;
}

// ---- backward-weight: dWeff[g][co][(j, ci)] += alpha * sum_rows dY[r][co] * lrelu?(X)[s*r + j*d - P][ci] ----------
// db != nullptr: the layer's bias gradient db[c] += alpha * sum_rows dY[row][c] as well — inside the resident-operand kernel when it takes the product
// (xva_gemm colsum_out), by the column-sum kernel otherwise
extern "C" int xva_hg_colsum(const void* X, int dt, float* out, int64_t rows, int C, float scale, void* stream);
// db_deferred != nullptr: when the kernel does not take the bias gradient, leave it to the caller (*db_deferred = true: e.g. a batched column-sum launch later)
inline int hg_conv_bwd_weight(const Seq& dY, const Seq& X, const ConvW& w, int x_lrelu, float x_slope, float alpha, int compute, void* st, float* db = nullptr,
                              bool* db_deferred = nullptr) {
    XVA_CHECK_ARG(X.C == w.Cin && dY.C == w.Cout && X.nseq == dY.nseq && w.dweff, "conv_bwd_weight: mismatch");
    const int Cig = w.Cin / w.groups, Cog = w.Cout / w.groups;
    xva_gemm_params g = hg_gp(compute, dY.dt);
    g.layout = XVA_GEMM_TN;
    g.C = w.dweff; g.c_dtype = XVA_F32; g.alpha = alpha;
    g.batch2 = w.groups; g.sC2 = (int64_t)Cog * w.k * Cig;
    const bool merged = hg_affine(X, dY, w);          // the reduction runs over all rows either way: pad rows are zeros
    const void* dy0 = merged ? dY.ptr() : dY.valid();
    const void* x0 = (const char*)(merged ? X.ptr() : X.valid()) - (int64_t)w.P * X.C * X.es();
    // The 128-row M tile wants the LARGER of (Cout_g, k*Cin_g) on M: for narrow layers (Cout_g <= 64) compute dW^T = Xcat^T dY
    // (M = k*Cin_g taps x channels as segmented columns of A, N = Cout_g on the narrow N tile) and store it transposed.
    bool swap = Cog <= 64 && w.k * Cig > Cog;
    if (swap) {   // the resident-operand kernel (wgrad_res.h) wants the natural form: dY on M, (tap, channel) column segments on N
        xva_gemm_params t = g;
        t.M = Cog; t.N = w.k * Cig; t.lda = dY.C; t.ldb = (int64_t)w.s * X.C; t.ldc = t.N; t.a_rowpitch = X.C;
        t.seglen = Cig; t.segstride = (int64_t)w.d * X.C - Cig; t.b_lrelu = x_lrelu; t.sA2 = Cog; t.sB2 = Cig;
        t.A = dy0; t.B = x0; t.accumulate = 1; t.splitk = 0;
        if (merged) t.K = (int)dY.rows();
        else { t.K = (int)((int64_t)X.nseq * dY.T); t.kb_len = dY.T; t.kb_sA = dY.item(); t.kb_sB = X.item(); }
        if (xva_gemm_wgrad_res_ok(t)) swap = false;
    }
    if (!swap) {
        g.M = Cog; g.N = w.k * Cig;
        g.lda = dY.C; g.ldb = (int64_t)w.s * X.C; g.ldc = g.N; g.a_rowpitch = X.C;
        g.seglen = Cig; g.segstride = (int64_t)w.d * X.C - Cig; g.seg0 = 0;
        g.b_lrelu = x_lrelu; g.b_slope = x_slope;
        g.sA2 = Cog; g.sB2 = Cig;
        g.A = dy0; g.B = x0;
    } else {
        g.M = w.k * Cig; g.N = Cog;
        g.lda = (int64_t)w.s * X.C; g.ldb = dY.C; g.ldc = w.k * Cig; g.c_trans = 1;
        g.a_seglen = Cig; g.a_segadj = (int64_t)w.d * X.C - Cig;
        g.a_lrelu = x_lrelu; g.a_slope = x_slope;
        g.sA2 = Cig; g.sB2 = Cog;
        g.A = x0; g.B = dy0;
    }
    if (merged) {
        g.K = (int)dY.rows();
        g.accumulate = 1; g.splitk = 0;   // xva_gemm sizes the split to its tile grid and the slab scratch
    } else {
        // per-item geometry: the reduction still runs over ALL items in one split-K GEMM (K blocks of T_out rows)
        g.K = (int)((int64_t)X.nseq * dY.T); g.kb_len = dY.T;
        if (!swap) { g.kb_sA = dY.item(); g.kb_sB = X.item(); } else { g.kb_sA = X.item(); g.kb_sB = dY.item(); }
        g.accumulate = 1; g.splitk = 0;   // xva_gemm sizes the split to its tile grid and the slab scratch
    }
    if (db) {
        static const int fused = 1;
        if (fused && !swap && xva_gemm_takes_colsum(&g)) g.colsum_out = db;
        else if (db_deferred) *db_deferred = true;
        else XVA_TRY(xva_hg_colsum(dY.ptr(), dY.dt, db, dY.rows(), dY.C, alpha, st));
    }
    return xva_gemm(&g, st);
}

// ---- ConvTranspose1d(Cin -> Cout, k, stride s, padding p = (k - s) / 2), k % s == 0 -------------------------------
struct ConvTW {
    const void* effF = nullptr;   // [phase][Cout][ntap * Cin]   (forward, per output phase)
    const void* effB = nullptr;   // [Cin][k * Cout]             (backward-data = strided conv of dY)
    const float* bias = nullptr;
    float* dweff = nullptr;       // fp32 [Cin][k * Cout]
    float* dbias = nullptr;
    int Cin = 0, Cout = 0, k = 0, s = 0, p = 0;
};
// Y[s*q + phi] = bias + sum_m lrelu(X)[q + c0(phi) - m] * W[:, :, j0(phi) + m*s]      (per item; any geometry)
inline int hg_convT_fwd(const Seq& X, const Seq& Y, const ConvTW& w, int a_lrelu, float a_slope, int compute, void* st,
                        const Seq* Y2 = nullptr, float y2_slope = 0.f) {   // Y2: also store lrelu(Y, y2_slope) (Y's geometry)
    XVA_CHECK_ARG(X.C == w.Cin && Y.C == w.Cout && X.nseq == Y.nseq && Y.T == X.T * w.s, "convT_fwd: geometry mismatch");
    XVA_CHECK_ARG(!Y2 || (Y2->same_geom(Y) && Y2->C == Y.C && Y2->dt == Y.dt), "convT_fwd: second output geometry");
    const int ntap = w.k / w.s;
    XVA_CHECK_ARG(X.padF >= ntap && X.padB >= ntap, "convT_fwd: input pads too small");
    for (int phi = 0; phi < w.s; ++phi) {
        const int c0 = (phi + w.p) / w.s;
        xva_gemm_params g = hg_gp(compute, X.dt);
        g.layout = XVA_GEMM_NT;
        g.M = X.T; g.N = w.Cout; g.K = ntap * w.Cin;
        g.lda = X.C; g.ldb = g.K; g.ldc = (int64_t)w.s * Y.C;
        g.a_seglen = w.Cin; g.a_segadj = -(int64_t)X.C - w.Cin;
        g.A = (const char*)X.valid() + (int64_t)c0 * X.C * X.es();
        g.B = (const char*)w.effF + (int64_t)phi * w.Cout * g.K * X.es();
        g.C = (char*)Y.valid() + (int64_t)phi * Y.C * Y.es();
        if (Y2) { g.C2 = (char*)Y2->valid() + (int64_t)phi * Y.C * Y.es(); g.c2_slope = y2_slope; }
        g.bias = w.bias; g.a_lrelu = a_lrelu; g.a_slope = a_slope;
        g.batch = X.nseq; g.sA = X.item(); g.sC = Y.item();
        XVA_TRY(xva_gemm(&g, st));
    }
    return XVA_OK;
}
// dX[t][ci] = lrelu'(X) * sum_{j, co} dY[s*t + j - p][co] * W[ci][co][j]   (a strided conv of dY; per item)
inline int hg_convT_bwd_data(const Seq& dY, const Seq& dX, const ConvTW& w, const Seq* gate, float gate_slope, int compute, void* st) {
    XVA_CHECK_ARG(dX.C == w.Cin && dY.C == w.Cout && dX.nseq == dY.nseq && dY.T == dX.T * w.s, "convT_bwd_data: geometry mismatch");
    XVA_CHECK_ARG(dY.padF >= w.p && dY.padB >= w.k, "convT_bwd_data: dY pads too small");
    xva_gemm_params g = hg_gp(compute, dY.dt);
    g.layout = XVA_GEMM_NT;
    g.M = dX.T; g.N = w.Cin; g.K = w.k * w.Cout;
    g.lda = (int64_t)w.s * dY.C; g.ldb = g.K; g.ldc = dX.C;      // taps are consecutive rows of dY: no segment adjustment
    g.A = (const char*)dY.valid() - (int64_t)w.p * dY.C * dY.es();
    g.B = w.effB; g.C = dX.valid();
    g.batch = dX.nseq; g.sA = dY.item(); g.sC = dX.item();
    if (gate) { g.G = gate->valid(); g.ldg = dX.C; g.sG = gate->item(); g.g_dtype = gate->dt; g.gate_slope = gate_slope; }
    return xva_gemm(&g, st);
}
// dW[ci][(j, co)] += sum_t lrelu(X)[t][ci] * dY[s*t + j - p][co]
inline int hg_convT_bwd_weight(const Seq& dY, const Seq& X, const ConvTW& w, int x_lrelu, float x_slope, int compute, void* st) {
    xva_gemm_params g = hg_gp(compute, dY.dt);
    g.layout = XVA_GEMM_TN;
    g.M = w.Cin; g.N = w.k * w.Cout; g.K = X.T;
    g.lda = X.C; g.ldb = (int64_t)w.s * dY.C; g.ldc = g.N;
    g.A = X.valid();
    g.B = (const char*)dY.valid() - (int64_t)w.p * dY.C * dY.es();
    g.C = w.dweff; g.c_dtype = XVA_F32;
    g.a_lrelu = x_lrelu; g.a_slope = x_slope;
    g.K = (int)((int64_t)X.nseq * X.T); g.kb_len = X.T; g.kb_sA = X.item(); g.kb_sB = dY.item();
    g.accumulate = 1; g.splitk = 0;
    return xva_gemm(&g, st);
}
