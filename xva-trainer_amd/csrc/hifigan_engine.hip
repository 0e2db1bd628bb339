// hifigan_engine.hip — HiFi-GAN v1 generator + MPD + MSD forward / backward as fixed C++ launch schedules.
//
// Reference: Generator / ResBlock1 (python/hifigan/models.py:17-128), DiscriminatorP / MultiPeriodDiscriminator
// (:140-200), DiscriminatorS / MultiScaleDiscriminator (:203-260), losses (:263-294) and the D + G step of
// python/hifigan/xva_train.py:479-515.  Every convolution with C_in, C_out > 1 runs on the MFMA implicit-conv GEMM
// (hg_conv.h) over time-major sequence tensors, with LeakyReLU fused into the consumer's operand staging, bias /
// residual / (sum of resblocks)/3 / tanh fused into epilogues and LeakyReLU backward fused as an epilogue gate.
// The 1-channel layers at the waveform boundary, the reparametrisations and the losses are the kernels of hg_ops.hip.
//
// Parameters: two flat fp32 buffers in the checkpoint's own tensor layouts — G (generator) and D (mpd.* then msd.*,
// trainable tensors first, then the spectral-norm power-iteration buffers).  Gradients mirror them.
#include "hg_conv.h"
#include "hg_wn.h"
#include "../../include/xva_hip.h"
#include <memory>
#include <mutex>
#include <string>
#include <vector>

extern "C" {
int xva_hg_mel_to_tm(const float*, void*, int, int, int, int, int, int, void*);
int xva_hg_cin1_fwd(const float*, const float*, const float*, void*, int, int, int, int, int, int, int, int, int, int, float, void*);
int xva_hg_cin1_bwd_weight(const float*, const void*, int, float*, float*, int, int, int, int, int, int, int, int, int, void*);
int xva_hg_cin1_bwd_data(const void*, int, const float*, float*, int, int, int, int, int, int, int, int, int, int, void*);
int xva_hg_cout1_bwd_data(const void*, const float*, const void*, void*, int, int64_t, int, int, int, int, int, int, int, int, float, void*);
int xva_hg_cout1_bwd_weight(const void*, const void*, float*, float*, int, int64_t, int, int, int, int, int, float, void*);
int xva_hg_avgpool_fwd(const float*, float*, int, int, void*);
int xva_hg_avgpool_bwd(const float*, float*, int, int, int, void*);
int xva_hg_reduce(const void*, const void*, int, int, int, int, int, int, int, float, float*, void*);
int xva_hg_seed_grad(const void*, const void*, void*, int, int, int, int, int, int, float, float, int, int, float, int, void*);
int xva_hg_seq1_to_wav(const void*, int, float*, int, int, int, int, void*);
int xva_hg_lrelu_copy(const void* src, void* dst, int dt, int64_t n, float slope, void* stream);
int xva_hg_tanh_bwd(const float*, const void*, void*, int, int, int, int, int, void*);
int xva_hg_colsum(const void*, int, float*, int64_t, int, float, void*);
int xva_hg_weight_norm_fwd(const float*, const float*, void*, void*, float*, int, int, int, int, int, int, int, void*);
int xva_hg_weight_norm_bwd(const float*, const float*, const float*, const float*, float*, float*, int, int, int, int, void*);
int xva_hg_spectral_norm_fwd(const float*, float*, float*, void*, float*, int, int, int, int, float*, void*);
int xva_hg_spectral_norm_bwd(const float*, const float*, const float*, const float*, const float*, float*, int, int, int, float*, void*);
int xva_hg_sn_scale(const float*, const float*, void*, int, int, int, int, void*);
int xva_hg_im2col1(const float*, void*, int, int, int, int, int, int, int, int, int, int, void*);
int xva_hg_col2im1(const void*, int, float*, int, int, int, int, int, int, int, int, int, int, void*);
int xva_hg_pad_cols(const float*, void*, int, int, int, int, void*);
int xva_hg_unpad_cols_add(const float*, float*, int, int, int, void*);
int xva_hg_add_f32(float* dst, const float* src, int64_t n, void* stream);
int xva_hg_add_item_vec(void* seq, int dt, const float* vec, int B, int Hp, int padF, int T, int C, void* stream);
}

void xva_prof_tag(int tag);

// forward of the 32 / 64-channel ResBlock pairs: see gen_forward.  env XVA_HG_PAIR; xva_hg_set_pair_mode for in-process A/B
static int g_hg_pair = [] { const char* e = getenv("XVA_HG_PAIR"); return e ? atoi(e) : 1; }();
extern "C" int xva_hg_set_pair_mode(int mode) { int old = g_hg_pair; g_hg_pair = mode; return old; }

namespace {

// profiling knob (tools/hg_disc_split.py): bit di set = discriminator di (MPD 0..4, MSD 5..7) runs; results are meaningless with bits cleared
static int g_disc_mask = [] { const char* e = getenv("XVA_HG_DISC_MASK"); return e ? (int)strtol(e, nullptr, 0) : 0xff; }();
static int disc_mask() { return g_disc_mask; }

constexpr float SLOPE = 0.1f;
constexpr int GUARD = 32;           // guard rows before / after every sequence tensor
constexpr int NPER = 5;
const int PERIODS[NPER] = {2, 3, 5, 7, 11};
const int UPS_RATE[4] = {8, 8, 2, 2}, UPS_K[4] = {16, 16, 4, 4};
const int RES_K[3] = {3, 7, 11}, RES_D[3] = {1, 3, 5};
enum { LK_WN = 0, LK_WNT = 1, LK_SN = 2, LK_PLAIN = 3 };   // LK_PLAIN: nn.Conv1d without a reparametrisation (the VITS decoder's conv_pre / conv_post / cond_layer)

struct TInfo { std::string name; int64_t off, numel; int ndim; int64_t shape[4]; int kind; };   // kind 0 trainable, 2 buffer

struct Layer {
    int kind = LK_WN;
    int Cin = 0, Cout = 0, k = 1, s = 1, d = 1, P = 0, groups = 1;
    bool has_bias = true;   // LK_PLAIN only: conv_post of the VITS decoder has none
    bool dense = false;     // grouped convolution with few channels per group (4: VITS scale discriminator) run in SUPER-GROUPS of dense_cin() input channels,
                            // each a dense product over a block-diagonal effective weight (hg_ops.hip weight norm kind 2): MFMA-sized K segments at 16x
                            // the group's flops instead of Cin / 4 x
    int dense_cin() const { return Cin < 64 ? Cin : 64; }
    int conv_groups() const { return dense ? Cin / dense_cin() : groups; }
    bool aux = false;       // not a sequence convolution (cond_layer: a (B, cond) x (cond, C) product): no effective-weight copy
    int64_t bias = -1, wg = -1, wv = -1, bu = -1, bv = -1;   // offsets (floats) in the flat parameter buffer
    // workspace byte offsets
    int64_t eff[2] = {-1, -1}, effB = -1, eff32[2] = {-1, -1}, norm[2] = {-1, -1}, su[2] = {-1, -1}, sv[2] = {-1, -1}, dweff[2] = {-1, -1};
    int64_t wp[2] = {-1, -1}, dwp[2] = {-1, -1};   // Cin == 1 layers: taps padded to kp columns (GEMM form of the boundary conv)
    int kp() const { return k <= 8 ? 8 : 16; }
    int D0() const { return kind == LK_WNT ? Cin : Cout; }
    int D1() const { return kind == LK_WNT ? Cout : Cin / groups; }
    int64_t wnumel() const { return (int64_t)D0() * D1() * k; }
    int64_t effnumel() const { return dense ? (int64_t)Cout * dense_cin() * k : wnumel(); }
};

struct Net {
    std::vector<TInfo> t;
    std::vector<Layer> L;
    int64_t total = 0, trainable = 0;
    int add_layer(const std::string& pre, Layer l, std::vector<TInfo>* buffers) {
        auto add = [&](const std::string& n, std::initializer_list<int64_t> shape, int kind, std::vector<TInfo>* dst) {
            TInfo ti; ti.name = n; ti.kind = kind; ti.ndim = (int)shape.size(); ti.numel = 1;
            int i = 0; for (auto s : shape) { ti.shape[i++] = s; ti.numel *= s; }
            for (; i < 4; ++i) ti.shape[i] = 1;
            ti.off = -1; dst->push_back(ti); return (int)dst->size() - 1;
        };
        const int64_t d0 = l.D0(), d1 = l.D1();
        if (l.kind == LK_PLAIN) {   // torch order: weight, bias
            l.wv = add(pre + "weight", {d0, d1, l.k}, 0, &t);
            l.bias = l.has_bias ? add(pre + "bias", {l.Cout}, 0, &t) : -1;
            L.push_back(l);
            return (int)L.size() - 1;
        }
        int ib = add(pre + "bias", {l.kind == LK_WNT ? l.Cout : l.Cout}, 0, &t);
        l.bias = ib;   // temporarily the tensor index; resolved to offsets in finalize()
        if (l.kind == LK_SN) {
            l.wv = add(pre + "weight_orig", {d0, d1, l.k}, 0, &t);
            l.bu = add(pre + "weight_u", {d0}, 2, buffers);
            l.bv = add(pre + "weight_v", {d1 * l.k}, 2, buffers);
        } else {
            if (conv2d) { l.wg = add(pre + "weight_g", {d0, 1, 1, 1}, 0, &t); l.wv = add(pre + "weight_v", {d0, d1, l.k, 1}, 0, &t); }
            else { l.wg = add(pre + "weight_g", {d0, 1, 1}, 0, &t); l.wv = add(pre + "weight_v", {d0, d1, l.k}, 0, &t); }
        }
        L.push_back(l);
        return (int)L.size() - 1;
    }
    bool conv2d = false;
    void finalize(std::vector<TInfo>& buffers) {
        for (auto& ti : t) { ti.off = total; total += (ti.numel + 3) & ~(int64_t)3; }
        trainable = total;
        int nb0 = (int)t.size();
        for (auto& ti : buffers) { ti.off = total; total += (ti.numel + 3) & ~(int64_t)3; t.push_back(ti); }
        for (auto& l : L) {
            if (l.bias >= 0) l.bias = t[l.bias].off;
            if (l.wg >= 0) l.wg = t[l.wg].off;
            l.wv = t[l.wv].off;
            if (l.bu >= 0) { l.bu = t[nb0 + l.bu].off; l.bv = t[nb0 + l.bv].off; }
        }
    }
};

// ---- generator layer indices ----
// gin = 80, gcond = 0, vits = false: HiFi-GAN v1 (python/hifigan/models.py:75-128).  vits: xVAPitch's waveform decoder
// (python/xvapitch/hifigan.py:156-262 as built at model.py:134-149): `gin` latent channels in, conv_pre / conv_post WITHOUT weight norm,
// conv_post without bias, cond_layer = Conv1d(gcond, 512, 1) on the speaker vector added to conv_pre's output; same ups / resblocks.
struct GenNet : Net {
    int pre, ups[4], rc1[12][3], rc2[12][3], post, cond = -1;
    int gin, gcond; bool vits;
    GenNet(int gin_ = 80, int gcond_ = 0, bool vits_ = false) : gin(gin_), gcond(gcond_), vits(vits_) {
        std::vector<TInfo> buf;
        Layer l; l.Cin = gin; l.Cout = 512; l.k = 7; l.P = 3; if (vits) l.kind = LK_PLAIN; pre = add_layer("conv_pre.", l, &buf);
        int ch = 512;
        for (int i = 0; i < 4; ++i) {
            Layer u; u.kind = LK_WNT; u.Cin = ch; u.Cout = ch / 2; u.k = UPS_K[i]; u.s = UPS_RATE[i]; u.P = (UPS_K[i] - UPS_RATE[i]) / 2;
            ups[i] = add_layer("ups." + std::to_string(i) + ".", u, &buf);
            ch /= 2;
        }
        ch = 512;
        for (int i = 0; i < 4; ++i) {
            ch /= 2;
            for (int j = 0; j < 3; ++j) {
                int rb = i * 3 + j;
                for (int m = 0; m < 3; ++m) {
                    Layer c; c.Cin = c.Cout = ch; c.k = RES_K[j]; c.d = RES_D[m]; c.P = (RES_K[j] * RES_D[m] - RES_D[m]) / 2;
                    rc1[rb][m] = add_layer("resblocks." + std::to_string(rb) + ".convs1." + std::to_string(m) + ".", c, &buf);
                }
                for (int m = 0; m < 3; ++m) {
                    Layer c; c.Cin = c.Cout = ch; c.k = RES_K[j]; c.d = 1; c.P = (RES_K[j] - 1) / 2;
                    rc2[rb][m] = add_layer("resblocks." + std::to_string(rb) + ".convs2." + std::to_string(m) + ".", c, &buf);
                }
            }
        }
        Layer p; p.Cin = 32; p.Cout = 1; p.k = 7; p.P = 3; if (vits) { p.kind = LK_PLAIN; p.has_bias = false; } post = add_layer("conv_post.", p, &buf);
        if (gcond > 0) { Layer cl; cl.kind = LK_PLAIN; cl.aux = true; cl.Cin = gcond; cl.Cout = 512; cl.k = 1; cond = add_layer("cond_layer.", cl, &buf); }
        finalize(buf);
    }
};
// vits: xVAPitch's VitsDiscriminator (python/xvapitch/model.py:1548-1640): nets.0 = ONE scale discriminator (weight norm; Conv1d 1->16 k15,
// 16->64 / 64->256 / 256->1024 / 1024->1024 k41 s4 in groups of FOUR input channels, 1024->1024 k5, post 1024->1 k3), nets.1-5 = the period
// discriminators of python/xvapitch/hifigan.py:301-367 (same as HiFi-GAN's).  The grouped layers run as dense super-groups (Layer::dense).
struct DiscNet : Net {
    int mpd[NPER][6], msd[3][8];
    bool vits;
    DiscNet(bool vits_ = false) : vits(vits_) {
        std::vector<TInfo> buf;
        const int pc[5][2] = {{1, 32}, {32, 128}, {128, 512}, {512, 1024}, {1024, 1024}};
        if (vits) {   // reference key order: nets.0 (scale) first
            const int vc[7][6] = {{1, 16, 15, 1, 1, 7}, {16, 64, 41, 4, 4, 20}, {64, 256, 41, 4, 16, 20}, {256, 1024, 41, 4, 64, 20},
                                  {1024, 1024, 41, 4, 256, 20}, {1024, 1024, 5, 1, 1, 2}, {1024, 1, 3, 1, 1, 1}};
            for (int i = 0; i < 7; ++i) {
                Layer l; l.kind = LK_WN;
                l.Cin = vc[i][0]; l.Cout = vc[i][1]; l.k = vc[i][2]; l.s = vc[i][3]; l.groups = vc[i][4]; l.P = vc[i][5]; l.dense = l.groups > 1;
                msd[0][i] = add_layer(std::string("nets.0.") + (i < 6 ? "convs." + std::to_string(i) + "." : std::string("conv_post.")), l, &buf);
            }
        }
        conv2d = true;
        for (int d = 0; d < NPER; ++d) {
            std::string pre = vits ? "nets." + std::to_string(d + 1) + "." : "mpd.discriminators." + std::to_string(d) + ".";
            for (int i = 0; i < 5; ++i) {
                Layer l; l.Cin = pc[i][0]; l.Cout = pc[i][1]; l.k = 5; l.s = i < 4 ? 3 : 1; l.P = 2;
                mpd[d][i] = add_layer(pre + "convs." + std::to_string(i) + ".", l, &buf);
            }
            Layer p; p.Cin = 1024; p.Cout = 1; p.k = 3; p.P = 1; mpd[d][5] = add_layer(pre + "conv_post.", p, &buf);
        }
        conv2d = false;
        const int sc[8][6] = {{1, 128, 15, 1, 1, 7}, {128, 128, 41, 2, 4, 20}, {128, 256, 41, 2, 16, 20}, {256, 512, 41, 4, 16, 20},
                              {512, 1024, 41, 4, 16, 20}, {1024, 1024, 41, 1, 16, 20}, {1024, 1024, 5, 1, 1, 2}, {1024, 1, 3, 1, 1, 1}};
        for (int d = 0; d < 3 && !vits; ++d) {
            std::string pre = "msd.discriminators." + std::to_string(d) + ".";
            for (int i = 0; i < 8; ++i) {
                Layer l; l.kind = d == 0 ? LK_SN : LK_WN;
                l.Cin = sc[i][0]; l.Cout = sc[i][1]; l.k = sc[i][2]; l.s = sc[i][3]; l.groups = sc[i][4]; l.P = sc[i][5];
                msd[d][i] = add_layer(pre + (i < 7 ? "convs." + std::to_string(i) + "." : std::string("conv_post.")), l, &buf);
            }
        }
        finalize(buf);
    }
};
const GenNet& gnet() { static GenNet n; return n; }
// the VITS decoder variants met so far (latent 192 / 256 x speaker-vector width): built once each, never freed
const GenNet& vits_gnet(int gin, int gcond) {
    static std::mutex mu;
    static std::vector<std::unique_ptr<GenNet>> nets;
    std::lock_guard<std::mutex> lk(mu);
    for (auto& n : nets) if (n->gin == gin && n->gcond == gcond) return *n;
    nets.emplace_back(new GenNet(gin, gcond, true));
    return *nets.back();
}
const DiscNet& dnet() { static DiscNet n; return n; }
const DiscNet& vits_dnet() { static DiscNet n(true); return n; }

// ------------------------------------------------------------------ workspace plan ----
// env XVA_HG_WIDE_PADS=1: the round-2 geometry of the merged discriminator tails (A/B switch)
static const int g_hg_wide_pads = [] { const char* e = getenv("XVA_HG_WIDE_PADS"); return e ? atoi(e) : 0; }();
struct Bump {
    int64_t cur = 0;
    int64_t take(int64_t bytes) { int64_t o = cur; cur += (bytes + 255) & ~(int64_t)255; return o; }
};
struct SeqSpec { int64_t off; int nseq, T, C, padF, padB; };
struct Plan {
    int B, seg, dt, es, T[5], Cst[5];
    std::vector<Layer> gl, dl;          // layers with workspace offsets resolved
    // generator activations
    SeqSpec xin, h0, u[4], xt1[12][3], xr[12][2], xs[4], y;
    SeqSpec ua[4], xra[12][2];          // lrelu(u), lrelu(xr): what the resblock convolutions (and their weight gradients) read
    // generator backward scratch
    SeqSpec g_dxs, g_da, g_db, g_dt1, g_du, g_dy;
    SeqSpec g_daj[3], g_dbj[3], g_dt1jm[3][3];   // per-resblock copies: the weight-gradient lane reads them while the chain moves on
    // discriminators: per MPD period: t1..t6 ; per MSD scale: t1..t8 (real+fake stacked: nseq = 2B*p / 2B; SN scale 0: two sets)
    SeqSpec pt[NPER][7], st[3][2][9];
    SeqSpec pd[NPER][7], sd[3][2][9];   // gradient tensors (same geometry)
    SeqSpec pxc[NPER][2], sxc[3][2][2];  // im2col of the waveform for conv0 [.][0] and its gradient [.][1]
    int64_t wav_s[3][2];                // pooled waveforms (fp32): [scale][real/fake] ; scale 0 = the inputs themselves
    int64_t dwav_s[3];                  // gradient w.r.t. the (pooled) fake waveforms
    int Tw[3];
    int64_t dw_g[2], dw_d[2];           // [begin, end) of the generator / discriminator effective-weight gradient regions
    int64_t sn_tmp, losses, skws[4], skws_bytes, dwav_lane[4], total;   // per stream lane: split-K slabs, partial d(waveform)
    // VITS decoder variant (gnetp != &gnet()): d(input), the speaker projection and its gradient, a zero bias / dummy bias gradient for conv_post
    const GenNet* gnetp = nullptr;
    const DiscNet* dnetp = nullptr;     // null: no discriminators in this plan
    SeqSpec g_dxin;
    int64_t cvec = -1, dcvec = -1, zbias = -1, dumdb = -1;
};

SeqSpec mk(Bump& b, int es, int nseq, int T, int C, int padF, int padB) {
    SeqSpec s; s.nseq = nseq; s.T = T; s.C = C; s.padF = padF; s.padB = padB;
    int64_t rows = (int64_t)nseq * (padF + T + padB) + 2 * GUARD;
    int64_t o = b.take(rows * C * es);
    s.off = o + (int64_t)GUARD * C * es;
    return s;
}
Seq seq(const SeqSpec& s, char* base, int dt) {
    Seq q; q.base = base; q.off = s.off; q.nseq = s.nseq; q.T = s.T; q.C = s.C; q.padF = s.padF; q.padB = s.padB; q.dt = dt;
    return q;
}

void plan_layer_ws(Layer& l, Bump& b, int es, bool grads) {
    if (l.aux) return;
    const int64_t n = l.effnumel();
    const int passes = l.kind == LK_SN ? 2 : 1;
    for (int p = 0; p < passes; ++p) {
        l.eff[p] = b.take(n * es + 64);
        if (l.Cin == 1 || l.Cout == 1) l.eff32[p] = b.take(n * 4);
        l.norm[p] = b.take((l.kind == LK_SN ? 4 : l.D0()) * 4);
        if (l.kind == LK_SN) { l.su[p] = b.take(l.D0() * 4); l.sv[p] = b.take((int64_t)l.D1() * l.k * 4); }
        if (l.Cin == 1) l.wp[p] = b.take((int64_t)l.Cout * l.kp() * es + 64);
    }
    if (l.kind == LK_WNT) l.effB = b.take(n * es + 64);
}
// effective-weight gradients of one network in ONE contiguous region (zeroed by a single memset per backward)
void plan_layer_grads(std::vector<Layer>& L, Bump& b, int64_t* begin, int64_t* end) {
    *begin = b.cur;
    for (auto& l : L) {
        if (l.aux) continue;
        const int passes = l.kind == LK_SN ? 2 : 1;
        for (int p = 0; p < passes; ++p) {
            l.dweff[p] = b.take(l.effnumel() * 4);
            if (l.Cin == 1) l.dwp[p] = b.take((int64_t)l.Cout * l.kp() * 4);
        }
    }
    *end = b.cur;
}

// gn / dn: the networks this plan serves — (null, null) = HiFi-GAN v1 generator + MPD + MSD; (vits decoder, null) = the decoder alone;
// (null, vits discriminator) = VitsDiscriminator alone
int make_plan(const xva_hg_dims* d, Plan* p, const GenNet* gn = nullptr, const DiscNet* dn = nullptr) {
    XVA_CHECK_ARG(d && d->B > 0 && d->seg >= 256 && d->seg % 256 == 0, "hifigan: bad dims (segment must be a positive multiple of 256)");
    XVA_CHECK_ARG(d->dt == XVA_F32 || d->dt == XVA_BF16, "hifigan: bad dtype");
    p->B = d->B; p->seg = d->seg; p->dt = d->dt; p->es = d->dt == XVA_BF16 ? 2 : 4;
    const int es = p->es, B = d->B;
    p->T[0] = d->seg / 256; p->T[1] = p->T[0] * 8; p->T[2] = p->T[1] * 8; p->T[3] = p->T[2] * 2; p->T[4] = p->T[3] * 2;
    p->Cst[0] = 512; p->Cst[1] = 256; p->Cst[2] = 128; p->Cst[3] = 64; p->Cst[4] = 32;
    Bump b;
    const bool vits = gn != nullptr;                 // decoder only: no discriminator tensors in the workspace
    const bool vd = dn != nullptr;                   // VitsDiscriminator only: no generator tensors
    p->gnetp = vits ? gn : &gnet();
    p->dnetp = vits ? nullptr : (vd ? dn : &dnet());
    if (!vd) p->gl = p->gnetp->L;
    if (!vits) p->dl = p->dnetp->L;
    for (auto& l : p->gl) plan_layer_ws(l, b, es, true);
    for (auto& l : p->dl) plan_layer_ws(l, b, es, true);
    plan_layer_grads(p->gl, b, &p->dw_g[0], &p->dw_g[1]);
    plan_layer_grads(p->dl, b, &p->dw_d[0], &p->dw_d[1]);
    const int PG = 32;   // generator pad rows (>= max dilation * (k - 1) / 2 = 25)
    if (!vd) {
        p->xin = mk(b, es, B, p->T[0], p->gnetp->gin, PG, PG);
        if (vits) {
            p->g_dxin = mk(b, es, B, p->T[0], p->gnetp->gin, PG, PG);
            p->cvec = b.take((int64_t)B * 512 * 4); p->dcvec = b.take((int64_t)B * 512 * 4);
            p->zbias = b.take(256); p->dumdb = b.take(256);
        }
        p->h0 = mk(b, es, B, p->T[0], 512, PG, PG);
        for (int i = 0; i < 4; ++i) {
            const int T = p->T[i + 1], C = p->Cst[i + 1];
            p->u[i] = mk(b, es, B, T, C, PG, PG);
            p->ua[i] = mk(b, es, B, T, C, PG, PG);
            for (int j = 0; j < 3; ++j) {
                for (int m = 0; m < 3; ++m) p->xt1[i * 3 + j][m] = mk(b, es, B, T, C, PG, PG);
                for (int m = 0; m < 2; ++m) { p->xr[i * 3 + j][m] = mk(b, es, B, T, C, PG, PG); p->xra[i * 3 + j][m] = mk(b, es, B, T, C, PG, PG); }
            }
            p->xs[i] = mk(b, es, B, T, C, PG, PG);
        }
        p->y = mk(b, es, B, p->T[4], 1, PG, PG);
        {   // backward scratch sized for the largest stage (C * T is constant from stage 2 on)
            int64_t best = 0; int bi = 0;
            for (int i = 0; i < 4; ++i) { int64_t v = (int64_t)p->Cst[i + 1] * (p->T[i + 1] + 2 * PG); if (v > best) { best = v; bi = i; } }
            const int T = p->T[bi + 1], C = p->Cst[bi + 1];
            p->g_dxs = mk(b, es, B, T, C, PG, PG); p->g_da = mk(b, es, B, T, C, PG, PG); p->g_db = mk(b, es, B, T, C, PG, PG);
            p->g_dt1 = mk(b, es, B, T, C, PG, PG); p->g_du = mk(b, es, B, T, C, PG, PG);
            p->g_dy = mk(b, es, B, p->T[4], 1, PG, PG);
            for (int j = 0; j < 3; ++j) {
                p->g_daj[j] = j == 0 ? p->g_da : mk(b, es, B, T, C, PG, PG);
                p->g_dbj[j] = j == 0 ? p->g_db : mk(b, es, B, T, C, PG, PG);
                for (int m = 0; m < 3; ++m) p->g_dt1jm[j][m] = (j == 0 && m == 0) ? p->g_dt1 : mk(b, es, B, T, C, PG, PG);
            }
        }
    }
    // ---- MPD: sequences (b, w); real items first, then fake
    for (int d5 = 0; d5 < NPER && !vits; ++d5) {
        const int pp = PERIODS[d5], ns = 2 * B * pp;
        int H[7];
        H[0] = (d->seg + pp - 1) / pp;
        for (int i = 1; i <= 4; ++i) H[i] = (H[i - 1] + 4 - 5) / 3 + 1;
        H[5] = H[4]; H[6] = H[4];
        // conv3 (k 5, stride 3) and conv4 (k 5, stride 1) run MERGED over all rows of all sequences, pad rows included: the 1024-channel
        // layers are 85 % of a period discriminator's FLOPs and H[4] is only 10 .. 51 rows, so every pad row counts (4 + 4 pad rows: 43 % more
        // rows than valid ones over the five periods; 1 + 1: 11 %).  Two zero rows between consecutive sequences is what a k = 5 convolution
        // needs; the first / last sequence borrow their second row from the guard rows around the tensor.
        const int pf4 = g_hg_wide_pads ? 4 : 1, hp4 = H[4] + (g_hg_wide_pads ? 8 : 2);
        for (int which = 0; which < 2; ++which) {
            SeqSpec* t = which == 0 ? p->pt[d5] : p->pd[d5];
            t[1] = mk(b, es, ns, H[1], 32, 4, 4);
            // t2 is aligned 3 : 1 with t3 (as t3 is with t4) so that conv2 (128 -> 512, k 5, stride 3) is ONE merged GEMM over all rows too: per
            // sequence it has only H[3] = 28 .. 152 output rows, a fifth to a full 128-row tile (measured per-sequence: 56 .. 161 us per launch)
            if (g_hg_wide_pads) t[2] = mk(b, es, ns, H[2], 128, 4, 4);
            else t[2] = mk(b, es, ns, H[2], 128, 9 * pf4, 9 * hp4 - H[2] - 9 * pf4);
            t[3] = mk(b, es, ns, H[3], 512, 3 * pf4, 3 * hp4 - H[3] - 3 * pf4);    // aligned 3:1 with t4 (merged strided conv3)
            t[4] = mk(b, es, ns, H[4], 1024, pf4, hp4 - H[4] - pf4);
            t[5] = mk(b, es, ns, H[5], 1024, pf4, hp4 - H[4] - pf4);
            t[6] = mk(b, es, ns, H[6], 1, pf4, hp4 - H[4] - pf4);
            p->pxc[d5][which] = mk(b, es, ns, H[1], 8, 4, 4);
        }
    }
    // ---- MSD
    p->Tw[0] = d->seg; p->Tw[1] = p->Tw[0] / 2 + 1; p->Tw[2] = p->Tw[1] / 2 + 1;
    if (vd) {   // the one scale discriminator of VitsDiscriminator: real + fake stacked, 7 layers
        int T[8];
        T[0] = d->seg; T[1] = T[0];
        const int str[7] = {1, 4, 4, 4, 4, 1, 1};
        for (int i = 2; i <= 7; ++i) T[i] = (T[i - 1] - 1) / str[i - 1] + 1;
        const int ch[8] = {1, 16, 64, 256, 1024, 1024, 1024, 1};
        for (int which = 0; which < 2; ++which) {
            SeqSpec* t = which == 0 ? p->st[0][0] : p->sd[0][0];
            for (int i = 1; i <= 7; ++i) t[i] = mk(b, es, 2 * B, T[i], ch[i], 24, 24);
            p->sxc[0][0][which] = mk(b, es, 2 * B, T[1], 16, 24, 24);
        }
        for (int sc = 0; sc < 3; ++sc) p->dwav_s[sc] = -1;
    }
    for (int sc = 0; sc < 3 && !vits && !vd; ++sc) {
        int T[9];
        T[0] = p->Tw[sc]; T[1] = T[0];
        const int str[8] = {1, 2, 2, 4, 4, 1, 1, 1};
        for (int i = 2; i <= 8; ++i) T[i] = (i <= 5) ? (T[i - 1] - 1) / str[i - 1] + 1 : T[i - 1];
        const int ch[9] = {1, 128, 128, 256, 512, 1024, 1024, 1024, 1};
        const int sets = sc == 0 ? 2 : 1, ns = sc == 0 ? B : 2 * B;
        for (int set = 0; set < sets; ++set)
            for (int which = 0; which < 2; ++which) {
                SeqSpec* t = which == 0 ? p->st[sc][set] : p->sd[sc][set];
                // t5 .. t8 (conv5 k 41, conv6 k 5, conv_post k 3: stride 1, merged over all rows) share one geometry: 20 zero rows in FRONT of
                // every sequence serve as the back pad of the previous one (the last sequence's are the guard rows): T + 20 rows instead of T + 48
                for (int i = 1; i <= 8; ++i) t[i] = (i >= 5 && !g_hg_wide_pads) ? mk(b, es, ns, T[i], ch[i], 20, 0) : mk(b, es, ns, T[i], ch[i], 24, 24);
                p->sxc[sc][set][which] = mk(b, es, ns, T[1], 16, 24, 24);
            }
        for (int rf = 0; rf < 2; ++rf) p->wav_s[sc][rf] = sc == 0 ? -1 : b.take((int64_t)B * p->Tw[sc] * 4);
        p->dwav_s[sc] = b.take((int64_t)B * p->Tw[sc] * 4);
    }
    p->sn_tmp = (vits || vd) ? -1 : b.take((1024 * 41 * 64 + 1024 + 64) * 4);
    p->losses = b.take(64 * 4);
    p->skws_bytes = (int64_t)96 << 20;      // split-K slabs of the weight-gradient GEMMs (largest: 3 x 1024 x 5120 fp32)
    for (int l = 0; l < 4; ++l) { p->skws[l] = b.take(p->skws_bytes); p->dwav_lane[l] = b.take((int64_t)B * d->seg * 4); }
    p->total = b.cur;
    return XVA_OK;
}

struct Ctx {
    Plan pl;
    char* W;          // workspace
    void* st;
    int compute, dt;
    int lane = 0;     // 0: the caller's stream, 1: the side stream
    Seq S(const SeqSpec& s) const { return seq(s, W, dt); }
    float* F(int64_t off) const { return (float*)(W + off); }
};
// ---- two streams for the discriminators ---------------------------------------------------------------------------------------------
// The five period discriminators and the three scale discriminators are independent networks (models.py:169-200, 234-260).  Issued on
// one stream, every kernel's last, partly filled round of workgroups and every dependent launch's latency is dead time on the other
// CUs; issued on two streams (MPD on the caller's, MSD on a side stream forked and joined with events inside the call), the two
// chains fill each other's gaps.  The side stream and its two events are created once per host thread; nothing else is hidden: the
// call still returns with all work ordered on the caller's stream.  env XVA_HG_STREAMS=n: number of lanes (1 = everything on the caller's
// stream; default 2).
// Side-lane stream priority (round 2 experiment, knob removed): default priority (kept), lowest, highest.  Measured (FastPitch / HiFi-GAN ms
// per step): default 10.38 / 38.4, lowest 10.49 / 45.9 (the lanes starve: HiFi-GAN falls back to its one-stream time), highest 10.47 / 55.0 (the caller's chain starves).
static hipError_t xva_create_lane_stream(hipStream_t* s) {
    static const int mode = 0;      // (measured in round 2: lowest / highest priority lanes starve one side; the knob is gone)
    int least = 0, greatest = 0;
    if (mode != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, mode == 1 ? least : greatest);
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
constexpr int MAXL = 4;                      // lanes: 0 = the caller's stream, 1 .. MAXL-1 side streams
struct SideStreams { hipStream_t s[MAXL] = {}; hipEvent_t fork = nullptr, join[MAXL] = {}, pool[10] = {}; int n = 0; bool init = false; };
static int g_hg_serial = 0;     // xva_hg_set_streams(1): everything on the caller's stream (per-kernel measurements)
static SideStreams& side_streams() {
    static thread_local SideStreams r;
    static thread_local SideStreams serial;   // n == 1
    serial.n = 1; serial.init = true;
    if (!r.init) {
        r.init = true;
        const char* e = getenv("XVA_HG_STREAMS");
        int want = e ? atoi(e) : 4;                 // the discriminators use four lanes (disc_lane), a generator stage's parallel resblocks three
        if (want > MAXL) want = MAXL;
        bool ok = want > 1 && hipEventCreateWithFlags(&r.fork, hipEventDisableTiming) == hipSuccess;
        for (int i = 1; ok && i < want; ++i)
            ok = xva_create_lane_stream(&r.s[i]) == hipSuccess &&
                 hipEventCreateWithFlags(&r.join[i], hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < 10; ++i) ok = hipEventCreateWithFlags(&r.pool[i], hipEventDisableTiming) == hipSuccess;
        r.n = ok ? want : 1;
    }
    return g_hg_serial ? serial : r;
}
// which lane runs discriminator di (MPD 0..4, MSD 0..2), for 2 / 3 / 4 lanes: the two families alternate so that neighbours in time are
// kernels of different shapes; env XVA_HG_LANES="a,b,c,d,e,f,g,h" overrides
static int disc_lane(int di, int nl) {
    static int tab[8] = {-1};
    static bool init = false;
    if (!init) {
        init = true;
        const char* e = getenv("XVA_HG_LANES");
        if (e) { int i = 0; for (const char* q = e; *q && i < 8; ++q) if (*q >= '0' && *q <= '9') tab[i++] = *q - '0'; if (i < 8) tab[0] = -1; }
    }
    if (tab[0] >= 0) return tab[di] % nl;
    // Round 2 (before the merged tails were tightened): 2 lanes (MPD | MSD) 45.8 -> 41.3 ms per iteration, 3 and 4 lanes 41.7 - 42.0.
    // Round 3: the scale discriminators are the longer chain now — full-rate scale 0 (spectral norm: real and fake passes apart) alone on
    // lane 1, the two pooled scales on lane 2, the period discriminators over lanes 0 and 3: 34.3 -> 33.1 ms (sweep of ten assignments:
    // 33.1 - 33.7 for every split that keeps scale 0 alone; 34.1 - 35.0 for 2 / 3 lanes).
    static const int four[8] = {0, 0, 0, 3, 3, 1, 2, 2}, three[8] = {0, 0, 0, 0, 0, 1, 2, 1};
    if (nl >= 4) return four[di];
    if (nl == 3) return three[di];
    return nl >= 2 ? (di >= NPER ? 1 : 0) : 0;
}
struct Ctx;
struct Lanes { Ctx* c[MAXL]; int n; };
static void use_slabs(const Ctx& c);

int make_ctx(Ctx& c, const xva_hg_dims* d, void* ws, int64_t ws_bytes, void* st, const GenNet* gn = nullptr, const DiscNet* dn = nullptr) {
    XVA_TRY(make_plan(d, &c.pl, gn, dn));
    XVA_CHECK_ARG(ws && ((uintptr_t)ws % 256) == 0, "hifigan: workspace null or not 256-byte aligned");
    XVA_CHECK_ARG(ws_bytes >= c.pl.total, "hifigan: workspace too small (%ld < %ld bytes)", (long)ws_bytes, (long)c.pl.total);
    c.W = (char*)ws; c.st = st; c.dt = d->dt; c.compute = d->dt == XVA_BF16 ? 1 : 0;
    use_slabs(c);
    return XVA_OK;
}

static void use_slabs(const Ctx& c) { hg_skws().ptr = c.W + c.pl.skws[c.lane]; hg_skws().bytes = c.pl.skws_bytes; }
// cs[1 .. n-1] = c0 on the side streams (own split-K slabs), ordered after everything issued on c0's stream so far; returns the lane count
static int fork_lanes(const Ctx& c0, Ctx* cs) {
    SideStreams& ss = side_streams();
    cs[0] = c0;
    if (ss.n <= 1) return 1;
    if (hipEventRecord(ss.fork, (hipStream_t)c0.st) != hipSuccess) return 1;
    for (int i = 1; i < ss.n; ++i) {
        if (hipStreamWaitEvent(ss.s[i], ss.fork, 0) != hipSuccess) return 1;
        cs[i] = c0; cs[i].st = ss.s[i]; cs[i].lane = i;
    }
    return ss.n;
}
static int join_lanes(const Ctx* cs, int n) {
    SideStreams& ss = side_streams();
    for (int i = 1; i < n; ++i)
        if (hipEventRecord(ss.join[i], (hipStream_t)cs[i].st) != hipSuccess || hipStreamWaitEvent((hipStream_t)cs[0].st, ss.join[i], 0) != hipSuccess) {
            xva_set_error("hifigan: joining a side stream failed"); return XVA_ERR_HIP;
        }
    use_slabs(cs[0]);
    return XVA_OK;
}

ConvW cw(const Ctx& c, const Layer& l, const float* params, int pass = 0) {
    ConvW w; w.eff = c.W + l.eff[pass]; w.bias = l.bias >= 0 ? params + l.bias : c.F(c.pl.zbias); w.dweff = l.dweff[pass] >= 0 ? c.F(l.dweff[pass]) : nullptr;
    w.Cin = l.Cin; w.Cout = l.Cout; w.k = l.k; w.s = l.s; w.d = l.d; w.P = l.P; w.groups = l.conv_groups();
    return w;
}
ConvTW ctw(const Ctx& c, const Layer& l, const float* params) {
    ConvTW w; w.effF = c.W + l.eff[0]; w.effB = c.W + l.effB; w.bias = params + l.bias; w.dweff = c.F(l.dweff[0]);
    w.Cin = l.Cin; w.Cout = l.Cout; w.k = l.k; w.s = l.s; w.p = l.P;
    return w;
}

const float* eff32(const Ctx& c, const Layer& l, int pass) { return c.dt == XVA_F32 ? (const float*)(c.W + l.eff[pass]) : c.F(l.eff32[pass]); }
// effective weights of weight-norm layers (and fp32 copies for the 1-channel direct kernels)
int prep_wn(const Ctx& c, const std::vector<Layer>& L, const float* params) {
    std::vector<xva_wn_desc> ds;
    ds.reserve(L.size() + 8);
    for (const Layer& l : L) {
        if (l.kind == LK_SN || l.aux) continue;
        xva_wn_desc d;
        memset(&d, 0, sizeof(d));
        d.v = params + l.wv; d.g = l.kind == LK_PLAIN ? nullptr : params + l.wg; d.norm = c.F(l.norm[0]);   // g == null: effective weight = v
        d.eff = c.W + l.eff[0]; d.effB = l.effB >= 0 ? c.W + l.effB : nullptr;
        d.dt = c.dt; d.kind = l.kind == LK_WNT ? 1 : 0; d.D0 = l.D0(); d.D1 = l.D1(); d.k = l.k; d.s = l.s; d.pconv = l.P;
        if (l.dense) { d.kind = 2; d.s = l.dense_cin(); d.pconv = l.Cout / l.groups; }
        ds.push_back(d);
        if (l.eff32[0] >= 0 && c.dt != XVA_F32) {   // fp32 copy for the 1-channel direct kernels
            d.eff = c.W + l.eff32[0]; d.effB = nullptr; d.dt = XVA_F32; d.kind = 0;
            ds.push_back(d);
        }
    }
    XVA_TRY(xva_hg_weight_norm_batch(ds.data(), (int)ds.size(), 0, c.st));
    for (const Layer& l : L) {
        if (l.kind == LK_SN || l.aux) continue;
        if (l.Cin == 1) XVA_TRY(xva_hg_pad_cols(eff32(c, l, 0), c.W + l.wp[0], c.dt, l.Cout, l.k, l.kp(), c.st));
    }
    return XVA_OK;
}

int zero(const Ctx& c, void* p, int64_t bytes) {
    if (hipMemsetAsync(p, 0, bytes, (hipStream_t)c.st) != hipSuccess) { xva_set_error("hifigan: memset failed"); return XVA_ERR_HIP; }
    return XVA_OK;
}

// ================================================================== generator ====
// gvec (B, gcond) fp32: the VITS decoder's speaker vector (null: no conditioning term)
int gen_forward(Ctx& c, const float* P, const float* mel, float* wav_out, const float* gvec = nullptr) {
    const Plan& pl = c.pl; const GenNet& N = *pl.gnetp; const auto& L = pl.gl;
    xva_prof_tag(1000);
    XVA_TRY(prep_wn(c, L, P));
    Seq xin = c.S(pl.xin), h0 = c.S(pl.h0);
    XVA_TRY(xva_hg_mel_to_tm(mel, xin.ptr(), c.dt, pl.B, N.gin, pl.T[0], xin.Hp(), xin.padF, c.st));
    ConvEpi e0;
    XVA_TRY(hg_conv_fwd(xin, h0, cw(c, L[N.pre], P), e0, c.compute, c.st));                         // conv_pre        (models.py:111)
    if (N.cond >= 0 && gvec) {                                                                     // o = o + cond_layer(g)   (xvapitch/hifigan.py:247-248)
        const Layer& cl = L[N.cond];
        xva_gemm_params gp;
        memset(&gp, 0, sizeof(gp));
        gp.A = gvec; gp.B = P + cl.wv; gp.C = c.F(pl.cvec); gp.M = pl.B; gp.N = 512; gp.K = N.gcond; gp.lda = N.gcond; gp.ldb = N.gcond; gp.ldc = 512;
        gp.batch = 1; gp.batch2 = 1; gp.layout = XVA_GEMM_NT; gp.alpha = 1.f; gp.beta = 1.f; gp.bias = P + cl.bias; gp.splitk = 1; gp.compute = 0;
        gp.a_dtype = gp.b_dtype = gp.c_dtype = XVA_F32;
        XVA_TRY(xva_gemm(&gp, c.st));
        XVA_TRY(xva_hg_add_item_vec(h0.ptr(), c.dt, c.F(pl.cvec), pl.B, h0.Hp(), h0.padF, h0.T, 512, c.st));
    }
    Seq prev = h0;
    for (int i = 0; i < 4; ++i) {
        Seq u = c.S(pl.u[i]), ua = c.S(pl.ua[i]), xs = c.S(pl.xs[i]);
        xva_prof_tag(1000 + (i + 1) * 10);
        // lrelu + ups[i] (:115-116).  Every producer on the residual stream also stores the LeakyReLU of its output (u -> ua,
        // xr -> xra): the resblock convolutions read the activated copy instead of re-activating their operand once per tap, and
        // xt1 (only ever consumed through a LeakyReLU) is stored activated.
        const bool dual = c.compute != 0;   // bf16 GEMM epilogues write the activated copy; the exact-fp32 mode makes it with one more pass
        auto act_copy = [&](const Seq& raw, const Seq& actv) {
            return xva_hg_lrelu_copy(raw.ptr(), actv.ptr(), c.dt, raw.rows() * raw.C, SLOPE, c.st);
        };
        XVA_TRY(hg_convT_fwd(prev, u, ctw(c, L[N.ups[i]], P), 1, SLOPE, c.compute, c.st, dual ? &ua : nullptr, SLOPE));
        if (!dual) XVA_TRY(act_copy(u, ua));
        // The three resblocks of a stage (:118-121) are independent chains of six convolutions: one stream lane each (when there are
        // three).  Only the last convolution of a chain touches the shared stage tensor xs (it accumulates the running mean), and it waits
        // for the previous resblock's, so the accumulation order — and every bit of xs — is what one stream produces.
        SideStreams& ss = side_streams();
        const bool lanes = ss.n >= 3;
        auto fail = [&]() { xva_set_error("hifigan: event record / wait failed"); return XVA_ERR_HIP; };
        if (lanes) {
            if (hipEventRecord(ss.fork, (hipStream_t)c.st) != hipSuccess) return fail();
            for (int j = 1; j < 3; ++j) if (hipStreamWaitEvent(ss.s[j], ss.fork, 0) != hipSuccess) return fail();
        }
        for (int j = 0; j < 3; ++j) {
            const int rb = i * 3 + j;
            void* st = (lanes && j > 0) ? (void*)ss.s[j] : c.st;
            Seq xcur = u, xact = ua;
            for (int m = 0; m < 3; ++m) {                                                          // ResBlock1.forward (:41-48)
                Seq xt1 = c.S(pl.xt1[rb][m]);
                ConvEpi e1; e1.act = XVA_ACT_LRELU; e1.act_slope = SLOPE;                           // xt1 = lrelu(c1(lrelu(x)))
                ConvEpi e2; e2.R = &xcur;
                // The 32 / 64-channel stages run a pair as ONE launch with the intermediate in LDS (conv_pair.hip).  XVA_HG_PAIR: 0 = two launches,
                // 1 = fused, operand = the stored activated copy (bit-identical), 2 / 3 = fused from the raw block input (one pass less: LeakyReLU on the
                // operand fragments + residual from the tile / tile activated in place + residual re-read)
                const int pmode = dual ? g_hg_pair : 0;
                auto pair = [&](const Seq& out) {
                    return hg_conv_pair_fwd(pmode >= 2 ? xcur : xact, pmode - 1, SLOPE, xt1, out, cw(c, L[N.rc1[rb][m]], P), cw(c, L[N.rc2[rb][m]], P), SLOPE, e2, c.compute, st);
                };
                if (m < 2) {
                    Seq xn = c.S(pl.xr[rb][m]), xna = c.S(pl.xra[rb][m]);
                    if (dual) { e2.Y2 = &xna; e2.y2_slope = SLOPE; }
                    int fused = pmode ? pair(xn) : 0;
                    if (fused < 0) return fused;
                    if (!fused) {
                        XVA_TRY(hg_conv_fwd(xact, xt1, cw(c, L[N.rc1[rb][m]], P), e1, c.compute, st));
                        XVA_TRY(hg_conv_fwd(xt1, xn, cw(c, L[N.rc2[rb][m]], P), e2, c.compute, st));
                    }
                    if (!dual) XVA_TRY(xva_hg_lrelu_copy(xn.ptr(), xna.ptr(), c.dt, xn.rows() * xn.C, SLOPE, st));
                    xcur = xn; xact = xna;
                } else {                                                                           // xs = sum_j resblock_j / 3  (:118-123)
                    e2.alpha = 1.f / 3; e2.beta = 1.f / 3; e2.accumulate = j > 0;
                    // (the fused pair accumulates into xs in its epilogue: it has to wait for the previous resblock's BEFORE the launch)
                    const bool try_pair = pmode && (xs.C == 32 || xs.C == 64);
                    if (!try_pair) XVA_TRY(hg_conv_fwd(xact, xt1, cw(c, L[N.rc1[rb][m]], P), e1, c.compute, st));
                    if (lanes && j > 0 && hipStreamWaitEvent((hipStream_t)st, ss.pool[j - 1], 0) != hipSuccess) return fail();
                    int fused = try_pair ? pair(xs) : 0;
                    if (fused < 0) return fused;
                    if (!fused) {
                        if (try_pair) XVA_TRY(hg_conv_fwd(xact, xt1, cw(c, L[N.rc1[rb][m]], P), e1, c.compute, st));
                        XVA_TRY(hg_conv_fwd(xt1, xs, cw(c, L[N.rc2[rb][m]], P), e2, c.compute, st));
                    }
                    if (lanes && hipEventRecord(ss.pool[j], (hipStream_t)st) != hipSuccess) return fail();
                }
            }
        }
        if (lanes && hipStreamWaitEvent((hipStream_t)c.st, ss.pool[2], 0) != hipSuccess) return fail();   // join: lane 2 finished after lanes 0 and 1
        prev = xs;
    }
    Seq y = c.S(pl.y);
    ConvEpi ep; ep.a_lrelu = 1; ep.a_slope = 0.01f; ep.act = XVA_ACT_TANH;                            // leaky_relu(0.01), conv_post, tanh (:124-126)
    XVA_TRY(hg_conv_fwd(prev, y, cw(c, L[N.post], P), ep, c.compute, c.st));
    if (wav_out) XVA_TRY(xva_hg_seq1_to_wav(y.ptr(), c.dt, wav_out, pl.B, pl.T[4], y.Hp(), y.padF, c.st));
    return XVA_OK;
}

// weight-norm backward of layers li[0..n) of L (li == nullptr: all of L)
int wn_backward(Ctx& c, const std::vector<Layer>& L, const float* P, float* G, const int* li = nullptr, int n = 0) {
    std::vector<xva_wn_desc> ds;
    ds.reserve(L.size());
    const int cnt = li ? n : (int)L.size();
    for (int q = 0; q < cnt; ++q) {
        const Layer& l = L[li ? li[q] : q];
        if (l.kind == LK_SN || l.aux) continue;
        xva_wn_desc d;
        memset(&d, 0, sizeof(d));
        const bool plain = l.kind == LK_PLAIN;           // dv += dW re-laid out; no g
        d.dW = c.F(l.dweff[0]); d.v = P + l.wv; d.g = plain ? nullptr : P + l.wg; d.norm = c.F(l.norm[0]); d.dv = G + l.wv; d.dg = plain ? nullptr : G + l.wg;
        d.kind = l.kind == LK_WNT ? 1 : 0; d.D0 = l.D0(); d.D1 = l.D1(); d.k = l.k;
        if (l.dense) { d.kind = 2; d.s = l.dense_cin(); d.pconv = l.Cout / l.groups; }
        ds.push_back(d);
    }
    if (ds.empty()) return XVA_OK;
    return xva_hg_weight_norm_batch(ds.data(), (int)ds.size(), 1, c.st);
}
// Bucket i is final on the lane this context issues to: record its event and, when the data-parallel host has registered a callback, call it NOW — while
// the host is still issuing the backward pass — so that the bucket's wait + all-reduce are enqueued at once (a wait issued after the whole pass had been
// issued resolved when the recording lane had drained: tools/dp_overlap_probe.py, csrc/fastpitch_engine.hip Ctx::record).
typedef void (*xva_bucket_cb_t)(int bucket, void* user);
static thread_local xva_bucket_cb_t g_hg_bucket_cb = nullptr;
static thread_local void* g_hg_bucket_user = nullptr;
int record(const Ctx& c, void* const* events, int i) {
    if (!events || !events[i]) return XVA_OK;
    if (hipEventRecord((hipEvent_t)events[i], (hipStream_t)c.st) != hipSuccess) { xva_set_error("hifigan: hipEventRecord failed"); return XVA_ERR_HIP; }
    if (g_hg_bucket_cb) g_hg_bucket_cb(i, g_hg_bucket_user);
    return XVA_OK;
}
int zero_dweff(Ctx& c, const std::vector<Layer>& L) {
    const int64_t* r = (&L == &c.pl.gl) ? c.pl.dw_g : c.pl.dw_d;
    return zero(c, c.W + r[0], r[1] - r[0]);
}
// gradients of the padded first-layer weights -> the layer's dweff ([Cout][k], tap-major with Cin = 1)
int fold_dwp(Ctx& c, const std::vector<Layer>& L, const int* li, int n) {
    for (int q = 0; q < n; ++q) {
        const Layer& l = L[li[q]];
        for (int p = 0; p < 2; ++p)
            if (l.dwp[p] >= 0 && l.dweff[p] >= 0) XVA_TRY(xva_hg_unpad_cols_add(c.F(l.dwp[p]), c.F(l.dweff[p]), l.Cout, l.k, l.kp(), c.st));
    }
    return XVA_OK;
}

// view of a scratch spec with another stage's geometry (scratch buffers are sized for the largest stage)
Seq as_stage(const Ctx& c, const SeqSpec& s, int T, int C) {
    SeqSpec v = s; v.T = T; v.C = C;
    int64_t shift = (int64_t)GUARD * (s.C - C) * c.pl.es;   // keep GUARD rows of the NEW width in front
    v.off = s.off - shift;
    return c.S(v);
}

// Gradient buckets of the generator in backward-completion order (each a contiguous range of the flat buffer):
// 0..3 = the resblocks of stages 3..0, 4 = conv_pre + ups.0-3, 5 = conv_post.
constexpr int G_BUCKETS = 6;
// VITS decoder: gvec as in gen_forward; d_in (B, gin, T0) fp32 (may be null) receives the gradient w.r.t. the input features
int gen_backward(Ctx& c, const float* P, float* G, const float* d_wav, void* const* events, const float* gvec = nullptr, float* d_in = nullptr) {
    const Plan& pl = c.pl; const GenNet& N = *pl.gnetp; const auto& L = pl.gl;
    xva_prof_tag(5000);
    XVA_TRY(zero_dweff(c, L));
    Seq y = c.S(pl.y), dy = c.S(pl.g_dy);
    XVA_TRY(xva_hg_tanh_bwd(d_wav, y.ptr(), dy.ptr(), c.dt, pl.B, pl.T[4], y.Hp(), y.padF, c.st));
    // conv_post backward (single output channel: direct kernels)
    {
        const Layer& l = L[N.post];
        Seq xs = c.S(pl.xs[3]);
        Seq dxs = as_stage(c, pl.g_dxs, pl.T[4], pl.Cst[4]);
        XVA_TRY(xva_hg_cout1_bwd_weight(dy.ptr(), xs.ptr(), c.F(l.dweff[0]), l.bias >= 0 ? G + l.bias : c.F(pl.dumdb), c.dt, xs.rows(), xs.C, l.k, 1, l.P, 1,
                                        0.01f, c.st));
        XVA_TRY(xva_hg_cout1_bwd_data(dy.ptr(), eff32(c, l, 0), xs.ptr(), dxs.ptr(), c.dt, xs.rows(), xs.C, l.k, 1, l.P, xs.Hp(), xs.padF, xs.T, 1,
                                      0.01f, c.st));
    }
    // Two lanes (see "two streams for the discriminators"): the data-gradient chain on the caller's stream, the weight gradients and bias
    // sums of each convolution pair on a side stream behind an event recorded once the pair's d(conv1 output) exists.  The tensors the
    // side lane reads (d of each block's output, d of each conv1 output) have their own buffer per (resblock, block) within a stage;
    // the stage's buffers are reused by the next stage, so the chain joins the side lane before a stage's ups backward.
    SideStreams& ss = side_streams();
    const bool two = ss.n > 1;
    Ctx cw_ = c;
    if (two) { cw_.st = ss.s[1]; cw_.lane = 1; }
    for (int i = 3; i >= 0; --i) {
        const int T = pl.T[i + 1], C = pl.Cst[i + 1];
        xva_prof_tag(5000 + (i + 1) * 10);
        Seq dxs = as_stage(c, pl.g_dxs, T, C), du = as_stage(c, pl.g_du, T, C);
        Seq ua = c.S(pl.ua[i]);
        for (int j = 0; j < 3; ++j) {
            const int rb = i * 3 + j;
            const int jj = two ? j : 0;
            Seq da = as_stage(c, pl.g_daj[jj], T, C), db = as_stage(c, pl.g_dbj[jj], T, C);
            // d(pair output) for m = 2 is dxs / 3; the 1/3 is folded into the first GEMMs' alpha / beta
            Seq dcur = dxs; float sc = 1.f / 3;
            for (int m = 2; m >= 0; --m) {
                // the ACTIVATED conv inputs: they are the weight-gradient operands, and (LeakyReLU keeps the sign) the gates
                Seq xin = m == 0 ? ua : c.S(pl.xra[rb][m - 1]);
                Seq xt1 = c.S(pl.xt1[rb][m]);
                Seq dt1 = as_stage(c, pl.g_dt1jm[jj][two ? m : 0], T, C);
                ConvW w2 = cw(c, L[N.rc2[rb][m]], P), w1 = cw(c, L[N.rc1[rb][m]], P);
                BwdEpi b2; b2.gate = &xt1; b2.gate_slope = SLOPE; b2.alpha = sc;
                XVA_TRY(hg_conv_bwd_data(dcur, dt1, w2, b2, c.compute, c.st));                    // dt1 = d(conv1 output)
                if (two) {
                    hipEvent_t ev = ss.pool[j * 3 + m];
                    if (hipEventRecord(ev, (hipStream_t)c.st) != hipSuccess || hipStreamWaitEvent(ss.s[1], ev, 0) != hipSuccess) {
                        xva_set_error("hifigan: event record / wait failed"); return XVA_ERR_HIP;
                    }
                }
                use_slabs(cw_);
                // (weight and bias gradient of a convolution in one launch where the resident-operand kernel takes it: hg_conv.h)
                XVA_TRY(hg_conv_bwd_weight(dcur, xt1, w2, 0, 0.f, sc, c.compute, cw_.st, G + L[N.rc2[rb][m]].bias));
                XVA_TRY(hg_conv_bwd_weight(dt1, xin, w1, 0, 0.f, 1.f, c.compute, cw_.st, G + L[N.rc1[rb][m]].bias));
                use_slabs(c);
                BwdEpi b1; b1.gate = &xin; b1.gate_slope = SLOPE; b1.R = &dcur; b1.beta = sc;
                Seq dst = (m == 0) ? du : ((m == 2) ? da : db);
                if (m == 0) b1.accumulate = 0;
                // m == 0 writes the resblock's contribution to d(u): first resblock overwrites, the others accumulate
                if (m == 0 && j > 0) {
                    // residual (beta * dcur) and conv term both accumulate into du
                    b1.accumulate = 1;
                }
                XVA_TRY(hg_conv_bwd_data(dt1, dst, w1, b1, c.compute, c.st));
                dcur = dst; sc = 1.f;
            }
        }
        if (two) {   // join: the next kernels overwrite what the side lane reads (d xs), and the stage's weight gradients must be final
            if (hipEventRecord(ss.pool[9], ss.s[1]) != hipSuccess || hipStreamWaitEvent((hipStream_t)c.st, ss.pool[9], 0) != hipSuccess) {
                xva_set_error("hifigan: joining the weight-gradient lane failed"); return XVA_ERR_HIP;
            }
        }
        // ups[i] backward: d(prev) = lrelu'(prev) * strided-conv(du) ; dW, db
        Seq prev = i == 0 ? c.S(pl.h0) : c.S(pl.xs[i - 1]);
        Seq dprev = as_stage(c, pl.g_dxs, pl.T[i], pl.Cst[i]);
        ConvTW wt = ctw(c, L[N.ups[i]], P);
        XVA_TRY(hg_convT_bwd_weight(du, prev, wt, 1, SLOPE, c.compute, c.st));
        XVA_TRY(xva_hg_colsum(du.ptr(), c.dt, G + L[N.ups[i]].bias, du.rows(), C, 1.f, c.st));
        XVA_TRY(zero(c, dprev.ptr(), dprev.rows() * dprev.C * dprev.es()));     // per-item GEMM writes valid rows only: pads must be zero
        XVA_TRY(hg_convT_bwd_data(du, dprev, wt, &prev, SLOPE, c.compute, c.st));
        {   // this stage's resblock gradients are final: reparametrisation backward, then the bucket's event
            int li[18], n = 0;
            for (int j = 0; j < 3; ++j) for (int m = 0; m < 3; ++m) { li[n++] = N.rc1[i * 3 + j][m]; li[n++] = N.rc2[i * 3 + j][m]; }
            XVA_TRY(wn_backward(c, L, P, G, li, n));
            XVA_TRY(record(c, events, 3 - i));
        }
    }
    {   // conv_pre backward (weights; the VITS decoder also wants d(input) and the cond_layer gradients)
        Seq dh0 = as_stage(c, pl.g_dxs, pl.T[0], 512), xin = c.S(pl.xin);
        XVA_TRY(hg_conv_bwd_weight(dh0, xin, cw(c, L[N.pre], P), 0, 0.f, 1.f, c.compute, c.st));
        XVA_TRY(xva_hg_colsum(dh0.ptr(), c.dt, G + L[N.pre].bias, dh0.rows(), 512, 1.f, c.st));
        if (N.cond >= 0 && gvec) {   // d cvec[b] = sum_t d h0[b][t] (pad rows are zero) ; dW_c += d cvec^T g ; db_c += sum_b d cvec
            const Layer& cl = L[N.cond];
            float* dcv = c.F(pl.dcvec);
            XVA_TRY(zero(c, dcv, (int64_t)pl.B * 512 * 4));
            const int64_t item = (int64_t)dh0.Hp() * 512 * dh0.es();
            for (int b = 0; b < pl.B; ++b) XVA_TRY(xva_hg_colsum((const char*)dh0.ptr() + b * item, c.dt, dcv + (int64_t)b * 512, dh0.Hp(), 512, 1.f, c.st));
            xva_gemm_params gp;
            memset(&gp, 0, sizeof(gp));
            gp.A = dcv; gp.B = gvec; gp.C = G + cl.wv; gp.M = 512; gp.N = N.gcond; gp.K = pl.B; gp.lda = 512; gp.ldb = N.gcond; gp.ldc = N.gcond;
            gp.batch = 1; gp.batch2 = 1; gp.layout = XVA_GEMM_TN; gp.alpha = 1.f; gp.beta = 1.f; gp.accumulate = 1; gp.splitk = 1; gp.compute = 0;
            gp.a_dtype = gp.b_dtype = gp.c_dtype = XVA_F32;
            XVA_TRY(xva_gemm(&gp, c.st));
            XVA_TRY(xva_hg_colsum(dcv, XVA_F32, G + cl.bias, pl.B, 512, 1.f, c.st));
        }
        if (d_in) {
            Seq dxin = c.S(pl.g_dxin);
            BwdEpi b0;
            XVA_TRY(hg_conv_bwd_data(dh0, dxin, cw(c, L[N.pre], P), b0, c.compute, c.st));
            XVA_TRY(xva_seq_to_bct(dxin.ptr(), d_in, c.dt == XVA_BF16 ? 1 : 0, pl.B, N.gin, pl.T[0], dxin.padF, 0, c.st));
        }
    }
    const int rest[6] = {N.pre, N.ups[0], N.ups[1], N.ups[2], N.ups[3], N.post};
    XVA_TRY(wn_backward(c, L, P, G, rest, 6));
    XVA_TRY(record(c, events, 4));
    return record(c, events, 5);
}

// ================================================================== discriminators ====
// One discriminator = conv0 (1 input channel, direct kernel) + GEMM layers + conv_post (1 output channel).
struct DiscRun {
    const int* li; int n;            // layer indices into pl.dl (n layers incl. conv_post): MPD 6, MSD 8
    const SeqSpec* t; const SeqSpec* d;   // activations / gradients t[1..n]
    const SeqSpec* xc;                    // xc[0] im2col of the waveform, xc[1] its gradient
    int p;                           // period (MSD: 1)
    int Tw;                          // input waveform length
    int pass;                        // effective-weight set (spectral norm: 0 real pass, 1 fake pass)
};


// conv0 (1 input channel) in GEMM form: W padded to kp taps, input = im2col of the waveform
ConvW cw0(const Ctx& c, const Layer& l, const float* P, int pass) {
    ConvW w; w.eff = c.W + l.wp[pass]; w.bias = P + l.bias; w.dweff = c.F(l.dwp[pass]);
    w.Cin = l.kp(); w.Cout = l.Cout; w.k = 1; w.s = 1; w.d = 1; w.P = 0; w.groups = 1;
    return w;
}
int conv0_im2col(Ctx& c, const DiscRun& r, const float* wav, int i0, int ni) {
    const Layer& l0 = c.pl.dl[r.li[0]];
    Seq xc = c.S(r.xc[0]).slice(i0, ni);
    return xva_hg_im2col1(wav, xc.ptr(), c.dt, ni / r.p, r.Tw, r.p, l0.k, l0.s, l0.P, l0.kp(), xc.Hp(), xc.padF, c.st);
}
int conv0_fwd(Ctx& c, const float* Pd, const DiscRun& r, int i0, int ni) {
    const Layer& l0 = c.pl.dl[r.li[0]];
    Seq xc = c.S(r.xc[0]).slice(i0, ni), t1 = c.S(r.t[1]).slice(i0, ni);
    ConvEpi e; e.act = XVA_ACT_LRELU; e.act_slope = SLOPE;
    return hg_conv_fwd(xc, t1, cw0(c, l0, Pd, r.pass), e, c.compute, c.st);
}

// conv0 straight from the waveform (hg_ops.hip hg_cin1_fwd_mfma_kernel): no im2col launch, no K = 8 / 16 product.  XVA_HG_CONV0_DIRECT=0 keeps the GEMM form
// (the im2col of the waveform is then made here; with the direct form the D-step backward makes it, for the weight gradient — once per iteration, not twice)
static const int g_conv0_direct = [] { const char* e = getenv("XVA_HG_CONV0_DIRECT"); return e ? atoi(e) : 1; }();
// bf16 mode only (hg_cin1_fwd_mfma_kernel); the exact-fp32 parity mode keeps the GEMM form, and so does anything the matrix-pipe kernel does not take.  (The
// packed-fp32 direct kernel of round 5, found giving sporadically wrong values on the side stream lanes, was deleted in round 6 without a root cause:
// tests/test_lanes_gpu.py compares every engine's lanes-on results with its one-stream results bit for bit.)
static bool conv0_direct(const Ctx& c) { return g_conv0_direct && c.dt == XVA_BF16; }
int conv0_any(Ctx& c, const float* Pd, const DiscRun& r, const float* wav, int i0, int ni) {
    if (!conv0_direct(c)) { XVA_TRY(conv0_im2col(c, r, wav, i0, ni)); return conv0_fwd(c, Pd, r, i0, ni); }
    const Layer& l0 = c.pl.dl[r.li[0]];
    Seq t1 = c.S(r.t[1]).slice(i0, ni);
    return xva_hg_cin1_fwd(wav, eff32(c, l0, r.pass), Pd + l0.bias, t1.ptr(), c.dt, ni / r.p, r.Tw, r.p, l0.k, l0.s, l0.P, l0.Cout, t1.Hp(), t1.padF, SLOPE, c.st);
}

int sn_prepare(Ctx& c, float* Pd, const DiscRun& r) {
    // one power iteration for every spectral-norm layer of the discriminator: 5 launches for all 8 layers (hg_wn.h: xva_sn_desc)
    xva_sn_desc ds[XVA_SN_BATCH];
    int n = 0;
    int64_t toff = 0;
    for (int i = 0; i < r.n; ++i) {
        const Layer& l = c.pl.dl[r.li[i]];
        if (l.kind != LK_SN) continue;
        xva_sn_desc d;
        memset(&d, 0, sizeof(d));
        d.W = Pd + l.wv; d.u = Pd + l.bu; d.v = Pd + l.bv;
        d.su = c.F(l.su[r.pass]); d.sv = c.F(l.sv[r.pass]);
        d.eff = c.W + l.eff[r.pass]; d.eff2 = (l.eff32[r.pass] >= 0 && c.dt != XVA_F32) ? c.W + l.eff32[r.pass] : nullptr;
        d.sigma = c.F(l.norm[r.pass]); d.tmp = c.F(c.pl.sn_tmp) + toff;
        d.dt = c.dt; d.D0 = l.D0(); d.D1 = l.D1(); d.k = l.k;
        toff += ((int64_t)l.D1() * l.k + l.D0() + 7) & ~(int64_t)3;
        XVA_CHECK_ARG(n < XVA_SN_BATCH, "sn_prepare: more than %d spectral-norm layers", XVA_SN_BATCH);
        ds[n++] = d;
    }
    if (n == 0) return XVA_OK;
    XVA_CHECK_ARG(toff * 4 <= (1024 * 41 * 64 + 1024 + 64) * 4, "sn_prepare: scratch too small");
    XVA_TRY(xva_hg_spectral_norm_fwd_batch(ds, n, c.st));
    for (int i = 0; i < r.n; ++i) {
        const Layer& l = c.pl.dl[r.li[i]];
        if (l.kind == LK_SN && l.Cin == 1) XVA_TRY(xva_hg_pad_cols(eff32(c, l, r.pass), c.W + l.wp[r.pass], c.dt, l.Cout, l.k, l.kp(), c.st));
    }
    return XVA_OK;
}

// forward of sequences [i0, i0 + ni) (ni = nb * p) fed from waveform `wav` (nb items)
int disc_forward(Ctx& c, const float* Pd, const DiscRun& r, const float* wav, int i0, int ni) {
    const auto& L = c.pl.dl;
    XVA_TRY(conv0_any(c, Pd, r, wav, i0, ni));
    for (int i = 1; i < r.n; ++i) {
        Seq x = c.S(r.t[i]).slice(i0, ni), y = c.S(r.t[i + 1]).slice(i0, ni);
        ConvEpi e;
        if (i < r.n - 1) { e.act = XVA_ACT_LRELU; e.act_slope = SLOPE; }
        XVA_TRY(hg_conv_fwd(x, y, cw(c, L[r.li[i]], Pd, r.pass), e, c.compute, c.st));
    }
    return XVA_OK;
}

// D-step backward over sequences [i0, i0+ni): d[n] (score gradient) must be seeded.  Accumulates dweff / bias grads.
// wav_a / wav_b: the waveforms feeding the first / second half of the slice (nb_a + nb_b items); wav_b may be null.
int disc_backward_params(Ctx& c, const float* Pd, float* Gd, const DiscRun& r, int i0, int ni, const float* wav_a, int nb_a, const float* wav_b,
                         std::vector<xva_cs_desc>& colsums) {
    auto defer_colsum = [&](const Seq& t, float* out) {   // the gradient tensors persist: bias gradients are summed in one batched launch later
        xva_cs_desc d; memset(&d, 0, sizeof(d));
        d.X = t.ptr(); d.out = out; d.rows = t.rows(); d.scale = 1.f; d.dt = c.dt; d.C = t.C;
        colsums.push_back(d);
    };
    const auto& L = c.pl.dl;
    for (int i = r.n - 1; i >= 1; --i) {
        const Layer& l = L[r.li[i]];
        Seq x = c.S(r.t[i]).slice(i0, ni), dy = c.S(r.d[i + 1]).slice(i0, ni), dx = c.S(r.d[i]).slice(i0, ni);
        if (i == r.n - 1) {       // conv_post: single output channel
            XVA_TRY(xva_hg_cout1_bwd_weight(dy.ptr(), x.ptr(), c.F(l.dweff[r.pass]), Gd + l.bias, c.dt, x.rows(), x.C, l.k, 1, l.P, 0, 0.f, c.st));
            XVA_TRY(xva_hg_cout1_bwd_data(dy.ptr(), eff32(c, l, r.pass), x.ptr(), dx.ptr(), c.dt, x.rows(), x.C, l.k, 1, l.P, x.Hp(), x.padF, x.T, 1, SLOPE,
                                          c.st));
        } else {
            ConvW w = cw(c, l, Pd, r.pass);
            bool later = false;          // (the grouped scale-discriminator layers: the bias gradient comes out of the weight-gradient launch)
            XVA_TRY(hg_conv_bwd_weight(dy, x, w, 0, 0.f, 1.f, c.compute, c.st, Gd + l.bias, &later));
            if (later) defer_colsum(dy, Gd + l.bias);
            BwdEpi b; b.gate = &x; b.gate_slope = SLOPE;
            XVA_TRY(hg_conv_bwd_data(dy, dx, w, b, c.compute, c.st));
        }
    }
    const Layer& l0 = L[r.li[0]];
    Seq d1 = c.S(r.d[1]).slice(i0, ni), xc = c.S(r.xc[0]).slice(i0, ni);
    if (conv0_direct(c)) {   // the weight gradient's operand: the im2col of the waveform(s) feeding this slice (the GEMM form's forward leaves it in xc)
        XVA_TRY(conv0_im2col(c, r, wav_a, i0, nb_a * r.p));
        if (wav_b) XVA_TRY(conv0_im2col(c, r, wav_b, i0 + nb_a * r.p, ni - nb_a * r.p));
    }
    XVA_TRY(hg_conv_bwd_weight(d1, xc, cw0(c, l0, Pd, r.pass), 0, 0.f, 1.f, c.compute, c.st));
    defer_colsum(d1, Gd + l0.bias);
    return XVA_OK;
}

// G-step backward of the FAKE sequences [f0, f0+nf) against the REAL sequences [r0, r0+nf) (feature matching + LSGAN):
// data gradients only, down to d(wave).  rt = tensors holding the real fmaps (may be another set for spectral norm).
// feat: weight of the feature-matching term in the gradient (1; 0 for xVAPitch, whose feature loss detaches the GENERATED features —
// python/xvapitch/model.py:345-347 hands (fake, real) to feature_loss(feats_real, feats_generated), python/xvapitch/losses.py:64-72)
int disc_backward_wave(Ctx& c, const float* Pd, const DiscRun& r, const SeqSpec* rt, int r0, int f0, int nf, float* dwav, int accumulate, float feat = 1.f) {
    const auto& L = c.pl.dl;
    auto numel = [&](int i) { const SeqSpec& s = r.t[i]; return (float)((int64_t)nf * s.T * s.C); };
    {   // score: LSGAN generator loss mean((1 - g)^2) + feature term of the last fmap
        Seq g = c.S(r.t[r.n]).slice(f0, nf), rr = c.S(rt[r.n]).slice(r0, nf), d = c.S(r.d[r.n]).slice(f0, nf);
        XVA_TRY(xva_hg_seed_grad(rr.ptr(), g.ptr(), d.ptr(), c.dt, nf, g.Hp(), g.padF, g.T, g.C, feat * 2.f / numel(r.n), 1.f / numel(r.n), 1, 0, 0.f, 1, c.st));
    }
    for (int i = r.n - 1; i >= 1; --i) {
        const Layer& l = L[r.li[i]];
        Seq x = c.S(r.t[i]).slice(f0, nf), xr = c.S(rt[i]).slice(r0, nf), dy = c.S(r.d[i + 1]).slice(f0, nf), dx = c.S(r.d[i]).slice(f0, nf);
        if (i == r.n - 1) {
            XVA_TRY(xva_hg_cout1_bwd_data(dy.ptr(), eff32(c, l, r.pass), x.ptr(), dx.ptr(), c.dt, x.rows(), x.C, l.k, 1, l.P, x.Hp(), x.padF, x.T, 0, 0.f, c.st));
            XVA_TRY(xva_hg_seed_grad(xr.ptr(), x.ptr(), dx.ptr(), c.dt, nf, x.Hp(), x.padF, x.T, x.C, feat * 2.f / numel(i), 0.f, 0, 1, SLOPE, 0, c.st));
        } else {
            // + feature-matching gradient of this fmap, then LeakyReLU backward on the total — both in the product's epilogue (round 4; a separate
            // read-modify-write pass over dx before: 46 launches and 3.9 GB of traffic per iteration at B = 64)
            BwdEpi b; b.gate = &x; b.gate_slope = SLOPE;
            if (feat != 0.f) { b.fm = &xr; b.fm_c = feat * 2.f / numel(i); }
            XVA_TRY(hg_conv_bwd_data(dy, dx, cw(c, l, Pd, r.pass), b, c.compute, c.st));
        }
    }
    const Layer& l0 = L[r.li[0]];
    Seq d1 = c.S(r.d[1]).slice(f0, nf), dxc = c.S(r.xc[1]).slice(f0, nf);
    BwdEpi b0;
    XVA_TRY(hg_conv_bwd_data(d1, dxc, cw0(c, l0, Pd, r.pass), b0, c.compute, c.st));
    return xva_hg_col2im1(dxc.ptr(), c.dt, dwav, nf / r.p, r.Tw, r.p, l0.k, l0.s, l0.P, l0.kp(), dxc.Hp(), dxc.padF, accumulate, c.st);
}

// loss sums: out[0] += mean((1-r)^2) + mean(g^2) (discriminator loss), out[1] += mean((1-g)^2) (generator loss),
// out[2] += 2 * sum_l mean|r_l - g_l| (feature loss)
// (the reductions of all discriminators are collected and issued as one batched launch by the caller)
// loss_mask: bit 0 = the discriminator loss (D step), bit 1 = generator + feature-matching losses (G step: these read every fmap)
void disc_losses(Ctx& c, const DiscRun& r, const SeqSpec* rt, int r0, int f0, int nf, float* out, std::vector<xva_red_desc>& reds, int loss_mask) {
    auto add = [&](const Seq& a, const Seq* b, const Seq& geo, int mode, float scale, float* dst) {
        xva_red_desc d;
        memset(&d, 0, sizeof(d));
        d.a = a.ptr(); d.b = b ? b->ptr() : nullptr; d.out = dst; d.scale = scale;
        d.dt = c.dt; d.nseq = nf; d.Hp = geo.Hp(); d.padF = geo.padF; d.T = geo.T; d.C = geo.C; d.mode = mode;
        reds.push_back(d);
    };
    for (int i = 1; i <= r.n; ++i) {
        Seq g = c.S(r.t[i]).slice(f0, nf), rr = c.S(rt[i]).slice(r0, nf);
        const float inv = 1.f / (float)((int64_t)nf * g.T * g.C);
        if (loss_mask & 2) add(rr, &g, g, 0, 2.f * inv, out + 2);
        if (i == r.n) {
            if (loss_mask & 1) { add(rr, nullptr, g, 1, inv, out + 0); add(g, nullptr, g, 2, inv, out + 0); }
            if (loss_mask & 2) add(g, nullptr, g, 1, inv, out + 1);
        }
    }
}

struct DiscSet { DiscRun run; const SeqSpec* rt; int r0, f0, nf; bool sn; const float* wr; const float* wg; int nb; };

// enumerate the 8 discriminators: (MPD x 5, then MSD x 3).  For spectral-norm scale 0 the real / fake passes are separate sets.
void build_sets(Ctx& c, const float* yr, const float* yg, std::vector<DiscSet>& out, std::vector<DiscRun>& sn_real) {
    const Plan& pl = c.pl; const DiscNet& N = *pl.dnetp;
    for (int d5 = 0; d5 < NPER; ++d5) {
        DiscSet s; s.run.li = N.mpd[d5]; s.run.n = 6; s.run.t = pl.pt[d5]; s.run.d = pl.pd[d5]; s.run.xc = pl.pxc[d5]; s.run.p = PERIODS[d5]; s.run.Tw = pl.seg; s.run.pass = 0;
        s.rt = pl.pt[d5]; s.nf = pl.B * PERIODS[d5]; s.r0 = 0; s.f0 = s.nf; s.sn = false; s.wr = yr; s.wg = yg; s.nb = pl.B;
        out.push_back(s);
    }
    if (N.vits) {   // one full-rate scale discriminator, weight norm, real + fake stacked
        DiscSet s; s.run.li = N.msd[0]; s.run.n = 7; s.run.p = 1; s.run.Tw = pl.seg; s.nf = pl.B; s.wr = yr; s.wg = yg; s.nb = pl.B;
        s.run.t = pl.st[0][0]; s.run.d = pl.sd[0][0]; s.run.xc = pl.sxc[0][0]; s.run.pass = 0; s.rt = pl.st[0][0]; s.r0 = 0; s.f0 = pl.B; s.sn = false;
        out.push_back(s);
        return;
    }
    for (int sc = 0; sc < 3; ++sc) {
        const float* wr = sc == 0 ? yr : c.F(pl.wav_s[sc][0]);
        const float* wg = sc == 0 ? yg : c.F(pl.wav_s[sc][1]);
        DiscSet s; s.run.li = N.msd[sc]; s.run.n = 8; s.run.p = 1; s.run.Tw = pl.Tw[sc]; s.nf = pl.B; s.wr = wr; s.wg = wg; s.nb = pl.B;
        if (sc == 0) {
            s.run.t = pl.st[0][1]; s.run.d = pl.sd[0][1]; s.run.xc = pl.sxc[0][1]; s.run.pass = 1; s.rt = pl.st[0][0]; s.r0 = 0; s.f0 = 0; s.sn = true;
            DiscRun rr = s.run; rr.t = pl.st[0][0]; rr.d = pl.sd[0][0]; rr.xc = pl.sxc[0][0]; rr.pass = 0; sn_real.push_back(rr);
        } else {
            s.run.t = pl.st[sc][0]; s.run.d = pl.sd[sc][0]; s.run.xc = pl.sxc[sc][0]; s.run.pass = 0; s.rt = pl.st[sc][0]; s.r0 = 0; s.f0 = pl.B; s.sn = false;
        }
        out.push_back(s);
    }
}

int pool_waves(Ctx& c, const float* yr, const float* yg) {
    const Plan& pl = c.pl;
    XVA_TRY(xva_hg_avgpool_fwd(yr, c.F(pl.wav_s[1][0]), pl.B, pl.Tw[0], c.st));
    XVA_TRY(xva_hg_avgpool_fwd(yg, c.F(pl.wav_s[1][1]), pl.B, pl.Tw[0], c.st));
    XVA_TRY(xva_hg_avgpool_fwd(c.F(pl.wav_s[1][0]), c.F(pl.wav_s[2][0]), pl.B, pl.Tw[1], c.st));
    XVA_TRY(xva_hg_avgpool_fwd(c.F(pl.wav_s[1][1]), c.F(pl.wav_s[2][1]), pl.B, pl.Tw[1], c.st));
    return XVA_OK;
}

// forward of all 8 discriminators on (real, fake); losses[0..2] = {disc loss, gen loss, feature loss}
int discs_forward(Ctx& c0, float* Pd, const float* yr, const float* yg, float* losses, int loss_mask) {
    // bit 2 of loss_mask: the caller vouches that the effective weights in this workspace belong to these parameters (the D-step forward of iteration
    // i + 1 runs on the parameters the G-step forward of iteration i prepared: 70 M parameters re-read and 140 MB re-written for nothing otherwise)
    if (!(loss_mask & 4)) XVA_TRY(prep_wn(c0, c0.pl.dl, Pd));
    if (!c0.pl.dnetp->vits) XVA_TRY(pool_waves(c0, yr, yg));
    std::vector<DiscSet> sets; std::vector<DiscRun> snr;
    std::vector<xva_red_desc> reds[MAXL];      // per lane: a lane's loss reductions run on that lane as soon as its discriminators are through,
                                               // under the other lanes' convolutions (one batched launch after the join was 0.3 ms of serial HBM-bound time per forward)
    build_sets(c0, yr, yg, sets, snr);
    if (losses) XVA_TRY(zero(c0, losses, 4 * sizeof(float)));
    Ctx cs[MAXL]; const int nl = fork_lanes(c0, cs);
    int di = 0;
    for (auto& s : sets) {
        const int ln = disc_lane(di, nl);
        Ctx& c = cs[ln];
        xva_prof_tag(2000 + di * 10);
        ++di;
        if (!((disc_mask() >> (di - 1)) & 1)) continue;
        if (s.sn) {   // models.py:244-253: d(y) then d(y_hat), one power iteration each
            DiscRun rr = snr[0];
            XVA_TRY(sn_prepare(c, Pd, rr));
            XVA_TRY(disc_forward(c, Pd, rr, s.wr, 0, s.nb));
            XVA_TRY(sn_prepare(c, Pd, s.run));
            XVA_TRY(disc_forward(c, Pd, s.run, s.wg, 0, s.nb));
        } else {
            // im2col per half (different waveforms), then every layer jointly over real + fake
            const auto& L = c.pl.dl;
            if (conv0_direct(c)) {
                XVA_TRY(conv0_any(c, Pd, s.run, s.wr, 0, s.nf));
                XVA_TRY(conv0_any(c, Pd, s.run, s.wg, s.nf, s.nf));
            } else {
                XVA_TRY(conv0_im2col(c, s.run, s.wr, 0, s.nf));
                XVA_TRY(conv0_im2col(c, s.run, s.wg, s.nf, s.nf));
                XVA_TRY(conv0_fwd(c, Pd, s.run, 0, 2 * s.nf));
            }
            for (int i = 1; i < s.run.n; ++i) {
                Seq x = c.S(s.run.t[i]), y = c.S(s.run.t[i + 1]);
                ConvEpi e;
                if (i < s.run.n - 1) { e.act = XVA_ACT_LRELU; e.act_slope = SLOPE; }
                XVA_TRY(hg_conv_fwd(x, y, cw(c, L[s.run.li[i]], Pd, 0), e, c.compute, c.st));
            }
        }
        if (losses) disc_losses(c, s.run, s.rt, s.r0, s.f0, s.nf, losses, reds[ln], loss_mask);
    }
    for (int ln = 0; ln < nl; ++ln)
        if (!reds[ln].empty()) XVA_TRY(xva_hg_reduce_batch(reds[ln].data(), (int)reds[ln].size(), cs[ln].st));
    XVA_TRY(join_lanes(cs, nl));
    return XVA_OK;
}

// D-step backward: gradients of sum_d [mean((1 - D(y))^2) + mean(D(G(x))^2)] w.r.t. all discriminator parameters
// Gradient buckets of the discriminators = the 8 discriminators in backward order (MPD 0..4, MSD 0..2); each owns one
// contiguous range of the flat buffer and is finalised (bias sums, reparametrisation backward) as soon as its backward is done.
constexpr int D_BUCKETS = NPER + 3;
int discs_backward_d(Ctx& c0, float* Pd, float* Gd, const float* yr, const float* yg, void* const* events) {
    std::vector<DiscSet> sets; std::vector<DiscRun> snr;
    build_sets(c0, yr, yg, sets, snr);
    std::vector<xva_cs_desc> colsums;
    XVA_TRY(zero_dweff(c0, c0.pl.dl));
    Ctx cs[MAXL]; const int nl = fork_lanes(c0, cs);
    int di = 0;
    for (auto& s : sets) {
        Ctx& c = cs[disc_lane(di, nl)];                      // each lane has its own split-K slabs
        xva_prof_tag(3000 + di * 10);
        if (!((disc_mask() >> di) & 1)) { XVA_TRY(record(c, events, di)); ++di; continue; }
        use_slabs(c);
        colsums.clear();
        const int n = s.run.n;
        const float inv = 1.f / (float)((int64_t)s.nf * s.run.t[n].T);
        Seq g = c.S(s.run.t[n]).slice(s.f0, s.nf), dg = c.S(s.run.d[n]).slice(s.f0, s.nf);
        XVA_TRY(xva_hg_seed_grad(nullptr, g.ptr(), dg.ptr(), c.dt, s.nf, g.Hp(), g.padF, g.T, 1, 0.f, inv, 2, 0, 0.f, 1, c.st));
        if (s.sn) {
            DiscRun rr = snr[0];
            Seq r = c.S(rr.t[n]), dr = c.S(rr.d[n]);
            XVA_TRY(xva_hg_seed_grad(r.ptr(), nullptr, dr.ptr(), c.dt, s.nf, r.Hp(), r.padF, r.T, 1, 0.f, inv, 3, 0, 0.f, 1, c.st));
            XVA_TRY(disc_backward_params(c, Pd, Gd, rr, 0, s.nf, s.wr, s.nb, nullptr, colsums));
            XVA_TRY(disc_backward_params(c, Pd, Gd, s.run, 0, s.nf, s.wg, s.nb, nullptr, colsums));
        } else {
            Seq r = c.S(s.run.t[n]).slice(s.r0, s.nf), dr = c.S(s.run.d[n]).slice(s.r0, s.nf);
            XVA_TRY(xva_hg_seed_grad(r.ptr(), nullptr, dr.ptr(), c.dt, s.nf, r.Hp(), r.padF, r.T, 1, 0.f, inv, 3, 0, 0.f, 1, c.st));
            XVA_TRY(disc_backward_params(c, Pd, Gd, s.run, 0, 2 * s.nf, s.wr, s.nb, s.wg, colsums));
        }
        XVA_TRY(xva_hg_colsum_batch(colsums.data(), (int)colsums.size(), c.st));
        XVA_TRY(fold_dwp(c, c.pl.dl, s.run.li, s.run.n));
        XVA_TRY(wn_backward(c, c.pl.dl, Pd, Gd, s.run.li, s.run.n));
        for (int q = 0; q < s.run.n; ++q) {
            const Layer& l = c.pl.dl[s.run.li[q]];
            if (l.kind != LK_SN) continue;
            for (int pass = 0; pass < 2; ++pass)
                XVA_TRY(xva_hg_spectral_norm_bwd(c.F(l.dweff[pass]), Pd + l.wv, c.F(l.su[pass]), c.F(l.sv[pass]), c.F(l.norm[pass]), Gd + l.wv, l.D0(), l.D1(),
                                                 l.k, c.F(c.pl.sn_tmp), c.st));
        }
        XVA_TRY(record(c, events, di));
        ++di;
    }
    XVA_TRY(join_lanes(cs, nl));
    return XVA_OK;
}

// G-step backward: d/d(fake wave) of sum_d [mean((1 - D(G))^2) + 2 * sum_l mean|fmap_l(y) - fmap_l(G)|]
int discs_backward_g(Ctx& c0, float* Pd, const float* yr, const float* yg, float* d_wav, float feat = 1.f) {
    const Plan& pl = c0.pl;
    std::vector<DiscSet> sets; std::vector<DiscRun> snr;
    build_sets(c0, yr, yg, sets, snr);
    Ctx cs[MAXL]; const int nl = fork_lanes(c0, cs);
    bool first[MAXL] = {true, true, true, true};
    int di = 0;
    for (auto& s : sets) {
        const int ln = disc_lane(di, nl);
        Ctx& c = cs[ln];
        xva_prof_tag(4000 + di * 10);
        if (!((disc_mask() >> di) & 1)) { ++di; continue; }
        use_slabs(c);
        const int sc = di >= NPER ? di - NPER : 0;
        // the full-rate discriminators all add into d(waveform): lane 0 into d_wav itself, every other lane into its own partial buffer
        float* dst = sc != 0 ? c.F(pl.dwav_s[sc]) : (ln == 0 ? d_wav : c.F(pl.dwav_lane[ln]));
        const bool acc = sc == 0 ? !first[ln] : false;
        XVA_TRY(disc_backward_wave(c, Pd, s.run, s.rt, s.r0, s.f0, s.nf, dst, acc ? 1 : 0, feat));
        if (sc == 0) first[ln] = false;
        ++di;
    }
    XVA_TRY(join_lanes(cs, nl));
    if (first[0]) XVA_TRY(zero(c0, d_wav, (int64_t)pl.B * pl.Tw[0] * 4));
    for (int ln = 1; ln < nl; ++ln)
        if (!first[ln]) XVA_TRY(xva_hg_add_f32(d_wav, c0.F(pl.dwav_lane[ln]), (int64_t)pl.B * pl.Tw[0], c0.st));
    if (pl.dnetp->vits) return XVA_OK;
    // pooled scales: d(y) += pool_bwd(d(pool(y))) ; scale 2 goes through scale 1
    XVA_TRY(xva_hg_avgpool_bwd(c0.F(pl.dwav_s[2]), c0.F(pl.dwav_s[1]), pl.B, pl.Tw[1], 1, c0.st));
    XVA_TRY(xva_hg_avgpool_bwd(c0.F(pl.dwav_s[1]), d_wav, pl.B, pl.Tw[0], 1, c0.st));
    return XVA_OK;
}

}  // namespace

// =========================================================================== C ABI ====
static const Net& net_of(int which) { return which == 0 ? (const Net&)gnet() : (const Net&)dnet(); }
extern "C" int64_t xva_hg_param_floats(int which) { return net_of(which).total; }
extern "C" int64_t xva_hg_trainable_floats(int which) { return net_of(which).trainable; }
extern "C" int xva_hg_num_tensors(int which) { return (int)net_of(which).t.size(); }
extern "C" int xva_hg_tensor_info(int which, int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim, int64_t* shape4,
                                  int32_t* kind) {
    const Net& n = net_of(which);
    XVA_CHECK_ARG(i >= 0 && i < (int)n.t.size() && name && name_cap > 0, "hg_tensor_info: bad index");
    const TInfo& ti = n.t[i];
    snprintf(name, name_cap, "%s", ti.name.c_str());
    if (offset) *offset = ti.off;
    if (numel) *numel = ti.numel;
    if (ndim) *ndim = ti.ndim;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = ti.shape[k];
    if (kind) *kind = ti.kind;
    return XVA_OK;
}
extern "C" int64_t xva_hg_workspace_bytes(const xva_hg_dims* d) {
    Plan p;
    if (make_plan(d, &p) != XVA_OK) return -1;
    return p.total;
}
extern "C" int xva_hg_generator_forward(const xva_hg_dims* d, const float* params_g, const float* mel, void* ws, int64_t ws_bytes, float* wav_out,
                                        void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, ws, ws_bytes, stream));
    XVA_CHECK_ARG(params_g && mel, "generator_forward: null");
    return gen_forward(c, params_g, mel, wav_out);
}
extern "C" void xva_hg_set_bucket_callback(void (*cb)(int, void*), void* user) { g_hg_bucket_cb = cb; g_hg_bucket_user = user; }
extern "C" int xva_hg_generator_backward_ex(const xva_hg_dims* d, const float* params_g, float* grads_g, const float* d_wav, void* ws, int64_t ws_bytes,
                                            void* const* bucket_events, void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, ws, ws_bytes, stream));
    XVA_CHECK_ARG(params_g && grads_g && d_wav, "generator_backward: null");
    return gen_backward(c, params_g, grads_g, d_wav, bucket_events);
}
extern "C" int xva_hg_generator_backward(const xva_hg_dims* d, const float* params_g, float* grads_g, const float* d_wav, void* ws, int64_t ws_bytes,
                                         void* stream) {
    return xva_hg_generator_backward_ex(d, params_g, grads_g, d_wav, ws, ws_bytes, nullptr, stream);
}
// ---- the VITS waveform decoder (xVAPitch): the same generator with a latent input, plain conv_pre / conv_post and the speaker projection ----
static int vits_dims(const xva_vits_dec_dims* d, xva_hg_dims* hd, const GenNet** gn) {
    XVA_CHECK_ARG(d && d->in_channels > 0 && d->in_channels % 8 == 0 && d->cond_channels >= 0 && d->cond_channels % 4 == 0,
                  "vits_dec: in_channels must be a positive multiple of 8, cond_channels a multiple of 4");
    hd->B = d->B; hd->seg = d->seg; hd->dt = d->dt;
    *gn = &vits_gnet(d->in_channels, d->cond_channels);
    return XVA_OK;
}
extern "C" int64_t xva_vits_dec_param_floats(const xva_vits_dec_dims* d) { xva_hg_dims hd; const GenNet* gn; return vits_dims(d, &hd, &gn) == XVA_OK ? gn->total : -1; }
extern "C" int xva_vits_dec_num_tensors(const xva_vits_dec_dims* d) { xva_hg_dims hd; const GenNet* gn; return vits_dims(d, &hd, &gn) == XVA_OK ? (int)gn->t.size() : -1; }
extern "C" int xva_vits_dec_tensor_info(const xva_vits_dec_dims* d, int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim,
                                        int64_t* shape4) {
    xva_hg_dims hd; const GenNet* gn;
    XVA_TRY(vits_dims(d, &hd, &gn));
    XVA_CHECK_ARG(i >= 0 && i < (int)gn->t.size() && name && name_cap > 0, "vits_dec_tensor_info: bad index");
    const TInfo& ti = gn->t[i];
    snprintf(name, name_cap, "%s", ti.name.c_str());
    if (offset) *offset = ti.off;
    if (numel) *numel = ti.numel;
    if (ndim) *ndim = ti.ndim;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = ti.shape[k];
    return XVA_OK;
}
extern "C" int64_t xva_vits_dec_workspace_bytes(const xva_vits_dec_dims* d) {
    xva_hg_dims hd; const GenNet* gn; Plan p;
    if (vits_dims(d, &hd, &gn) != XVA_OK || make_plan(&hd, &p, gn) != XVA_OK) return -1;
    return p.total;
}
extern "C" int xva_vits_dec_forward(const xva_vits_dec_dims* d, const float* params, const float* z, const float* g, void* ws, int64_t ws_bytes,
                                    float* wav_out, void* stream) {
    xva_hg_dims hd; const GenNet* gn; Ctx c;
    XVA_TRY(vits_dims(d, &hd, &gn));
    XVA_TRY(make_ctx(c, &hd, ws, ws_bytes, stream, gn));
    XVA_CHECK_ARG(params && z && (g || d->cond_channels == 0), "vits_dec_forward: null");
    return gen_forward(c, params, z, wav_out, g);
}
extern "C" int xva_vits_dec_backward(const xva_vits_dec_dims* d, const float* params, float* grads, const float* g, const float* d_wav, float* d_z,
                                     void* ws, int64_t ws_bytes, void* stream) {
    xva_hg_dims hd; const GenNet* gn; Ctx c;
    XVA_TRY(vits_dims(d, &hd, &gn));
    XVA_TRY(make_ctx(c, &hd, ws, ws_bytes, stream, gn));
    XVA_CHECK_ARG(params && grads && d_wav && (g || d->cond_channels == 0), "vits_dec_backward: null");
    return gen_backward(c, params, grads, d_wav, nullptr, g, d_z);
}
// ---- VitsDiscriminator (xVAPitch): five period discriminators + one scale discriminator, the same three passes as xva_hg_disc_* ----
extern "C" int64_t xva_vits_disc_param_floats(void) { return vits_dnet().total; }
extern "C" int xva_vits_disc_num_tensors(void) { return (int)vits_dnet().t.size(); }
extern "C" int xva_vits_disc_tensor_info(int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim, int64_t* shape4) {
    const Net& n = vits_dnet();
    XVA_CHECK_ARG(i >= 0 && i < (int)n.t.size() && name && name_cap > 0, "vits_disc_tensor_info: bad index");
    const TInfo& ti = n.t[i];
    snprintf(name, name_cap, "%s", ti.name.c_str());
    if (offset) *offset = ti.off;
    if (numel) *numel = ti.numel;
    if (ndim) *ndim = ti.ndim;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = ti.shape[k];
    return XVA_OK;
}
extern "C" int64_t xva_vits_disc_workspace_bytes(const xva_hg_dims* d) {
    Plan p;
    if (make_plan(d, &p, nullptr, &vits_dnet()) != XVA_OK) return -1;
    return p.total;
}
extern "C" int xva_vits_disc_forward(const xva_hg_dims* d, float* params_d, const float* yr, const float* yg, void* ws, int64_t ws_bytes, float* losses,
                                     int loss_mask, void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, ws, ws_bytes, stream, nullptr, &vits_dnet()));
    XVA_CHECK_ARG(params_d && yr && yg, "vits_disc_forward: null");
    return discs_forward(c, params_d, yr, yg, losses, loss_mask);
}
extern "C" int xva_vits_disc_backward_d(const xva_hg_dims* d, float* params_d, float* grads_d, const float* yr, const float* yg, void* ws, int64_t ws_bytes,
                                        void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, ws, ws_bytes, stream, nullptr, &vits_dnet()));
    XVA_CHECK_ARG(params_d && grads_d && yr && yg, "vits_disc_backward_d: null");
    return discs_backward_d(c, params_d, grads_d, yr, yg, nullptr);
}
extern "C" int xva_vits_disc_backward_g(const xva_hg_dims* d, float* params_d, const float* yr, const float* yg, float* d_wav, int feature_grad, void* ws,
                                        int64_t ws_bytes, void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, ws, ws_bytes, stream, nullptr, &vits_dnet()));
    XVA_CHECK_ARG(params_d && yr && yg && d_wav, "vits_disc_backward_g: null");
    return discs_backward_g(c, params_d, yr, yg, d_wav, feature_grad ? 1.f : 0.f);
}
// Where an activation tensor of the last forward lives in the caller's workspace (parity tests feed a CPU restatement of ONE layer with
// the engine's own input and compare outputs: storage-dtype rounding is then checked layer by layer instead of through ~50 layers).
extern "C" int xva_hg_slot(const xva_hg_dims* d, int kind, int i0, int i1, int i2, int64_t* off_bytes, int32_t* geom5) {
    Plan pl;
    XVA_TRY(make_plan(d, &pl));
    XVA_CHECK_ARG(off_bytes && geom5, "hg_slot: null output");
    const SeqSpec* s = nullptr;
    auto in = [](int v, int n) { return v >= 0 && v < n; };
    switch (kind) {
        case 0: s = &pl.xin; break;
        case 1: s = &pl.h0; break;
        case 2: if (in(i0, 4)) s = &pl.u[i0]; break;
        case 3: if (in(i0, 4)) s = &pl.ua[i0]; break;
        case 4: if (in(i0, 12) && in(i1, 3)) s = &pl.xt1[i0][i1]; break;
        case 5: if (in(i0, 12) && in(i1, 2)) s = &pl.xr[i0][i1]; break;
        case 6: if (in(i0, 12) && in(i1, 2)) s = &pl.xra[i0][i1]; break;
        case 7: if (in(i0, 4)) s = &pl.xs[i0]; break;
        case 8: s = &pl.y; break;
        case 9: if (in(i0, NPER) && in(i1, 7)) s = &pl.pt[i0][i1]; break;
        case 10: if (in(i0, 3) && in(i1, 2) && in(i2, 9)) s = &pl.st[i0][i1][i2]; break;
        default: break;
    }
    XVA_CHECK_ARG(s != nullptr, "hg_slot: unknown slot (%d, %d, %d, %d)", kind, i0, i1, i2);
    *off_bytes = s->off;
    geom5[0] = s->nseq; geom5[1] = s->T; geom5[2] = s->C; geom5[3] = s->padF; geom5[4] = s->padB;
    return XVA_OK;
}
extern "C" int xva_hg_set_disc_mask(int m) { int old = g_disc_mask; g_disc_mask = m; return old; }   // profiling only (tools/hg_disc_split.py)
extern "C" int xva_hg_set_streams(int n) { int old = g_hg_serial ? 1 : side_streams().n; g_hg_serial = n <= 1; return old; }
extern "C" int xva_hg_num_buckets(int which) { return which == 0 ? G_BUCKETS : D_BUCKETS; }
// [begin, end) in floats of bucket i of the flat gradient buffer `which`, in backward-completion order
extern "C" int xva_hg_bucket_range(int which, int i, int64_t* begin, int64_t* end) {
    XVA_CHECK_ARG((which == 0 || which == 1) && i >= 0 && i < xva_hg_num_buckets(which) && begin && end, "hg_bucket_range: bad index");
    const Net& n = net_of(which);
    auto span = [&](const std::string& first, const std::string& last_prefix, int64_t* b, int64_t* e) {
        *b = -1; *e = -1;
        for (const TInfo& ti : n.t) {
            if (ti.kind != 0) continue;
            if (*b < 0 && ti.name.rfind(first, 0) == 0) *b = ti.off;
            if (ti.name.rfind(last_prefix, 0) == 0) *e = ti.off + ((ti.numel + 3) & ~(int64_t)3);
        }
    };
    if (which == 0) {
        if (i < 4) {
            const int st = 3 - i;
            span("resblocks." + std::to_string(st * 3) + ".", "resblocks." + std::to_string(st * 3 + 2) + ".", begin, end);
        } else if (i == 4) span("conv_pre.", "ups.3.", begin, end);
        else span("conv_post.", "conv_post.", begin, end);
    } else {
        const std::string pre = i < NPER ? "mpd.discriminators." + std::to_string(i) + "." : "msd.discriminators." + std::to_string(i - NPER) + ".";
        span(pre, pre, begin, end);
    }
    XVA_CHECK_ARG(*begin >= 0 && *end > *begin, "hg_bucket_range: empty bucket");
    return XVA_OK;
}

/* yr / yg: real / generated waveforms (B, seg) fp32.  losses (device, 4 floats, may be NULL): {discriminator loss,
 * generator LSGAN loss, feature-matching loss, -}.  params_d is non-const: the spectral-norm power iteration advances
 * weight_u / weight_v (one iteration per pass, python/hifigan/models.py:244-253). */
extern "C" int xva_hg_disc_forward_ex(const xva_hg_dims* d, float* params_d, const float* yr, const float* yg, void* ws, int64_t ws_bytes, float* losses,
                                      int loss_mask, void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, ws, ws_bytes, stream));
    XVA_CHECK_ARG(params_d && yr && yg, "disc_forward: null");
    return discs_forward(c, params_d, yr, yg, losses, loss_mask);
}
extern "C" int xva_hg_disc_forward(const xva_hg_dims* d, float* params_d, const float* yr, const float* yg, void* ws, int64_t ws_bytes, float* losses,
                                   void* stream) {
    return xva_hg_disc_forward_ex(d, params_d, yr, yg, ws, ws_bytes, losses, 3, stream);
}
extern "C" int xva_hg_disc_backward_d_ex(const xva_hg_dims* d, float* params_d, float* grads_d, const float* yr, const float* yg, void* ws,
                                         int64_t ws_bytes, void* const* bucket_events, void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, ws, ws_bytes, stream));
    XVA_CHECK_ARG(params_d && grads_d && yr && yg, "disc_backward_d: null");
    return discs_backward_d(c, params_d, grads_d, yr, yg, bucket_events);
}
extern "C" int xva_hg_disc_backward_d(const xva_hg_dims* d, float* params_d, float* grads_d, const float* yr, const float* yg, void* ws, int64_t ws_bytes,
                                      void* stream) {
    return xva_hg_disc_backward_d_ex(d, params_d, grads_d, yr, yg, ws, ws_bytes, nullptr, stream);
}
extern "C" int xva_hg_disc_backward_g(const xva_hg_dims* d, float* params_d, const float* yr, const float* yg, float* d_wav, void* ws, int64_t ws_bytes,
                                      void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, ws, ws_bytes, stream));
    XVA_CHECK_ARG(params_d && yr && yg && d_wav, "disc_backward_g: null");
    return discs_backward_g(c, params_d, yr, yg, d_wav);
}
