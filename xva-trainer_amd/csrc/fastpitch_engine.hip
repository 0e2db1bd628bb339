// fastpitch_engine.hip — the FastPitch1.1 forward / backward schedule as ONE host call each.
//
// Reference: FastPitch.forward (python/fastpitch1_1/fastpitch/model.py:325-423), FFTransformer /
// TransformerLayer / MultiHeadAttn / PositionwiseConvFF (fastpitch/transformer.py:59-243), TemporalPredictor
// (model.py:103-122), regulate_len / average_pitch (model.py:59-100) and the autograd backward PyTorch derives
// for them.  The reference issues ~1.5k tiny ATen kernels per step from Python; here the whole step is a fixed
// C++ launch schedule over a caller-owned workspace (no allocation, no sync, graph-capturable), every dense
// contraction on the MFMA GEMM (gemm.hip) with bias / ReLU / residual / mask fused in its epilogue.
//
// Memory model: parameters live in ONE flat fp32 buffer (table below; conv-k3 weights are stored tap-major
// [Cout][3][Cin] — the host converts to/from the checkpoint layout [Cout][Cin][3]); gradients mirror it.
// Activations use the padded token-major layout described in fp_ops.hip, stored in fp32 (parity mode) or bf16
// (training mode: every GEMM operand is read as bf16, halving operand traffic; a bf16 shadow of the parameters
// is refreshed by one cast kernel per forward).  Dropout (p = 0.1 in the reference's training mode) uses stateless
// hash masks regenerated in backward: nothing is stored for it except the dropped attention probabilities.
#include "xva_common.h"
#include "../../include/xva_hip.h"
#include "../../include/xva_gemm.h"
#include <string>
#include <vector>

// per-op launchers (fp_ops.hip) are declared in include/xva_hip.h

namespace {

constexpr int NL = 6, DM = 384, DI = 1536, DH = 64, DQKV = 192, DP = 256, NMEL = 80, NSYM = 148;

// ------------------------------------------------------------------ parameter table ----
struct TensorInfo {
    std::string name;
    int64_t offset, numel;
    int ndim;
    int64_t shape[4];  // checkpoint (reference) shape
    int kind;          // 0 plain, 1 conv-k3 stored tap-major
};

struct LayerP { int64_t qkv_w, qkv_b, o_w, ln1_g, ln1_b, c1_w, c1_b, c2_w, c2_b, ln2_g, ln2_b; };
struct PredP { int64_t c1_w, c1_b, n1_g, n1_b, c2_w, c2_b, n2_g, n2_b, fc_w, fc_b; };
struct AttnP { int64_t q0_w, q0_b, q2_w, q2_b, q4_w, q4_b, k0_w, k0_b, k2_w, k2_b; };
struct ParamTable {
    std::vector<TensorInfo> t;
    AttnP at;
    int64_t total = 0;
    int64_t word_emb;
    LayerP enc[NL], dec[NL];
    PredP dur, pitch, energy;
    int64_t pitch_emb_w, pitch_emb_b, energy_emb_w, energy_emb_b, proj_w, proj_b;
    // contiguous [begin, end) ranges of the flat buffer by module group (used for stage freezing)
    int64_t enc_begin, enc_end, dur_begin, dur_end, pitch_begin, pitch_end, pemb_begin, pemb_end, energy_begin, energy_end,
        eemb_begin, eemb_end, dec_begin, dec_end, proj_begin, proj_end, attn_begin, attn_end;

    int64_t add(const std::string& name, std::initializer_list<int64_t> shape, int kind = 0) {
        TensorInfo ti;
        ti.name = name; ti.kind = kind; ti.ndim = (int)shape.size();
        ti.numel = 1; int i = 0;
        for (auto s : shape) { ti.shape[i++] = s; ti.numel *= s; }
        for (; i < 4; ++i) ti.shape[i] = 1;
        ti.offset = total;
        total += (ti.numel + 7) & ~(int64_t)7;  // keep every tensor 32-byte aligned (16 bytes in the bf16 shadow)
        t.push_back(ti);
        return ti.offset;
    }
    void add_layer(const std::string& p, LayerP& L) {
        L.qkv_w = add(p + "dec_attn.qkv_net.weight", {DQKV, DM});
        L.qkv_b = add(p + "dec_attn.qkv_net.bias", {DQKV});
        L.o_w = add(p + "dec_attn.o_net.weight", {DM, DH});
        L.ln1_g = add(p + "dec_attn.layer_norm.weight", {DM});
        L.ln1_b = add(p + "dec_attn.layer_norm.bias", {DM});
        L.c1_w = add(p + "pos_ff.CoreNet.0.weight", {DI, DM, 3}, 1);
        L.c1_b = add(p + "pos_ff.CoreNet.0.bias", {DI});
        L.c2_w = add(p + "pos_ff.CoreNet.2.weight", {DM, DI, 3}, 1);
        L.c2_b = add(p + "pos_ff.CoreNet.2.bias", {DM});
        L.ln2_g = add(p + "pos_ff.layer_norm.weight", {DM});
        L.ln2_b = add(p + "pos_ff.layer_norm.bias", {DM});
    }
    void add_pred(const std::string& p, PredP& P) {
        P.c1_w = add(p + "layers.0.conv.weight", {DP, DM, 3}, 1);
        P.c1_b = add(p + "layers.0.conv.bias", {DP});
        P.n1_g = add(p + "layers.0.norm.weight", {DP});
        P.n1_b = add(p + "layers.0.norm.bias", {DP});
        P.c2_w = add(p + "layers.1.conv.weight", {DP, DP, 3}, 1);
        P.c2_b = add(p + "layers.1.conv.bias", {DP});
        P.n2_g = add(p + "layers.1.norm.weight", {DP});
        P.n2_b = add(p + "layers.1.norm.bias", {DP});
        P.fc_w = add(p + "fc.weight", {1, DP});
        P.fc_b = add(p + "fc.bias", {1});
    }
    ParamTable() {
        enc_begin = total;
        word_emb = add("encoder.word_emb.weight", {NSYM, DM});
        for (int i = 0; i < NL; ++i) add_layer("encoder.layers." + std::to_string(i) + ".", enc[i]);
        enc_end = dur_begin = total;
        add_pred("duration_predictor.", dur);
        dur_end = pitch_begin = total;
        add_pred("pitch_predictor.", pitch);
        pitch_end = pemb_begin = total;
        pitch_emb_w = add("pitch_emb.weight", {DM, 1, 3});
        pitch_emb_b = add("pitch_emb.bias", {DM});
        pemb_end = energy_begin = total;
        add_pred("energy_predictor.", energy);
        energy_end = eemb_begin = total;
        energy_emb_w = add("energy_emb.weight", {DM, 1, 3});
        energy_emb_b = add("energy_emb.bias", {DM});
        eemb_end = dec_begin = total;
        for (int i = 0; i < NL; ++i) add_layer("decoder.layers." + std::to_string(i) + ".", dec[i]);
        dec_end = proj_begin = total;
        proj_w = add("proj.weight", {NMEL, DM});
        proj_b = add("proj.bias", {NMEL});
        proj_end = attn_begin = total;
        // Stage-1 aligner (ConvAttention, attention.py:82-133): query_proj = conv3(80->160) ReLU conv1(160->80) ReLU conv1(80->80),
        // key_proj = conv3(384->768) ReLU conv1(768->80); attn_proj is an unused member of the reference module (checkpoint only)
        at.q0_w = add("attention.query_proj.0.conv.weight", {160, 80, 3}, 1);
        at.q0_b = add("attention.query_proj.0.conv.bias", {160});
        at.q2_w = add("attention.query_proj.2.conv.weight", {80, 160, 1});
        at.q2_b = add("attention.query_proj.2.conv.bias", {80});
        at.q4_w = add("attention.query_proj.4.conv.weight", {80, 80, 1});
        at.q4_b = add("attention.query_proj.4.conv.bias", {80});
        add("attention.attn_proj.weight", {1, 80, 1, 1});
        add("attention.attn_proj.bias", {1});
        at.k0_w = add("attention.key_proj.0.conv.weight", {768, 384, 3}, 1);
        at.k0_b = add("attention.key_proj.0.conv.bias", {768});
        at.k2_w = add("attention.key_proj.2.conv.weight", {80, 768, 1});
        at.k2_b = add("attention.key_proj.2.conv.bias", {80});
        attn_end = total;
    }
};
const ParamTable& table() { static ParamTable T; return T; }

// ------------------------------------------------------------------ workspace plan ----
// All offsets are BYTES.  Activation tensors have element size es (4 or 2); statistics / token scalars are fp32.
struct Bump {
    int64_t cur = 0;
    int64_t take(int64_t bytes) { int64_t o = cur; cur += (bytes + 255) & ~(int64_t)255; return o; }
    // sequence buffer of `rows` x C elements of size es with one guard row before and after; returns the offset of row 0
    int64_t seq(int64_t rows, int C, int es) { int64_t o = take((rows + 2) * (int64_t)C * es); return o + (int64_t)C * es; }
};

constexpr int QKV_SPARE = 8;
struct LayerA { int64_t qkv, P, Pd, lse, av, sum1, mean1, rstd1, y1, h, sum2, mean2, rstd2, xp, yp; };   // xp / yp: the layer input and y1 as split-bf16 pairs (planes mode)
struct PredA { int64_t c1, m1, r1, n1, c2, m2, r2, n2, out; };
struct Plan {
    int B, Tt, Tm, Ttp, Tmp, es, dt;
    int64_t Re, Rd, Tse, Tsd;
    int64_t enc_x[NL + 1], dec_x[NL + 1];
    LayerA enc[NL], dec[NL];
    PredA dur, pitch, energy;
    int64_t pin_a, pin_b;   // fp32 copies of the predictor inputs (enc_out, enc_c1) in bf16 mode; -1 in fp32 mode
    int64_t ptgt, etgt, enc_c1, enc_c2, tok, tstart, dec_lens, mel_out, dur_pred;
    int64_t acc, losses, d_mel, d_pitch, d_energy, d_logdur;
    // backward scratch, sized for the decoder; the encoder reuses it.  gBm / gDm: dropout-masked copies for the branches.
    int64_t gA, gB, gBm, gC, gD, gDm, gH, gAV, gP, gQKV, gE, pa, pb;
    // second copies of the tensors the weight-gradient lane reads (layers alternate between the two sets: see layers_bwd)
    int64_t gB2, gBm2, gD2, gDm2, gH2, gQKV2;
    int64_t skws, skws2, skws3, skws_bytes;   // split-K slab scratch of the weight-gradient GEMMs (main stream / weight-gradient lane / predictor lane)
    int64_t gX1, gX2;   // d(encoder output) contributions of the energy / pitch predictors (predictor lane)
    int64_t wshadow;    // activation-dtype copy of the flat parameters (bf16 mode); unused in fp32 mode
    // fp32 mode with split-bf16 products (xva_gemm_set_fp32_products(1)): the feed-forward convolutions (94 % of the FLOPs) run the direct-to-LDS kernels on
    // split-bf16 PLANES (include/xva_gemm.h): the whole parameter table as a pair (wplanes; lo plane wplane_stride elements after hi), transposed tap-reversed
    // pairs of both convolution weights of the 2 x NL layers (wtp_c1 / wtp_c2), and per layer parity a pair buffer for y1 and for d(sum2) (yp / gp; rows -1 .. R)
    int64_t wplanes, wplane_stride, wtp_c1, wtp_c2;
    int64_t gp[2][2], dp[2][2], gPp;   // [stack: 0 encoder, 1 decoder][layer parity]   // the same for the layer input x and d(sum1) (rows -1 .. R, DM channels), and d(scores) as a pair (B x Tp x Ts)
    int64_t wt_c2;      // bf16 mode: transposed, tap-reversed copies of the 2 x NL conv2 weights ([DI][3][DM] each; encoder layers first) for the NT backward-data form
    int64_t total;
};

// Backward-data of the feed-forward's second convolution through a transposed weight copy and the NT main loop (1, default) or the NN loop on the
// weight as stored (0). A/B and test switch (the results differ by fp32 summation order only: same products, same K order).
static int g_bwd_nt = 1;
extern "C" int xva_fp_set_bwd_nt(int mode) { int old = g_bwd_nt; g_bwd_nt = mode; return old; }

// fp32 mode, split products: 1 (default) = the feed-forward convolutions through split-bf16 planes on the direct-to-LDS kernels (round 5), 0 = every product on
// the register-staged kernel that splits while staging (rounds 3 - 4).  Changes the workspace plan: set before xva_fp_workspace_bytes.  env XVA_FP_FFN_PLANES
static int g_ffn_planes = [] { const char* e = getenv("XVA_FP_FFN_PLANES"); return e ? atoi(e) : 1; }();
extern "C" int xva_fp_set_ffn_planes(int mode) { int old = g_ffn_planes; g_ffn_planes = mode; return old; }
// the process-global switches the workspace plan depends on, as one word: a caller that caches xva_fp_workspace_bytes / xva_fp_slot_offset keys its cache on it
extern "C" int xva_fp_plan_knobs(void) { return (g_ffn_planes ? 1 : 0) | (g_bwd_nt ? 2 : 0); }

// bf16 mode: 1 (default) = o_net + dropout + residual + LayerNorm of a transformer layer's attention block as one kernel (xva_fp_onet_ln_fwd), 0 = GEMM + LayerNorm
static int g_onet_fused = [] { const char* e = getenv("XVA_FP_ONET_FUSED"); return e ? atoi(e) : 1; }();
extern "C" int xva_fp_set_onet_fused(int mode) { int old = g_onet_fused; g_onet_fused = mode; return old; }
// fp32 mode, split products on pairs: 1 (default) = the attention core as the flash-style kernels on pairs (attention.hip: xva_fp_attention_*_pairs), 0 = scores ->
// softmax -> P V through HBM (two T x T fp32 tensors per layer and direction)
static int g_att_flash = 1;
extern "C" void xva_fp_set_att_flash(int on) { g_att_flash = on; }
// the same mode: 1 (default) = the LayerNorm kernels leave y1 / the next layer's input / d(sum) ALSO as split-bf16 pairs (xva_fp_layernorm_*_pair) and the
// backward reads the forward's pairs, 0 = a split launch in front of every product that reads them (6 per layer)
static int g_ln_pairs = 1;
extern "C" void xva_fp_set_ln_pairs(int on) { g_ln_pairs = on; }

int make_plan(const xva_fp_dims* d, Plan* p) {
    XVA_CHECK_ARG(d && d->B > 0 && d->Tt > 0 && d->Tm > 0, "fastpitch: bad dims");
    XVA_CHECK_ARG(d->stage >= 2 && d->stage <= 4, "fastpitch: stage must be 2, 3 or 4 (stage 1 aligner is not built yet)");
    XVA_CHECK_ARG(d->Tm + 2 <= 2048 && d->Tt + 2 <= 2048, "fastpitch: sequence longer than 2046 unsupported");
    XVA_CHECK_ARG(d->p_dropout >= 0.f && d->p_dropout < 1.f, "fastpitch: bad dropout probability");
    p->B = d->B; p->Tt = d->Tt; p->Tm = d->Tm; p->Ttp = d->Tt + 2; p->Tmp = d->Tm + 2;
    XVA_CHECK_ARG(d->compute >= 0 && d->compute <= 2, "fastpitch: compute must be 0 (fp32), 1 (bf16) or 2 (fp16 operands, fp32 residual stream)");
    // compute 2 (round 6): the storage plan of the fp32 mode with the planes path on — the residual stream, LayerNorm inputs / outputs and every gradient of them
    // fp32, the MFMA operands (layer inputs, y1, qkv, A V, the feed-forward intermediate h and their gradients, all weights) single IEEE-half tensors where
    // the split-products mode keeps bf16 pairs
    const bool bf16s = d->compute == 1, h16 = d->compute == 2, planned_planes = h16 || (d->compute == 0 && g_ffn_planes);
    p->dt = bf16s ? XVA_BF16 : XVA_F32; p->es = bf16s ? 2 : 4;
    const int es = p->es;
    const bool drop = d->p_dropout > 0.f;
    const bool fused = bf16s;
    p->Re = (int64_t)d->B * p->Ttp; p->Rd = (int64_t)d->B * p->Tmp;
    p->Tse = (p->Ttp + 7) & ~7; p->Tsd = (p->Tmp + 7) & ~7;
    Bump b;
    auto plan_layers = [&](int64_t R, int Tp, int64_t Ts, int64_t* x, LayerA* L) {
        x[0] = b.seq(R, DM, es);
        for (int i = 0; i < NL; ++i) {
            // fp32 mode with the split-products planes path: QKV_SPARE spare rows — the products over the keys take K = Ts = round-up-8(Tp) and read up to 7 rows past the last item
            L[i].qkv = b.seq(R + (planned_planes ? QKV_SPARE : 0), DQKV, es);
            // bf16 mode runs the fused attention kernels (attention.hip): no (T x T) probability matrices, one logsumexp per row
            L[i].P = fused ? -1 : b.take((int64_t)p->B * Tp * Ts * es + 64);
            // (the split-products planes path keeps the dropped copy as a split-bf16 pair in the same bytes: it needs its own buffer without dropout too)
            L[i].Pd = fused ? -1 : ((drop || planned_planes) ? b.take((int64_t)p->B * Tp * Ts * es + 64) : L[i].P);
            L[i].lse = fused ? b.take(R * 4) : -1;
            L[i].av = b.seq(R, DH, es);
            L[i].sum1 = b.seq(R, DM, es);
            L[i].mean1 = b.take(R * 4); L[i].rstd1 = b.take(R * 4);
            L[i].y1 = b.seq(R, DM, es);
            L[i].h = b.seq(R, DI, es);
            L[i].sum2 = b.seq(R, DM, es);
            L[i].mean2 = b.take(R * 4); L[i].rstd2 = b.take(R * 4);
            x[i + 1] = b.seq(R, DM, es);
            // fp32 mode, split products: the layer input and y1 ALSO as split-bf16 pairs, kept from the forward pass for the weight gradients (rows -1 .. R of
            // two planes: the bytes of one fp32 sequence slot; guard rows are never written and stay zero)
            L[i].xp = L[i].yp = -1;
            if (planned_planes) { L[i].xp = b.take(2 * (R + 2) * DM * 2); L[i].yp = b.take(2 * (R + 2) * DM * 2); }
        }
    };
    auto plan_pred = [&](PredA& A) {
        // predictors are stored fp32 in both modes (see pred_fwd)
        A.c1 = b.seq(p->Re, DP, 4); A.m1 = b.take(p->Re * 4); A.r1 = b.take(p->Re * 4); A.n1 = b.seq(p->Re, DP, 4);
        A.c2 = b.seq(p->Re, DP, 4); A.m2 = b.take(p->Re * 4); A.r2 = b.take(p->Re * 4); A.n2 = b.seq(p->Re, DP, 4);
        A.out = b.take((p->Re + 8) * 4) + 16;
    };
    plan_layers(p->Re, p->Ttp, p->Tse, p->enc_x, p->enc);
    plan_pred(p->dur); plan_pred(p->pitch); plan_pred(p->energy);
    p->pin_a = bf16s ? b.seq(p->Re, DM, 4) : -1; p->pin_b = bf16s ? b.seq(p->Re, DM, 4) : -1;
    p->ptgt = b.take((p->Re + 8) * 4) + 16; p->etgt = b.take((p->Re + 8) * 4) + 16;
    p->enc_c1 = b.seq(p->Re, DM, es); p->enc_c2 = b.seq(p->Re, DM, es);
    p->tok = b.take((int64_t)p->B * p->Tm * 4); p->tstart = b.take((int64_t)p->B * (p->Tt + 1) * 4); p->dec_lens = b.take(p->B * 4);
    plan_layers(p->Rd, p->Tmp, p->Tsd, p->dec_x, p->dec);
    p->mel_out = b.seq(p->Rd, NMEL, es);
    p->dur_pred = b.take((p->Re + 8) * 4) + 16;
    p->acc = b.take(8 * 4); p->losses = b.take(8 * 4);
    p->d_mel = b.seq(p->Rd, NMEL, es);
    p->d_pitch = b.take((p->Re + 8) * 4) + 16; p->d_energy = b.take((p->Re + 8) * 4) + 16; p->d_logdur = b.take((p->Re + 8) * 4) + 16;
    int64_t Rm = p->Rd > p->Re ? p->Rd : p->Re;
    int64_t Tsm = p->Tsd > p->Tse ? p->Tsd : p->Tse;
    int Tpm = p->Tmp > p->Ttp ? p->Tmp : p->Ttp;
    p->gA = b.seq(Rm, DM, es); p->gB = b.seq(Rm, DM, es); p->gC = b.seq(Rm, DM, es); p->gD = b.seq(Rm, DM, es);
    p->gBm = drop ? b.seq(Rm, DM, es) : p->gB; p->gDm = drop ? b.seq(Rm, DM, es) : p->gD;
    p->gH = b.seq(Rm, DI, es); p->gAV = b.seq(Rm, DH, es); p->gP = fused ? b.take(Rm * 4) : b.take((int64_t)p->B * Tpm * Tsm * es + 64); p->gQKV = b.seq(Rm, DQKV, es);
    p->gE = b.seq(p->Re, DM, es); p->pa = b.seq(p->Re, DP, 4); p->pb = b.seq(p->Re, DP, 4);
    p->skws_bytes = (int64_t)8 * DI * 3 * DM * 4;   // 8 splits of the largest weight gradient (1536 x 1152 fp32)
    p->skws = b.take(p->skws_bytes);
    p->skws2 = b.take(p->skws_bytes); p->skws3 = b.take(p->skws_bytes);
    p->gX1 = b.seq(p->Re, DM, es); p->gX2 = b.seq(p->Re, DM, es);
    p->gB2 = b.seq(Rm, DM, es); p->gD2 = b.seq(Rm, DM, es);
    p->gBm2 = drop ? b.seq(Rm, DM, es) : p->gB2; p->gDm2 = drop ? b.seq(Rm, DM, es) : p->gD2;
    p->gH2 = b.seq(Rm, DI, es); p->gQKV2 = b.seq(Rm, DQKV, es);
    p->wshadow = bf16s ? b.take(table().total * es) : -1;
    p->wt_c2 = (bf16s && g_bwd_nt) ? b.take((int64_t)2 * NL * DI * 3 * DM * 2) : -1;
    p->wplanes = p->wtp_c1 = p->wtp_c2 = p->gPp = -1;
    for (int st = 0; st < 2; ++st) for (int q = 0; q < 2; ++q) p->gp[st][q] = p->dp[st][q] = -1;
    p->wplane_stride = (table().total + 7) / 8 * 8;
    if (planned_planes) {
        p->wplanes = b.take(2 * p->wplane_stride * 2);
        p->wtp_c1 = b.take((int64_t)2 * (2 * NL) * DI * 3 * DM * 2); p->wtp_c2 = b.take((int64_t)2 * (2 * NL) * DI * 3 * DM * 2);
        for (int st = 0; st < 2; ++st)           // d(sum2) / d(sum1) masked by their dropouts, as pairs: per stack (the guard rows sit at the stack's own R)
            for (int q = 0; q < 2; ++q) {
                const int64_t R = st ? p->Rd : p->Re;
                p->gp[st][q] = b.take(2 * (R + 2) * DM * 2); p->dp[st][q] = b.take(2 * (R + 2) * DM * 2);
            }
        p->gPp = b.take((int64_t)p->B * Tpm * Tsm * 4 + 64);
    }
    p->total = b.cur;
    static const bool dump = getenv("XVA_FP_DUMP_PLAN") != nullptr;      // debugging: the byte offsets of the plan's slots
    if (dump) {
        fprintf(stderr, "PLAN enc_x0 %ld", (long)p->enc_x[0]);
        for (int i = 0; i < NL; ++i) fprintf(stderr, " | enc%d qkv %ld P %ld Pd %ld av %ld sum1 %ld y1 %ld h %ld sum2 %ld x %ld", i, (long)p->enc[i].qkv, (long)p->enc[i].P, (long)p->enc[i].Pd,
                                             (long)p->enc[i].av, (long)p->enc[i].sum1, (long)p->enc[i].y1, (long)p->enc[i].h, (long)p->enc[i].sum2, (long)p->enc_x[i + 1]);
        fprintf(stderr, " | dur.c1 %ld pitch.c1 %ld energy.c1 %ld energy.out %ld ptgt %ld etgt %ld enc_c1 %ld enc_c2 %ld tok %ld dec_x0 %ld dec0.qkv %ld mel_out %ld gA %ld gH %ld gAV %ld gP %ld gQKV %ld gE %ld pa %ld pb %ld skws %ld gX1 %ld gX2 %ld wplanes %ld total %ld\n",
                (long)p->dur.c1, (long)p->pitch.c1, (long)p->energy.c1, (long)p->energy.out, (long)p->ptgt, (long)p->etgt, (long)p->enc_c1, (long)p->enc_c2, (long)p->tok, (long)p->dec_x[0],
                (long)p->dec[0].qkv, (long)p->mel_out, (long)p->gA, (long)p->gH, (long)p->gAV, (long)p->gP, (long)p->gQKV, (long)p->gE, (long)p->pa, (long)p->pb, (long)p->skws, (long)p->gX1, (long)p->gX2,
                (long)p->wplanes, (long)p->total);
    }
    return XVA_OK;
}

// ------------------------------------------------------------------ GEMM helpers ----
struct Ctx {
    const xva_fp_dims* d;
    Plan pl;
    const float* P;   // fp32 parameters
    const char* Pw;   // GEMM-operand view of the parameters (activation dtype): the shadow in bf16 mode, P itself in fp32 mode
    float* G;         // grads (may be null in forward)
    char* W;          // workspace
    void* st;
    int compute, dt, es;   // compute: 1 = bf16 storage (the throughput mode), 0 = fp32 storage
    bool h16 = false;      // fp32 storage, the planes schedule on single IEEE-half operand tensors (dims.compute == 2)
    float pd;         // dropout probability
    uint64_t seed;
    int lane = 0;     // 0: the caller's stream, 1: the weight-gradient side stream, 2: the predictor side stream (own split-K slabs each)
    void* const* events = nullptr;  // optional hipEvent_t per gradient bucket (data-parallel overlap)
    int ev_base = 0;
    int record(int i);
    char* A(int64_t off) const { return W + off; }                       // activation tensor at byte offset
    float* F(int64_t off) const { return (float*)(W + off); }            // fp32 tensor at byte offset
    const void* wt(int64_t elem_off) const { return Pw + elem_off * es; } // parameter tensor as a GEMM operand
    char* sh(char* p, int64_t elems) const { return p + elems * es; }     // shift an activation pointer by elements
    const void* wt_c2(const struct LayerP* LP, int l) const;              // transposed conv2 weight of layer l of stack LP (null: not planned)
};

const void* Ctx::wt_c2(const LayerP* LP, int l) const {
    if (pl.wt_c2 < 0 || !compute) return nullptr;
    const ParamTable& T = table();
    const int idx = LP == T.enc ? l : (LP == T.dec ? NL + l : -1);
    return idx < 0 ? nullptr : W + pl.wt_c2 + (int64_t)idx * DI * 3 * DM * 2;
}
// the transposed conv2 weights of both stacks from the fp32 parameters (one launch; see xva_fp_wt_transpose3)
static int refresh_wt_c2(const Ctx& c, const float* params, void* st) {
    if (c.pl.wt_c2 < 0) return XVA_OK;
    const ParamTable& T = table();
    int64_t so[2 * NL], dof[2 * NL];
    for (int l = 0; l < NL; ++l) {
        so[l] = T.enc[l].c2_w; dof[l] = (int64_t)l * DI * 3 * DM;
        so[NL + l] = T.dec[l].c2_w; dof[NL + l] = (int64_t)(NL + l) * DI * 3 * DM;
    }
    return xva_fp_wt_transpose3(params, c.W + c.pl.wt_c2, so, dof, 2 * NL, DM, DI, st);
}
// Bucket i's gradients are final on the lane this context issues to: record its event there and, when the data-parallel host has registered a
// callback, call it NOW — while the host is still issuing backward.  The host enqueues the bucket's wait + all-reduce from inside it.  Measured
// (tools/dp_overlap_probe.py): a hipStreamWaitEvent issued only after the whole backward had been issued resolved when the recording lane had DRAINED —
// every bucket's exchange started at the end of backward, whatever the event flags; issued right behind the record it resolves at the record.
typedef void (*xva_bucket_cb_t)(int bucket, void* user);
static thread_local xva_bucket_cb_t g_bucket_cb = nullptr;
static thread_local void* g_bucket_user = nullptr;
extern "C" void xva_fp_set_bucket_callback(xva_bucket_cb_t cb, void* user) { g_bucket_cb = cb; g_bucket_user = user; }
int Ctx::record(int i) {
    if (!events || !events[i]) return XVA_OK;
    if (hipEventRecord((hipEvent_t)events[i], (hipStream_t)st) != hipSuccess) { xva_set_error("bucket event record failed"); return XVA_ERR_HIP; }
    if (g_bucket_cb) g_bucket_cb(i, g_bucket_user);
    return XVA_OK;
}

static xva_gemm_params gp0(const Ctx& c) {
    xva_gemm_params g;
    memset(&g, 0, sizeof(g));
    // (fp16-operand mode: the products that stay on fp32-stored operands — temporal predictors, projection, toy sequences — take the split-bf16 products
    // of the register-staged kernel, ~1e-5 per product, whatever xva_gemm_set_fp32_products says)
    g.batch = 1; g.batch2 = 1; g.alpha = 1.f; g.beta = 1.f; g.splitk = 1; g.compute = c.h16 ? 2 : c.compute; g.mask_pad = 1; g.mask_mul = 1;
    g.a_dtype = g.b_dtype = g.c_dtype = c.dt; g.r_dtype = g.g_dtype = c.dt;
    return g;
}
// dropout descriptor of a GEMM epilogue / kernel site
struct Drop { float p; uint32_t stream; };

// Y[rows, N] = X[rows, K] W[N, K]^T (+bias) [dropout] (+R) ...   (nn.Linear)
static int linear_fwd(Ctx& c, const void* X, int64_t rows, int K, int64_t ldx, int64_t w_off, const float* bias, void* Y,
                      int N, int64_t ldy, const void* R, int64_t ldr, int mask, const int32_t* lens, int Tp, Drop dr = {0.f, 0}, int c_f32 = 0) {
    xva_gemm_params g = gp0(c);
    g.layout = XVA_GEMM_NT; g.A = X; g.B = c.wt(w_off); g.C = Y; g.M = (int)rows; g.N = N; g.K = K; g.lda = ldx; g.ldb = K; g.ldc = ldy;
    g.bias = bias; g.R = R; g.ldr = ldr; g.mask_mode = mask; g.lens = lens; g.Tp = Tp;
    g.drop_p = dr.p; g.drop_seed = c.seed; g.drop_stream = dr.stream;
    if (c_f32) g.c_dtype = XVA_F32;
    return xva_gemm(&g, c.st);
}
// dX[rows, K] = dY[rows, N] W[N, K] (+R)
static int linear_bwd_data(Ctx& c, const void* dY, int64_t rows, int N, int64_t ldy, int64_t w_off, int K, void* dX,
                           int64_t ldx, const void* R, int64_t ldr, int mask, const int32_t* lens, int Tp) {
    xva_gemm_params g = gp0(c);
    g.layout = XVA_GEMM_NN; g.A = dY; g.B = c.wt(w_off); g.C = dX; g.M = (int)rows; g.N = K; g.K = N; g.lda = ldy; g.ldb = K; g.ldc = ldx;
    g.R = R; g.ldr = ldr; g.mask_mode = mask; g.lens = lens; g.Tp = Tp;
    return xva_gemm(&g, c.st);
}
// dW[N, K] += dY[rows, N]^T X[rows, K]      (fp32 gradient)
static int linear_bwd_weight(Ctx& c, const void* dY, int64_t rows, int N, int64_t ldy, const void* X, int K, int64_t ldx, float* dW) {
    xva_gemm_params g = gp0(c);
    g.layout = XVA_GEMM_TN; g.A = dY; g.B = X; g.C = dW; g.c_dtype = XVA_F32; g.M = N; g.N = K; g.K = (int)rows; g.lda = ldy; g.ldb = ldx; g.ldc = K;
    g.accumulate = 1; g.splitk = 0;
    g.sk_ws = c.W + (c.lane == 2 ? c.pl.skws3 : (c.lane ? c.pl.skws2 : c.pl.skws)); g.sk_ws_bytes = c.pl.skws_bytes;
    return xva_gemm(&g, c.st);
}
// Long reductions into narrow outputs (the encoder's 4 864 rows x 384 columns over K = 4 608): let xva_gemm split K through this lane's slab scratch
// and apply the epilogue in the reduce pass (gemm.hip); it only does so when the tile grid would leave most of the chip idle.
static void offer_split(const Ctx& c, xva_gemm_params& g) {
    if (c.compute && g.K >= 2048 && !g.accumulate) {
        g.splitk = 0;
        g.sk_ws = c.W + (c.lane == 2 ? c.pl.skws3 : (c.lane ? c.pl.skws2 : c.pl.skws)); g.sk_ws_bytes = c.pl.skws_bytes;
    }
}
// Conv1d(k=3, pad=1) over a padded token-major sequence: Y = act(Xcat Wt^T + b) [dropout] (+R), Wt tap-major [Cout][3*Cin]
static int conv3_fwd(Ctx& c, const char* X, int64_t rows, int Cin, int64_t w_off, const float* bias, void* Y, int Cout,
                     int relu, const void* R, int mask, const int32_t* lens, int Tp, Drop dr = {0.f, 0}) {
    xva_gemm_params g = gp0(c);
    g.layout = XVA_GEMM_NT; g.A = X - (int64_t)Cin * c.es; g.B = c.wt(w_off); g.C = Y; g.M = (int)rows; g.N = Cout; g.K = 3 * Cin;
    g.lda = Cin; g.ldb = 3 * Cin; g.ldc = Cout; g.bias = bias; g.act = relu ? XVA_ACT_RELU : XVA_ACT_NONE; g.R = R; g.ldr = Cout;
    g.mask_mode = mask; g.lens = lens; g.Tp = Tp;
    g.drop_p = dr.p; g.drop_seed = c.seed; g.drop_stream = dr.stream;
    offer_split(c, g);
    return xva_gemm(&g, c.st);
}
// dX[r] = sum_j dY[r-1+j] W[:, tap 2-j, :]  (gated by Gate > 0) (+R)
// wt: a transposed, tap-reversed bf16 copy of the weight ([Cin][3][Cout], xva_fp_wt_transpose3) — the product then runs as the forward form of
// the transposed convolution (NT main loop: k-contiguous weight rows)
static int conv3_bwd_data(Ctx& c, const char* dY, int64_t rows, int Cout, int64_t w_off, int Cin, void* dX, const void* R,
                          const void* Gate, int mask, const int32_t* lens, int Tp, int accumulate, int c_dt = -1, const void* wt = nullptr) {
    xva_gemm_params g = gp0(c);
    if (c_dt >= 0) g.c_dtype = c_dt;
    if (wt) {
        g.layout = XVA_GEMM_NT; g.A = dY - (int64_t)Cout * c.es; g.B = wt; g.C = dX; g.M = (int)rows; g.N = Cin; g.K = 3 * Cout;
        g.lda = Cout; g.ldb = 3 * Cout; g.ldc = Cin;
        g.R = R; g.ldr = Cin; g.G = Gate; g.ldg = Cin; g.mask_mode = mask; g.lens = lens; g.Tp = Tp; g.accumulate = accumulate;
        offer_split(c, g);
        return xva_gemm(&g, c.st);
    }
    g.layout = XVA_GEMM_NN; g.A = dY - (int64_t)Cout * c.es; g.B = c.wt(w_off); g.C = dX; g.M = (int)rows; g.N = Cin; g.K = 3 * Cout;
    g.lda = Cout; g.ldb = 3 * Cin; g.ldc = Cin; g.seglen = Cout; g.seg0 = 2 * Cin; g.segstride = -Cin;
    g.R = R; g.ldr = Cin; g.G = Gate; g.ldg = Cin; g.mask_mode = mask; g.lens = lens; g.Tp = Tp; g.accumulate = accumulate;
    offer_split(c, g);
    return xva_gemm(&g, c.st);
}
// dWt[Cout][3*Cin] += dY^T Xcat
static int conv3_bwd_weight(Ctx& c, const void* dY, int64_t rows, int Cout, const char* X, int Cin, float* dWt) {
    xva_gemm_params g = gp0(c);
    g.layout = XVA_GEMM_TN; g.A = dY; g.B = X - (int64_t)Cin * c.es; g.C = dWt; g.c_dtype = XVA_F32; g.M = Cout; g.N = 3 * Cin; g.K = (int)rows;
    g.lda = Cout; g.ldb = Cin; g.ldc = 3 * Cin; g.accumulate = 1; g.splitk = 0;
    g.sk_ws = c.W + (c.lane == 2 ? c.pl.skws3 : (c.lane ? c.pl.skws2 : c.pl.skws)); g.sk_ws_bytes = c.pl.skws_bytes;
    return xva_gemm(&g, c.st);
}


// ------------------------------------------------------------------ split-bf16 planes (fp32 mode, split products) ----
// A sequence tensor as a split-bf16 pair (include/xva_gemm.h): hi plane rows -1 .. R (the guard rows are zeros), lo plane `plane` elements after it.
// fp16-operand mode (Ctx::h16): the same buffers and the same schedule with plane == 0 — ONE IEEE-half tensor where the pair's hi plane would be (the
// lo plane's bytes stay unused); every consumer of a pair takes plane == 0 as "single XVA_F16 tensor" (include/xva_hip.h).
static thread_local bool t_h16 = false;      // set by make_ctx: the helpers below build PlaneT descriptors without a context
struct PlaneT { char* base; int64_t plane; int C; int extra = 0; };     // extra: spare rows after row R in each plane
static inline char* prow(const PlaneT& t, int64_t row) { return t.base + (row + 1) * (int64_t)t.C * 2; }                  // hi plane, row `row`
// the pair that lives IN an fp32 sequence slot of R rows x C channels (same bytes: 2 planes x (R + 2) rows x 2 B = (R + 2) rows x 4 B)
static inline PlaneT planes_in_slot(char* row0_fp32, int64_t R, int C, int extra = 0) { return PlaneT{row0_fp32 - (int64_t)C * 4, t_h16 ? 0 : (R + 2 + extra) * (int64_t)C, C, extra}; }
static inline PlaneT planes_own(const Ctx& c, int64_t off, int64_t R) { return PlaneT{c.W + off, c.h16 ? 0 : (R + 2) * (int64_t)DM, DM}; }      // a dedicated pair buffer of R rows x DM
static bool planes_mode(const Ctx& c) { return !c.compute && c.pl.wplanes >= 0 && (c.h16 || (g_ffn_planes && xva_gemm_get_fp32_products() == 1)); }
static inline int pair_dt(const Ctx& c) { return c.h16 ? XVA_F16 : XVA_BF16; }
static inline int64_t wplane_off(const Ctx& c) { return c.h16 ? 0 : c.pl.wplane_stride; }
// per stack: the direct-to-LDS kernels want at least a K tile of rows / keys (toy sequences stay on the register-staged kernel)
// (former debugging mask, now fixed at 7) bit 0 encoder stack, bit 1 decoder stack, bit 2 the attention block (clear: feed-forward only); default 7
static const int g_planes_mask = 7;
static bool ffn_planes_on(const Ctx& c, int64_t R, int Tp) {
    const bool enc = R == c.pl.Re && Tp == c.pl.Ttp;
    return planes_mode(c) && R >= 64 && Tp >= 16 && (c.h16 || (g_planes_mask & (enc ? 1 : 2)));
}
static bool att_planes_on(const Ctx& c, int64_t R, int Tp) { return ffn_planes_on(c, R, Tp) && (c.h16 || (g_planes_mask & 4)); }
// fp32 rows -1 .. R of a sequence tensor -> the pair
static int split_rows(const Ctx& c, const char* row0_fp32, int64_t R, const PlaneT& dst, void* st) {
    return xva_split_bf16(reinterpret_cast<const float*>(row0_fp32 - (int64_t)dst.C * 4), dst.base, dst.plane, (R + 2) * (int64_t)dst.C, st);
}
// the slot held fp32 rows before (exact mode on the same workspace): the two guard rows in the MIDDLE of the pair (hi row R, lo row -1) are then stale
// (hi rows R .. R + extra and lo row -1 are adjacent.)  The pair outputs of the products never touch them: one batched launch per stack and pass.
struct GuardSpans {
    void* p[64]; int64_t n[64]; int cnt = 0;
    void add(const PlaneT& t, int64_t R) { p[cnt] = prow(t, R); n[cnt] = (int64_t)(2 + t.extra) * t.C * 2; ++cnt; }
    int zero(void* st) { return cnt ? xva_zero_spans(p, n, cnt, st) : XVA_OK; }
};
static xva_gemm_params gpp(const Ctx& c) {
    xva_gemm_params g = gp0(c);
    g.compute = 1; g.a_dtype = g.b_dtype = pair_dt(c); g.planes = c.h16 ? 0 : 1;
    g.c_dtype = g.r_dtype = g.g_dtype = XVA_F32;
    return g;
}
// the encoder's long reductions into few tiles (4 864 rows): let xva_gemm split K through this lane's slabs and apply the epilogue in the reduce pass (offer_split)
static void offer_split_p(const Ctx& c, xva_gemm_params& g) {
    if (g.K >= 1024 && !g.accumulate) {
        g.splitk = 0;
        g.sk_ws = c.W + (c.lane == 2 ? c.pl.skws3 : (c.lane ? c.pl.skws2 : c.pl.skws)); g.sk_ws_bytes = c.pl.skws_bytes;
    }
}
static const void* wplane(const Ctx& c, int64_t w_off) { return c.W + c.pl.wplanes + w_off * 2; }
static const void* wtplane(const Ctx& c, int64_t buf, const LayerP* LP, int l) {
    const ParamTable& T = table();
    const int idx = LP == T.enc ? l : NL + l;
    return c.W + buf + (int64_t)idx * DI * 3 * DM * 2;
}
constexpr int64_t WT_PLANE = (int64_t)2 * NL * DI * 3 * DM;           // elements between the hi and lo planes of the transposed weight sets
// Y (pair or fp32) = act(Xcat Wt^T + b) [dropout] (+R): conv3_fwd on pairs.  Yp: pair output (fp32 Y == nullptr) or nullptr (fp32 output Y)
static int conv3_fwd_p(Ctx& c, const PlaneT& X, int64_t rows, int Cin, int64_t w_off, const float* bias, const PlaneT* Yp, void* Y, int Cout, int relu,
                       const void* R, int mask, const int32_t* lens, int Tp, Drop dr = {0.f, 0}) {
    xva_gemm_params g = gpp(c);
    g.layout = XVA_GEMM_NT; g.A = prow(X, -1); g.a_plane = X.plane; g.B = wplane(c, w_off); g.b_plane = wplane_off(c);
    g.M = (int)rows; g.N = Cout; g.K = 3 * Cin; g.lda = Cin; g.ldb = 3 * Cin; g.ldc = Cout;
    if (Yp) { g.C = prow(*Yp, 0); g.c_dtype = pair_dt(c); g.c_plane = Yp->plane; } else { g.C = Y; g.c_dtype = XVA_F32; }
    g.bias = bias; g.act = relu ? XVA_ACT_RELU : XVA_ACT_NONE; g.R = R; g.ldr = Cout; g.r_dtype = XVA_F32;
    g.mask_mode = mask; g.lens = lens; g.Tp = Tp;
    g.drop_p = dr.p; g.drop_seed = c.seed; g.drop_stream = dr.stream;
    offer_split_p(c, g);
    return xva_gemm(&g, c.st);
}
// dX = (dY (*) W^T through the transposed pair wt) [gate] (+R): dXp pair output or fp32 dX
static int conv3_bwd_data_p(Ctx& c, const PlaneT& dY, int64_t rows, int Cout, const void* wt, int Cin, const PlaneT* dXp, void* dX, const void* R,
                            const PlaneT* gate, int mask, const int32_t* lens, int Tp) {
    xva_gemm_params g = gpp(c);
    g.layout = XVA_GEMM_NT; g.A = prow(dY, -1); g.a_plane = dY.plane; g.B = wt; g.b_plane = c.h16 ? 0 : WT_PLANE;
    g.M = (int)rows; g.N = Cin; g.K = 3 * Cout; g.lda = Cout; g.ldb = 3 * Cout; g.ldc = Cin;
    if (dXp) { g.C = prow(*dXp, 0); g.c_dtype = pair_dt(c); g.c_plane = dXp->plane; } else { g.C = dX; g.c_dtype = XVA_F32; }
    g.R = R; g.ldr = Cin; g.r_dtype = XVA_F32;
    if (gate) { g.G = prow(*gate, 0); g.ldg = Cin; g.g_dtype = pair_dt(c); }     // the hi plane: sign and zero-ness of the activation survive the rounding
    g.mask_mode = mask; g.lens = lens; g.Tp = Tp;
    offer_split_p(c, g);
    return xva_gemm(&g, c.st);
}
// dWt[Cout][3*Cin] += dY^T Xcat on pairs
static int conv3_bwd_weight_p(Ctx& c, const PlaneT& dY, int64_t rows, int Cout, const PlaneT& X, int Cin, float* dWt) {
    xva_gemm_params g = gpp(c);
    g.layout = XVA_GEMM_TN; g.A = prow(dY, 0); g.a_plane = dY.plane; g.B = prow(X, -1); g.b_plane = X.plane; g.C = dWt; g.c_dtype = XVA_F32;
    g.M = Cout; g.N = 3 * Cin; g.K = (int)rows; g.lda = Cout; g.ldb = Cin; g.ldc = 3 * Cin; g.accumulate = 1; g.splitk = 0;
    g.sk_ws = c.W + (c.lane == 2 ? c.pl.skws3 : (c.lane ? c.pl.skws2 : c.pl.skws)); g.sk_ws_bytes = c.pl.skws_bytes;
    return xva_gemm(&g, c.st);
}
// the parameter table and the transposed convolution weights as pairs (once per forward, like the bf16 mode's shadow)
// parameters [begin, end) of the table as a pair (begin a multiple of 8)
static int refresh_planes_range(const Ctx& c, const float* params, int64_t begin, int64_t end, void* st) {
    const int64_t n8 = (end - begin) / 8 * 8;
    return xva_split_bf16(params + begin, c.W + c.pl.wplanes + begin * 2, wplane_off(c), n8, st);
}
// the transposed convolution weights (read by the backward-data products only)
static int refresh_planes_wt(const Ctx& c, const float* params, void* st) {
    const ParamTable& T = table();
    int64_t so[2 * NL], dof[2 * NL];
    for (int which = 0; which < 2; ++which) {
        for (int l = 0; l < NL; ++l) {
            so[l] = which ? T.enc[l].c2_w : T.enc[l].c1_w; so[NL + l] = which ? T.dec[l].c2_w : T.dec[l].c1_w;
            dof[l] = (int64_t)l * DI * 3 * DM; dof[NL + l] = (int64_t)(NL + l) * DI * 3 * DM;
        }
        // c1: [DI][3][DM] -> [DM][3][DI] ; c2: [DM][3][DI] -> [DI][3][DM]
        XVA_TRY(xva_fp_wt_transpose3_planes(params, c.W + (which ? c.pl.wtp_c2 : c.pl.wtp_c1), so, dof, 2 * NL, which ? DM : DI, which ? DI : DM, c.h16 ? 0 : WT_PLANE, st));
    }
    return XVA_OK;
}
static int refresh_planes(const Ctx& c, const float* params, void* st) {
    if (!planes_mode(c)) return XVA_OK;
    XVA_TRY(refresh_planes_range(c, params, 0, table().total, st));
    return refresh_planes_wt(c, params, st);
}

// dropout sites: stream id = site_base + layer * 4 + {0 attention probs, 1 o_net output, 2 conv2 output}
enum { DS_ENC = 0, DS_DEC = 100, DS_PRED = 200 };


// ---- the attention block on split-bf16 pairs (fp32 mode, split products; transformer.py:100-152) -------------------------------------------------
// Every product of MultiHeadAttn — qkv projection, Q K^T, P V, o_net and their backward forms — as ONE `planes` launch each on the direct-to-LDS kernels;
// qkv, the dropped probabilities, A V and the gradients d(A V), d(scores), d(qkv) live as pairs in the bytes of their fp32 slots; the probabilities and
// d(probabilities) stay fp32 for the softmax kernels, which emit the pairs the next products read.  Products over the keys take K = Ts (the row pitch,
// a multiple of 8): columns Tp .. Ts - 1 of the probability rows are zeros.
#define XVA_HIP_TRY(x) do { if ((x) != hipSuccess) { xva_set_error("fastpitch: stream / event call failed: " #x); return XVA_ERR_HIP; } } while (0)
static void pgemm_common(xva_gemm_params& g, const PlaneT* A, const PlaneT* B, const PlaneT* Cp) {
    if (A) g.a_plane = A->plane;
    if (B) g.b_plane = B->plane;
    if (Cp) { g.c_dtype = g.a_dtype; g.c_plane = Cp->plane; }      // (the operands' 16-bit format: bf16 pair / one half tensor)
}
// fused_tail (fp16-operand mode): o_net, dropout, residual and the LayerNorm behind them as one kernel (fp_fused.hip) — the caller skips its LayerNorm launch
static int attention_fwd_planes(Ctx& c, const LayerP& p, const LayerA& a, char* x, bool x_is_split, int64_t R, int Tp, int64_t Ts, const int32_t* lens, uint32_t s0,
                                bool fused_tail = false) {
    const int B = c.pl.B;
    const PlaneT xp = planes_own(c, a.xp, R), qp = planes_in_slot(c.A(a.qkv), R, DQKV, QKV_SPARE), avp = planes_in_slot(c.A(a.av), R, DH);
    const PlaneT wq{nullptr, wplane_off(c), 0};
    const PlaneT pdp{c.A(a.Pd), c.h16 ? 0 : (int64_t)B * Tp * Ts, 0};
    if (!x_is_split) XVA_TRY(split_rows(c, x, R, xp, c.st));       // (layers past the first: the previous layer's LayerNorm wrote the pair)
    {   // qkv = x Wqkv^T + b, stored as a pair
        xva_gemm_params g = gpp(c); pgemm_common(g, &xp, &wq, &qp);
        g.layout = XVA_GEMM_NT; g.A = prow(xp, 0); g.B = wplane(c, p.qkv_w); g.C = prow(qp, 0); g.M = (int)R; g.N = DQKV; g.K = DM; g.lda = DM; g.ldb = DM; g.ldc = DQKV;
        g.bias = c.P + p.qkv_b;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    if (g_att_flash) {   // softmax(scale Q K^T) V without the T x T tensors; the logsumexp rows sit at the start of the (unused) probability slot
        XVA_TRY(xva_fp_attention_fwd_pairs(prow(qp, 0), qp.plane, lens, prow(avp, 0), avp.plane, reinterpret_cast<float*>(c.A(a.P)), B, Tp, 0.125f, c.pd, c.seed,
                                           s0 + 0, c.st));
    } else {
    {   // S = scale * Q K^T per item (fp32)
        xva_gemm_params g = gpp(c); pgemm_common(g, &qp, &qp, nullptr);
        // (N = Ts, the row pitch: the two to seven extra score columns — keys past the item, finite values the softmax never reads — buy 16-byte row stores)
        g.layout = XVA_GEMM_NT; g.A = prow(qp, 0); g.B = prow(qp, 0) + DH * 2; g.C = c.A(a.P); g.c_dtype = XVA_F32; g.M = Tp; g.N = (int)Ts; g.K = DH;
        g.lda = DQKV; g.ldb = DQKV; g.ldc = Ts; g.batch = B; g.sA = (int64_t)Tp * DQKV; g.sB = g.sA; g.sC = (int64_t)Tp * Ts; g.alpha = 0.125f;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    XVA_TRY(xva_fp_softmax_fwd_pairs(c.A(a.P), pdp.base, pdp.plane, lens, B, Tp, Ts, c.pd, c.seed, s0 + 0, c.st));
    {   // AV = dropatt(P) V, stored as a pair
        xva_gemm_params g = gpp(c); pgemm_common(g, &pdp, &qp, &avp);
        g.layout = XVA_GEMM_NN; g.A = pdp.base; g.B = prow(qp, 0) + 2 * DH * 2; g.C = prow(avp, 0); g.M = Tp; g.N = DH; g.K = (int)Ts;
        g.lda = Ts; g.ldb = DQKV; g.ldc = DH; g.batch = B; g.sA = (int64_t)Tp * Ts; g.sB = (int64_t)Tp * DQKV; g.sC = (int64_t)Tp * DH;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    }
    if (fused_tail) {   // sum1 = x + drop(AV Wo^T) ; y1 = LN(sum1) * mask, fp32 and as the half copy conv1 reads
        const PlaneT yp = planes_own(c, a.yp, R);
        return xva_fp_onet_ln_fwd_f16(prow(avp, 0), wplane(c, p.o_w), reinterpret_cast<const float*>(x), c.P + p.ln1_g, c.P + p.ln1_b, reinterpret_cast<float*>(c.A(a.sum1)),
                                      reinterpret_cast<float*>(c.A(a.y1)), prow(yp, 0), c.F(a.mean1), c.F(a.rstd1), R, XVA_MASK_LEN, lens, Tp, c.pd, c.seed, s0 + 1, c.st);
    }
    {   // sum1 = x + drop(AV Wo^T) (fp32)
        xva_gemm_params g = gpp(c); pgemm_common(g, &avp, &wq, nullptr);
        g.layout = XVA_GEMM_NT; g.A = prow(avp, 0); g.B = wplane(c, p.o_w); g.C = c.A(a.sum1); g.c_dtype = XVA_F32; g.M = (int)R; g.N = DM; g.K = DH; g.lda = DH; g.ldb = DH; g.ldc = DM;
        g.R = x; g.ldr = DM; g.r_dtype = XVA_F32; g.drop_p = c.pd; g.drop_seed = c.seed; g.drop_stream = s0 + 1;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    return XVA_OK;
}
// backward of the same block: gDm = d(sum1) masked by o_net's dropout (fp32) -> gA = gD + d x through the block (fp32, LEN-masked).  Leaves the pairs the
// weight gradients read (dp, xp of this layer parity; gQKV's slot) in place.
static int attention_bwd_planes(Ctx& c, const LayerP& p, const LayerA& a, char* gDm, bool gDm_is_split, char* gD, char* gAV, char* gP, char* gQKV, char* gA, int stk,
                                int par, int64_t R, int Tp, int64_t Ts, const int32_t* lens, uint32_t s0) {
    const int B = c.pl.B;
    const int64_t Rmax = c.pl.Rd > c.pl.Re ? c.pl.Rd : c.pl.Re;
    const PlaneT dpp = planes_own(c, c.pl.dp[stk][par], R);
    const PlaneT qp = planes_in_slot(c.A(a.qkv), R, DQKV, QKV_SPARE), avp = planes_in_slot(c.A(a.av), R, DH);
    const PlaneT gavp = planes_in_slot(gAV, Rmax, DH), gqp = planes_in_slot(gQKV, Rmax, DQKV);
    const PlaneT wq{nullptr, wplane_off(c), 0};
    const PlaneT pdp{c.A(a.Pd), c.h16 ? 0 : (int64_t)B * Tp * Ts, 0}, dsp{c.A(c.pl.gPp), c.h16 ? 0 : (int64_t)B * Tp * Ts, 0};
    if (!gDm_is_split) XVA_TRY(split_rows(c, gDm, R, dpp, c.st));
    {   // gAV = gDm Wo
        xva_gemm_params g = gpp(c); pgemm_common(g, &dpp, &wq, &gavp);
        g.layout = XVA_GEMM_NN; g.A = prow(dpp, 0); g.B = wplane(c, p.o_w); g.C = prow(gavp, 0); g.M = (int)R; g.N = DH; g.K = DM; g.lda = DM; g.ldb = DH; g.ldc = DH;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    if (g_att_flash) {   // d(qkv) pair from the pairs of qkv, A V, d(A V) and the saved logsumexp rows (D = rowsum(dO . O) in gP's bytes)
        XVA_TRY(xva_fp_attention_bwd_pairs(prow(qp, 0), qp.plane, prow(avp, 0), avp.plane, prow(gavp, 0), gavp.plane, reinterpret_cast<const float*>(c.A(a.P)),
                                           reinterpret_cast<float*>(gP), lens, prow(gqp, 0), gqp.plane, B, Tp, 0.125f, c.pd, c.seed, s0 + 0, c.st));
    } else {
    {   // dPd = gAV V^T (fp32)
        xva_gemm_params g = gpp(c); pgemm_common(g, &gavp, &qp, nullptr);
        g.layout = XVA_GEMM_NT; g.A = prow(gavp, 0); g.B = prow(qp, 0) + 2 * DH * 2; g.C = gP; g.c_dtype = XVA_F32; g.M = Tp; g.N = (int)Ts; g.K = DH;   // N = Ts: see the forward
        g.lda = DH; g.ldb = DQKV; g.ldc = Ts; g.batch = B; g.sA = (int64_t)Tp * DH; g.sB = (int64_t)Tp * DQKV; g.sC = (int64_t)Tp * Ts;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    {   // dV = Pd^T gAV -> gQKV[:, 128:192]
        xva_gemm_params g = gpp(c); pgemm_common(g, &pdp, &gavp, &gqp);
        g.layout = XVA_GEMM_TN; g.A = pdp.base; g.B = prow(gavp, 0); g.C = prow(gqp, 0) + 2 * DH * 2; g.M = Tp; g.N = DH; g.K = Tp;
        g.lda = Ts; g.ldb = DH; g.ldc = DQKV; g.batch = B; g.sA = (int64_t)Tp * Ts; g.sB = (int64_t)Tp * DH; g.sC = (int64_t)Tp * DQKV;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    XVA_TRY(xva_fp_softmax_bwd_pairs(c.A(a.P), gP, dsp.base, dsp.plane, B, Tp, Ts, 0.125f, c.pd, c.seed, s0 + 0, c.st));      // dS (incl. 1/sqrt(d)) as a pair
    {   // dQ = dS K -> gQKV[:, 0:64]
        xva_gemm_params g = gpp(c); pgemm_common(g, &dsp, &qp, &gqp);
        g.layout = XVA_GEMM_NN; g.A = dsp.base; g.B = prow(qp, 0) + DH * 2; g.C = prow(gqp, 0); g.M = Tp; g.N = DH; g.K = (int)Ts;
        g.lda = Ts; g.ldb = DQKV; g.ldc = DQKV; g.batch = B; g.sA = (int64_t)Tp * Ts; g.sB = (int64_t)Tp * DQKV; g.sC = (int64_t)Tp * DQKV;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    {   // dK = dS^T Q -> gQKV[:, 64:128]
        xva_gemm_params g = gpp(c); pgemm_common(g, &dsp, &qp, &gqp);
        g.layout = XVA_GEMM_TN; g.A = dsp.base; g.B = prow(qp, 0); g.C = prow(gqp, 0) + DH * 2; g.M = Tp; g.N = DH; g.K = Tp;
        g.lda = Ts; g.ldb = DQKV; g.ldc = DQKV; g.batch = B; g.sA = (int64_t)Tp * Ts; g.sB = (int64_t)Tp * DQKV; g.sC = (int64_t)Tp * DQKV;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    }
    {   // d x = gD + gQKV Wqkv, LEN-masked -> gA (fp32)
        xva_gemm_params g = gpp(c); pgemm_common(g, &gqp, &wq, nullptr);
        g.layout = XVA_GEMM_NN; g.A = prow(gqp, 0); g.B = wplane(c, p.qkv_w); g.C = gA; g.c_dtype = XVA_F32; g.M = (int)R; g.N = DM; g.K = DQKV; g.lda = DQKV; g.ldb = DM; g.ldc = DM;
        g.R = gD; g.ldr = DM; g.r_dtype = XVA_F32; g.mask_mode = XVA_MASK_LEN; g.lens = lens; g.Tp = Tp;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    return XVA_OK;
}
// the block's weight gradients and the qkv bias sums from the pairs attention_bwd_planes left (issued to cw's lane)
static int attention_wgrad_planes(Ctx& cw, const LayerP& p, const LayerA& a, char* gQKV, int stk, int par, int64_t R, float* Gg) {
    const Ctx& c = cw;
    const int64_t Rmax = c.pl.Rd > c.pl.Re ? c.pl.Rd : c.pl.Re;
    const PlaneT dpp = planes_own(c, c.pl.dp[stk][par], R), xp = planes_own(c, a.xp, R);      // xp: the forward's pair of the layer input
    const PlaneT avp = planes_in_slot(c.A(a.av), R, DH), gqp = planes_in_slot(gQKV, Rmax, DQKV);
    auto wgrad = [&](const PlaneT& dY, int M, const PlaneT& X, int N, float* dW) {
        xva_gemm_params g = gpp(cw); pgemm_common(g, &dY, &X, nullptr);
        g.layout = XVA_GEMM_TN; g.A = prow(dY, 0); g.B = prow(X, 0); g.C = dW; g.c_dtype = XVA_F32; g.M = M; g.N = N; g.K = (int)R; g.lda = M; g.ldb = N; g.ldc = N;
        g.accumulate = 1; g.splitk = 0;
        g.sk_ws = cw.W + (cw.lane == 2 ? cw.pl.skws3 : (cw.lane ? cw.pl.skws2 : cw.pl.skws)); g.sk_ws_bytes = cw.pl.skws_bytes;
        return xva_gemm(&g, cw.st);
    };
    XVA_TRY(wgrad(dpp, DM, avp, DH, Gg + p.o_w));
    XVA_TRY(wgrad(gqp, DQKV, xp, DM, Gg + p.qkv_w));
    XVA_TRY(xva_fp_colsum(prow(gqp, 0), pair_dt(c), Gg + p.qkv_b, R, DQKV, DQKV, cw.st));
    if (!c.h16) XVA_TRY(xva_fp_colsum(prow(gqp, 0) + gqp.plane * 2, XVA_BF16, Gg + p.qkv_b, R, DQKV, DQKV, cw.st));
    return XVA_OK;
}

// ------------------------------------------------------------------ transformer stack ----
static int layers_fwd(Ctx& c, const LayerP* LP, const LayerA* LA, const int64_t* xo, int64_t R, int Tp, int64_t Ts,
                      const int32_t* lens, int site) {
    const int B = c.pl.B;
    const bool att_planes = att_planes_on(c, R, Tp), ffn_planes = ffn_planes_on(c, R, Tp);
    if (att_planes || ffn_planes) {      // the pairs of qkv, A V and h live in their fp32 slots: stale guard rows if the exact mode ran on this workspace
        GuardSpans gs;
        for (int l = 0; l < NL; ++l) {
            if (att_planes) { gs.add(planes_in_slot(c.A(LA[l].qkv), R, DQKV, QKV_SPARE), R); gs.add(planes_in_slot(c.A(LA[l].av), R, DH), R); }
            if (ffn_planes) gs.add(planes_in_slot(c.A(LA[l].h), R, DI), R);
        }
        XVA_TRY(gs.zero(c.st));
    }
    for (int l = 0; l < NL; ++l) {
        const LayerP& p = LP[l];
        const LayerA& a = LA[l];
        char* x = c.A(xo[l]);
        char* qkv = c.A(a.qkv); char* av = c.A(a.av);
        const uint32_t s0 = site + l * 4;
        const bool fused_tail = att_planes && ffn_planes && c.h16 && g_ln_pairs && g_onet_fused;
        if (att_planes) {
            XVA_TRY(attention_fwd_planes(c, p, a, x, g_ln_pairs && l > 0, R, Tp, Ts, lens, s0, fused_tail));
            if (fused_tail) goto ffn;
        } else {
        // qkv = x Wqkv^T + b                                             (transformer.py:109)
        XVA_TRY(linear_fwd(c, x, R, DM, DM, p.qkv_w, c.P + p.qkv_b, qkv, DQKV, DQKV, nullptr, 0, XVA_MASK_NONE, nullptr, 0));
        if (c.compute) {
            XVA_TRY(xva_fp_attention_fwd(qkv, lens, av, c.F(a.lse), B, Tp, 0.125f, c.pd, c.seed, s0 + 0, c.st));   // transformer.py:118-130
        } else {
            char* Pm = c.A(a.P); char* Pdm = c.pd > 0.f ? c.A(a.Pd) : c.A(a.P);
            {   // S = scale * Q K^T per item                                  (transformer.py:118-119)
                xva_gemm_params g = gp0(c);
                g.layout = XVA_GEMM_NT; g.A = qkv; g.B = c.sh(qkv, DH); g.C = Pm; g.M = Tp; g.N = Tp; g.K = DH;
                g.lda = DQKV; g.ldb = DQKV; g.ldc = Ts; g.batch = B; g.sA = (int64_t)Tp * DQKV; g.sB = g.sA; g.sC = (int64_t)Tp * Ts;
                g.alpha = 0.125f;
                XVA_TRY(xva_gemm(&g, c.st));
            }
            XVA_TRY(xva_fp_softmax_fwd(Pm, Pdm, c.dt, lens, B, Tp, Ts, c.pd, c.seed, s0 + 0, c.st));      // :121-128 (softmax, dropatt)
            {   // AV = dropatt(P) V                                            (transformer.py:130)
                xva_gemm_params g = gp0(c);
                g.layout = XVA_GEMM_NN; g.A = Pdm; g.B = c.sh(qkv, 2 * DH); g.C = av; g.M = Tp; g.N = DH; g.K = Tp;
                g.lda = Ts; g.ldb = DQKV; g.ldc = DH; g.batch = B; g.sA = (int64_t)Tp * Ts; g.sB = (int64_t)Tp * DQKV; g.sC = (int64_t)Tp * DH;
                XVA_TRY(xva_gemm(&g, c.st));
            }
        }
        // sum1 = x + drop(AV Wo^T) ; y1 = LN(sum1) * mask                  (transformer.py:137-146,166-167)
        if (c.compute && g_onet_fused) {   // bf16 mode: projection, dropout, residual and LayerNorm in one kernel (fp_fused.hip)
            XVA_TRY(xva_fp_onet_ln_fwd(av, c.wt(p.o_w), x, c.P + p.ln1_g, c.P + p.ln1_b, c.A(a.sum1), c.A(a.y1), c.F(a.mean1), c.F(a.rstd1), R, XVA_MASK_LEN, lens, Tp,
                                       c.pd, c.seed, s0 + 1, c.st));
            goto ffn;
        }
        XVA_TRY(linear_fwd(c, av, R, DH, DH, p.o_w, nullptr, c.A(a.sum1), DM, DM, x, DM, XVA_MASK_NONE, nullptr, 0, Drop{c.pd, s0 + 1}));
        }
        if (ffn_planes && g_ln_pairs) {        // y1 fp32 (the residual of conv2, LayerNorm backward) and as the pair conv1 reads
            const PlaneT yp = planes_own(c, a.yp, R);
            XVA_TRY(xva_fp_layernorm_fwd_pair(c.A(a.sum1), c.P + p.ln1_g, c.P + p.ln1_b, c.A(a.y1), prow(yp, 0), yp.plane, c.F(a.mean1), c.F(a.rstd1), R, DM,
                                              XVA_MASK_LEN, lens, Tp, c.st));
        } else
        XVA_TRY(xva_fp_layernorm_fwd(c.A(a.sum1), c.P + p.ln1_g, c.P + p.ln1_b, c.A(a.y1), c.dt, c.F(a.mean1), c.F(a.rstd1), R, DM,
                                     XVA_MASK_LEN, lens, Tp, 0.f, 0, 0, c.st));
    ffn:
        // h = relu(conv1(y1)) ; sum2 = y1 + drop(conv2(h)) ; x' = LN(sum2) * mask  (transformer.py:59-77,168-170)
        if (ffn_planes) {     // fp32 mode, split products: both convolutions on split-bf16 pairs (h is stored as a pair in its fp32 slot)
            const PlaneT yp = planes_own(c, a.yp, R), hp = planes_in_slot(c.A(a.h), R, DI);
            if (!g_ln_pairs) XVA_TRY(split_rows(c, c.A(a.y1), R, yp, c.st));
            XVA_TRY(conv3_fwd_p(c, yp, R, DM, p.c1_w, c.P + p.c1_b, &hp, nullptr, DI, 1, nullptr, XVA_MASK_PAD, lens, Tp));
            XVA_TRY(conv3_fwd_p(c, hp, R, DI, p.c2_w, c.P + p.c2_b, nullptr, c.A(a.sum2), DM, 0, c.A(a.y1), XVA_MASK_NONE, nullptr, 0, Drop{c.pd, s0 + 2}));
        } else {
        XVA_TRY(conv3_fwd(c, c.A(a.y1), R, DM, p.c1_w, c.P + p.c1_b, c.A(a.h), DI, 1, nullptr, XVA_MASK_PAD, lens, Tp));
        XVA_TRY(conv3_fwd(c, c.A(a.h), R, DI, p.c2_w, c.P + p.c2_b, c.A(a.sum2), DM, 0, c.A(a.y1), XVA_MASK_NONE, nullptr, 0, Drop{c.pd, s0 + 2}));
        }
        if (att_planes && g_ln_pairs && l + 1 < NL) {       // the next layer's input, also as the pair its qkv projection (and that weight gradient) reads
            const PlaneT xn = planes_own(c, LA[l + 1].xp, R);
            XVA_TRY(xva_fp_layernorm_fwd_pair(c.A(a.sum2), c.P + p.ln2_g, c.P + p.ln2_b, c.A(xo[l + 1]), prow(xn, 0), xn.plane, c.F(a.mean2), c.F(a.rstd2), R, DM,
                                              XVA_MASK_LEN, lens, Tp, c.st));
        } else
        XVA_TRY(xva_fp_layernorm_fwd(c.A(a.sum2), c.P + p.ln2_g, c.P + p.ln2_b, c.A(xo[l + 1]), c.dt, c.F(a.mean2), c.F(a.rstd2), R, DM,
                                     XVA_MASK_LEN, lens, Tp, 0.f, 0, 0, c.st));
    }
    return XVA_OK;
}

// Backward through the 6 layers.  On entry gA holds dL/d(x_out) ; on exit gA holds dL/d(x_in) (LEN-masked).
// ---- weight-gradient lane ----------------------------------------------------------------------------------------------------------
// Inside a layer's backward the four weight gradients (and three bias column sums) depend on the data-gradient chain but nothing
// depends on them until the optimizer.  Issued on the same stream they sit between the chain's kernels: every GEMM's last, partly
// filled round of workgroups and every small kernel's latency leaves CUs idle.  Here the chain stays on the caller's stream and the
// weight gradients of layer l go to a side stream behind an event recorded at the end of the chain's layer l; the tensors they read
// (gB / gBm, gD / gDm, gH, gQKV) alternate between two sets by layer parity, and the chain waits for the side stream's layer l + 2
// before it overwrites a set.  The side stream and its events are created once per host thread; every call joins before returning,
// and a gradient bucket's event (DP overlap) is recorded on the side stream, i.e. after both lanes finished the layer.
// A second side stream carries the temporal predictors (model.py:394-418): in training the decoder is conditioned on the TARGET pitch /
// energy, so neither its forward nor its backward depends on them; their small, latency-bound kernels run under the decoder's GEMMs.
// env XVA_FP_STREAMS=1 keeps everything on the caller's stream.
// Side-lane stream priority (round 2 experiment, knob removed): default priority (kept), lowest, highest.  Measured (FastPitch / HiFi-GAN ms
// per step): default 10.38 / 38.4, lowest 10.49 / 45.9 (the lanes starve: HiFi-GAN falls back to its one-stream time), highest 10.47 / 55.0 (the caller's chain starves).
static hipError_t xva_create_lane_stream(hipStream_t* s) {
    static const int mode = 0;      // (measured in round 2: lowest / highest priority lanes starve one side; the knob is gone)
    int least = 0, greatest = 0;
    if (mode != 0 && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least != greatest)
        return hipStreamCreateWithPriority(s, hipStreamNonBlocking, mode == 1 ? least : greatest);
    return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
}
struct WgLane { hipStream_t s = nullptr, sp = nullptr; hipEvent_t fork = nullptr, chain[NL] = {}, done[NL] = {}, pfork = nullptr, pmid = nullptr, pjoin = nullptr;
                bool init = false, ok = false; };
static WgLane& wg_lane() {
    static thread_local WgLane r;
    if (!r.init) {
        r.init = true;
        const char* e = getenv("XVA_FP_STREAMS");
        bool ok = !(e && atoi(e) == 1) && xva_create_lane_stream(&r.s) == hipSuccess &&
                  xva_create_lane_stream(&r.sp) == hipSuccess &&
                  hipEventCreateWithFlags(&r.fork, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&r.pfork, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&r.pmid, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&r.pjoin, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; ok && i < NL; ++i)
            ok = hipEventCreateWithFlags(&r.chain[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&r.done[i], hipEventDisableTiming) == hipSuccess;
        r.ok = ok;
    }
    return r;
}
static int g_fp_serial = 0;     // xva_fp_set_streams(1): everything on the caller's stream (per-kernel measurements)

static int layers_bwd(Ctx& c, const LayerP* LP, const LayerA* LA, const int64_t* xo, int64_t R, int Tp, int64_t Ts,
                      const int32_t* lens, bool last_bucket_deferred, int site) {
    const int B = c.pl.B;
    char *gA = c.A(c.pl.gA), *gB = c.A(c.pl.gB), *gBm = c.A(c.pl.gBm), *gC = c.A(c.pl.gC), *gD = c.A(c.pl.gD), *gDm = c.A(c.pl.gDm),
         *gH = c.A(c.pl.gH), *gAV = c.A(c.pl.gAV), *gP = c.A(c.pl.gP), *gQKV = c.A(c.pl.gQKV);
    const bool drop = c.pd > 0.f;
    float* Gg = c.G;
    WgLane& wl = wg_lane();
    const bool two = wl.ok && !g_fp_serial;
    Ctx cw = c;                                   // the weight-gradient lane
    if (two) {
        cw.st = wl.s; cw.lane = 1;
        XVA_HIP_TRY(hipEventRecord(wl.fork, (hipStream_t)c.st));
        XVA_HIP_TRY(hipStreamWaitEvent(wl.s, wl.fork, 0));
    }
    char* const setB[2] = {gB, c.A(c.pl.gB2)}; char* const setBm[2] = {gBm, c.A(c.pl.gBm2)}; char* const setD[2] = {gD, c.A(c.pl.gD2)};
    char* const setDm[2] = {gDm, c.A(c.pl.gDm2)}; char* const setH[2] = {gH, c.A(c.pl.gH2)}; char* const setQ[2] = {gQKV, c.A(c.pl.gQKV2)};
    if (ffn_planes_on(c, R, Tp)) {       // d(h), d(A V), d(qkv) as pairs in their (shared, Rmax-row) fp32 slots: the guard rows, once per pass
        const int64_t Rm = c.pl.Rd > c.pl.Re ? c.pl.Rd : c.pl.Re;
        GuardSpans gs;
        for (int q = 0; q < 2; ++q) { gs.add(planes_in_slot(setH[q], Rm, DI), Rm); gs.add(planes_in_slot(setQ[q], Rm, DQKV), Rm); }
        gs.add(planes_in_slot(gAV, Rm, DH), Rm);
        XVA_TRY(gs.zero(c.st));
    }
    for (int l = NL - 1; l >= 0; --l) {
        const LayerP& p = LP[l];
        const LayerA& a = LA[l];
        char* x = c.A(xo[l]);
        char* qkv = c.A(a.qkv); char* av = c.A(a.av);
        const uint32_t s0 = site + l * 4;
        const int par = two ? (l & 1) : 0;
        const int stk = LP == table().enc ? 0 : 1;
        gB = setB[par]; gBm = setBm[par]; gD = setD[par]; gDm = setDm[par]; gH = setH[par]; gQKV = setQ[par];
        if (two && l + 2 < NL) XVA_HIP_TRY(hipStreamWaitEvent((hipStream_t)c.st, wl.done[l + 2], 0));   // this set's last readers
        // LN2 backward -> gB = d sum2 (residual path) ; gBm = gB * dropmask (conv2 branch)
        const bool planes = ffn_planes_on(c, R, Tp), att_planes = att_planes_on(c, R, Tp);
        const PlaneT gBp = planes_own(c, c.pl.gp[stk][par], R), y1p = planes_own(c, a.yp, R);        // (planes mode only)
        if (planes && g_ln_pairs)
            // (fp16-operand mode: the dropout-masked copy exists as the half tensor only — its fp32 twin had one reader, the bias column sum, which reads the half tensor too)
            XVA_TRY(xva_fp_layernorm_bwd_pair(gA, c.A(a.sum2), c.F(a.mean2), c.F(a.rstd2), c.P + p.ln2_g, gB, (drop && !c.h16) ? gBm : nullptr, prow(gBp, 0), gBp.plane, Gg + p.ln2_g,
                                              Gg + p.ln2_b, R, DM, XVA_MASK_LEN, lens, Tp, c.pd, c.seed, s0 + 2, c.st));
        else
        XVA_TRY(xva_fp_layernorm_bwd(gA, c.A(a.sum2), c.F(a.mean2), c.F(a.rstd2), c.P + p.ln2_g, gB, drop ? gBm : nullptr, c.dt, Gg + p.ln2_g,
                                     Gg + p.ln2_b, R, DM, XVA_MASK_LEN, lens, Tp, 0, 0.f, 0, 0, c.pd, c.seed, s0 + 2, nullptr, nullptr, c.st));
        // conv2 backward: gH = (gBm (*) W2) * [h > 0], structural rows zero
        const int64_t Rmax = c.pl.Rd > c.pl.Re ? c.pl.Rd : c.pl.Re;
        const PlaneT gHp = planes_in_slot(gH, Rmax, DI), hp = planes_in_slot(c.A(a.h), R, DI);
        if (planes) {               // fp32 mode, split products: d(sum2) and y1 as pairs; gH lives as a pair in its fp32 slot
            if (!g_ln_pairs) XVA_TRY(split_rows(c, gBm, R, gBp, c.st));          // (y1p: the forward's pair)
            XVA_TRY(conv3_bwd_data_p(c, gBp, R, DM, wtplane(c, c.pl.wtp_c2, LP, l), DI, &gHp, nullptr, nullptr, &hp, XVA_MASK_PAD, lens, Tp));
            XVA_TRY(conv3_bwd_data_p(c, gHp, R, DI, wtplane(c, c.pl.wtp_c1, LP, l), DM, nullptr, gC, gB, nullptr, XVA_MASK_LEN, lens, Tp));
        } else {
        XVA_TRY(conv3_bwd_data(c, gBm, R, DM, p.c2_w, DI, gH, nullptr, c.A(a.h), XVA_MASK_PAD, lens, Tp, 0, -1, c.wt_c2(LP, l)));
        // conv1 backward + residual: gC = gB + gH (*) W1, LEN-masked (y1 was multiplied by mask)
        XVA_TRY(conv3_bwd_data(c, gH, R, DI, p.c1_w, DM, gC, gB, nullptr, XVA_MASK_LEN, lens, Tp, 0));
        }
        // LN1 backward -> gD = d sum1 ; gDm = gD * dropmask (o_net branch)
        if (att_planes && g_ln_pairs) {
            const PlaneT dpp = planes_own(c, c.pl.dp[stk][par], R);
            XVA_TRY(xva_fp_layernorm_bwd_pair(gC, c.A(a.sum1), c.F(a.mean1), c.F(a.rstd1), c.P + p.ln1_g, gD, (drop && !c.h16) ? gDm : nullptr, prow(dpp, 0), dpp.plane, Gg + p.ln1_g,
                                              Gg + p.ln1_b, R, DM, XVA_MASK_LEN, lens, Tp, c.pd, c.seed, s0 + 1, c.st));
        } else
        XVA_TRY(xva_fp_layernorm_bwd(gC, c.A(a.sum1), c.F(a.mean1), c.F(a.rstd1), c.P + p.ln1_g, gD, drop ? gDm : nullptr, c.dt, Gg + p.ln1_g,
                                     Gg + p.ln1_b, R, DM, XVA_MASK_LEN, lens, Tp, 0, 0.f, 0, 0, c.pd, c.seed, s0 + 1, nullptr, nullptr, c.st));
        if (att_planes) {           // the attention block on split-bf16 pairs (qkv, the dropped probabilities and A V are pairs since the forward pass)
            XVA_TRY(attention_bwd_planes(c, p, a, gDm, g_ln_pairs != 0, gD, gAV, gP, gQKV, gA, stk, par, R, Tp, Ts, lens, s0));
        } else {
        // o_net backward
        XVA_TRY(linear_bwd_data(c, gDm, R, DM, DM, p.o_w, DH, gAV, DH, nullptr, 0, XVA_MASK_NONE, nullptr, 0));
        if (c.compute) {
            XVA_TRY(xva_fp_attention_bwd(qkv, av, gAV, c.F(a.lse), (float*)gP, lens, gQKV, B, Tp, 0.125f, c.pd, c.seed, s0 + 0, c.st));
        } else {
            char* Pm = c.A(a.P); char* Pdm = c.pd > 0.f ? c.A(a.Pd) : c.A(a.P);
            {   // dPd = dAV V^T
                xva_gemm_params g = gp0(c);
                g.layout = XVA_GEMM_NT; g.A = gAV; g.B = c.sh(qkv, 2 * DH); g.C = gP; g.M = Tp; g.N = Tp; g.K = DH;
                g.lda = DH; g.ldb = DQKV; g.ldc = Ts; g.batch = B; g.sA = (int64_t)Tp * DH; g.sB = (int64_t)Tp * DQKV; g.sC = (int64_t)Tp * Ts;
                XVA_TRY(xva_gemm(&g, c.st));
            }
            {   // dV = Pd^T dAV  -> gQKV[:, 128:192]
                xva_gemm_params g = gp0(c);
                g.layout = XVA_GEMM_TN; g.A = Pdm; g.B = gAV; g.C = c.sh(gQKV, 2 * DH); g.M = Tp; g.N = DH; g.K = Tp;
                g.lda = Ts; g.ldb = DH; g.ldc = DQKV; g.batch = B; g.sA = (int64_t)Tp * Ts; g.sB = (int64_t)Tp * DH; g.sC = (int64_t)Tp * DQKV;
                XVA_TRY(xva_gemm(&g, c.st));
            }
            XVA_TRY(xva_fp_softmax_bwd(Pm, gP, c.dt, B, Tp, Ts, 0.125f, c.pd, c.seed, s0 + 0, c.st));    // gP = dS (incl. 1/sqrt(d))
            {   // dQ = dS K -> gQKV[:, 0:64]
                xva_gemm_params g = gp0(c);
                g.layout = XVA_GEMM_NN; g.A = gP; g.B = c.sh(qkv, DH); g.C = gQKV; g.M = Tp; g.N = DH; g.K = Tp;
                g.lda = Ts; g.ldb = DQKV; g.ldc = DQKV; g.batch = B; g.sA = (int64_t)Tp * Ts; g.sB = (int64_t)Tp * DQKV; g.sC = (int64_t)Tp * DQKV;
                XVA_TRY(xva_gemm(&g, c.st));
            }
            {   // dK = dS^T Q -> gQKV[:, 64:128]
                xva_gemm_params g = gp0(c);
                g.layout = XVA_GEMM_TN; g.A = gP; g.B = qkv; g.C = c.sh(gQKV, DH); g.M = Tp; g.N = DH; g.K = Tp;
                g.lda = Ts; g.ldb = DQKV; g.ldc = DQKV; g.batch = B; g.sA = (int64_t)Tp * Ts; g.sB = (int64_t)Tp * DQKV; g.sC = (int64_t)Tp * DQKV;
                XVA_TRY(xva_gemm(&g, c.st));
            }
        }
        // d x = gD + gQKV Wqkv, LEN-masked -> gA
        XVA_TRY(linear_bwd_data(c, gQKV, R, DQKV, DQKV, p.qkv_w, DM, gA, DM, gD, DM, XVA_MASK_LEN, lens, Tp));
        }
        // weight gradients and bias sums of this layer (models' transformer.py:21-147 parameters), on the side lane when there is one
        if (two) {
            XVA_HIP_TRY(hipEventRecord(wl.chain[l], (hipStream_t)c.st));
            XVA_HIP_TRY(hipStreamWaitEvent(wl.s, wl.chain[l], 0));
        }
        if (planes) {
            XVA_TRY(conv3_bwd_weight_p(cw, gBp, R, DM, hp, DI, Gg + p.c2_w));
            if (c.h16 && g_ln_pairs) XVA_TRY(xva_fp_colsum(prow(gBp, 0), XVA_F16, Gg + p.c2_b, R, DM, DM, cw.st));
            else XVA_TRY(xva_fp_colsum(gBm, c.dt, Gg + p.c2_b, R, DM, DM, cw.st));
            XVA_TRY(conv3_bwd_weight_p(cw, gHp, R, DI, y1p, DM, Gg + p.c1_w));
            XVA_TRY(xva_fp_colsum(prow(gHp, 0), pair_dt(c), Gg + p.c1_b, R, DI, DI, cw.st));                       // d b1 = column sums of hi + lo
            if (!c.h16) XVA_TRY(xva_fp_colsum(prow(gHp, 0) + gHp.plane * 2, XVA_BF16, Gg + p.c1_b, R, DI, DI, cw.st));
        } else {
        XVA_TRY(conv3_bwd_weight(cw, gBm, R, DM, c.A(a.h), DI, Gg + p.c2_w));
        XVA_TRY(xva_fp_colsum(gBm, c.dt, Gg + p.c2_b, R, DM, DM, cw.st));
        XVA_TRY(conv3_bwd_weight(cw, gH, R, DI, c.A(a.y1), DM, Gg + p.c1_w));
        XVA_TRY(xva_fp_colsum(gH, c.dt, Gg + p.c1_b, R, DI, DI, cw.st));
        }
        if (att_planes) {
            XVA_TRY(attention_wgrad_planes(cw, p, a, gQKV, stk, par, R, Gg));
        } else {
        XVA_TRY(linear_bwd_weight(cw, gDm, R, DM, DM, av, DH, DH, Gg + p.o_w));
        XVA_TRY(linear_bwd_weight(cw, gQKV, R, DQKV, DQKV, x, DM, DM, Gg + p.qkv_w));
        XVA_TRY(xva_fp_colsum(gQKV, c.dt, Gg + p.qkv_b, R, DQKV, DQKV, cw.st));
        }
        if (two) XVA_HIP_TRY(hipEventRecord(wl.done[l], wl.s));
        if (l > 0 || !last_bucket_deferred) XVA_TRY(cw.record(c.ev_base + (NL - 1 - l)));
    }
    if (two) XVA_HIP_TRY(hipStreamWaitEvent((hipStream_t)c.st, wl.done[0], 0));     // join (the side stream runs in order: layer 0 is its last)
    return XVA_OK;
}

// ------------------------------------------------------------------ temporal predictors ----
// The predictors (model.py:103-122) run on fp32-STORED tensors in both modes (bf16 MFMA with operands rounded while staged in
// bf16 mode): their gradients are sums of near-cancelling terms, which amplifies bf16 storage noise ~10x, and they are < 2 % of
// the step.  `xin` is the activation-dtype input; `pin` the workspace offset of its fp32 copy (bf16 mode).
static Ctx pred_ctx(const Ctx& c) {
    Ctx p = c;
    p.dt = XVA_F32; p.es = 4; p.Pw = (const char*)c.P;
    return p;
}
static int pred_fwd(Ctx& c0, const PredP& p, const PredA& a, const char* xin, int64_t pin, const int32_t* lens, int site) {
    Ctx c = pred_ctx(c0);
    const int64_t R = c.pl.Re; const int Tp = c.pl.Ttp;
    if (c0.dt != XVA_F32) {
        XVA_TRY(xva_cast_to_f32(xin, c0.dt, c.F(pin), R * DM, c.st));
        xin = c.A(pin);
    }
    XVA_TRY(conv3_fwd(c, xin, R, DM, p.c1_w, c.P + p.c1_b, c.A(a.c1), DP, 1, nullptr, XVA_MASK_PAD, lens, Tp));
    XVA_TRY(xva_fp_layernorm_fwd(c.A(a.c1), c.P + p.n1_g, c.P + p.n1_b, c.A(a.n1), c.dt, c.F(a.m1), c.F(a.r1), R, DP, XVA_MASK_PAD, lens, Tp,
                                 c.pd, c.seed, site + 0, c.st));
    XVA_TRY(conv3_fwd(c, c.A(a.n1), R, DP, p.c2_w, c.P + p.c2_b, c.A(a.c2), DP, 1, nullptr, XVA_MASK_PAD, lens, Tp));
    XVA_TRY(xva_fp_layernorm_fwd(c.A(a.c2), c.P + p.n2_g, c.P + p.n2_b, c.A(a.n2), c.dt, c.F(a.m2), c.F(a.r2), R, DP, XVA_MASK_PAD, lens, Tp,
                                 c.pd, c.seed, site + 1, c.st));
    XVA_TRY(linear_fwd(c, c.A(a.n2), R, DP, DP, p.fc_w, c.P + p.fc_b, c.F(a.out), 1, 1, nullptr, 0, XVA_MASK_LEN, lens, Tp, Drop{0.f, 0}, 1));
    return XVA_OK;
}
// d_out: (Re) fp32 gradient of the predictor output (zero on dead rows).  Accumulates (or writes) dL/d xin into gX (act dtype).
static int pred_bwd(Ctx& c0, const PredP& p, const PredA& a, const char* xin, int64_t pin, const float* d_out, char* gX, int accumulate,
                    const int32_t* lens, int site) {
    Ctx c = pred_ctx(c0);
    const int64_t R = c.pl.Re; const int Tp = c.pl.Ttp;
    if (c0.dt != XVA_F32) xin = c.A(pin);          // the fp32 copy made by pred_fwd
    char *pa = c.A(c.pl.pa), *pb = c.A(c.pl.pb);
    XVA_TRY(xva_fp_rowscale_colsum(c.A(a.n2), c.dt, d_out, c.G + p.fc_w, R, DP, c.st));
    XVA_TRY(xva_fp_colsum(d_out, XVA_F32, c.G + p.fc_b, R, 1, 1, c.st));
    // d (dropped n2) = d_out (x) fc_w is rank 1: the LayerNorm backward forms it on the fly                     (model.py:121)
    XVA_TRY(xva_fp_layernorm_bwd(nullptr, c.A(a.c2), c.F(a.m2), c.F(a.r2), c.P + p.n2_g, pb, nullptr, c.dt, c.G + p.n2_g, c.G + p.n2_b, R, DP,
                                 XVA_MASK_PAD, lens, Tp, 1, c.pd, c.seed, site + 1, 0.f, 0, 0, d_out, c.P + p.fc_w, c.st));   // pb = d conv2-preact
    XVA_TRY(conv3_bwd_data(c, pb, R, DP, p.c2_w, DP, pa, nullptr, nullptr, XVA_MASK_PAD, lens, Tp, 0));   // pa = d (dropped n1)
    XVA_TRY(conv3_bwd_weight(c, pb, R, DP, c.A(a.n1), DP, c.G + p.c2_w));
    XVA_TRY(xva_fp_colsum(pb, c.dt, c.G + p.c2_b, R, DP, DP, c.st));
    XVA_TRY(xva_fp_layernorm_bwd(pa, c.A(a.c1), c.F(a.m1), c.F(a.r1), c.P + p.n1_g, pb, nullptr, c.dt, c.G + p.n1_g, c.G + p.n1_b, R, DP,
                                 XVA_MASK_PAD, lens, Tp, 1, c.pd, c.seed, site + 0, 0.f, 0, 0, nullptr, nullptr, c.st));   // pb = d conv1-preact
    XVA_TRY(conv3_bwd_data(c, pb, R, DP, p.c1_w, DM, gX, nullptr, nullptr, XVA_MASK_LEN, lens, Tp, accumulate, c0.dt));
    XVA_TRY(conv3_bwd_weight(c, pb, R, DP, xin, DM, c.G + p.c1_w));
    XVA_TRY(xva_fp_colsum(pb, c.dt, c.G + p.c1_b, R, DP, DP, c.st));
    return XVA_OK;
}

static int make_ctx(Ctx& c, const xva_fp_dims* d, const float* params, float* grads, void* ws, int64_t ws_bytes, void* st) {
    XVA_TRY(make_plan(d, &c.pl));
    XVA_CHECK_ARG(params && ws, "fastpitch: null params/workspace");
    XVA_CHECK_ARG(((uintptr_t)params % 16) == 0 && ((uintptr_t)ws % 256) == 0, "fastpitch: params must be 16-byte and workspace 256-byte aligned");
    XVA_CHECK_ARG(ws_bytes >= c.pl.total, "fastpitch: workspace too small (%ld < %ld bytes)", (long)ws_bytes, (long)c.pl.total);
    c.d = d; c.P = params; c.G = grads; c.W = (char*)ws; c.st = st; c.compute = d->compute == 1 ? 1 : 0; c.h16 = d->compute == 2; c.dt = c.pl.dt; c.es = c.pl.es;
    c.pd = d->p_dropout; c.seed = d->seed;
    c.Pw = c.compute ? c.W + c.pl.wshadow : (const char*)params;
    t_h16 = c.h16;
    return XVA_OK;
}

}  // namespace

// =========================================================================== C ABI ====
extern "C" int xva_fp_set_streams(int n) { int old = g_fp_serial ? 1 : 3; g_fp_serial = n <= 1; return old; }
extern "C" int64_t xva_fp_param_floats(void) { return table().total; }
extern "C" int xva_fp_num_tensors(void) { return (int)table().t.size(); }
extern "C" int xva_fp_tensor_info(int i, char* name, int name_cap, int64_t* offset, int64_t* numel, int32_t* ndim, int64_t* shape4,
                                  int32_t* kind) {
    const ParamTable& T = table();
    XVA_CHECK_ARG(i >= 0 && i < (int)T.t.size() && name && name_cap > 0, "tensor_info: bad index");
    const TensorInfo& ti = T.t[i];
    snprintf(name, name_cap, "%s", ti.name.c_str());
    if (offset) *offset = ti.offset;
    if (numel) *numel = ti.numel;
    if (ndim) *ndim = ti.ndim;
    if (shape4) for (int k = 0; k < 4; ++k) shape4[k] = ti.shape[k];
    if (kind) *kind = ti.kind;
    return XVA_OK;
}
// Trainable [begin, end) ranges of the flat buffer for a stage (freezing of xva_train.py:589-672). Returns count.
extern "C" int xva_fp_trainable_ranges(int stage, int64_t* begins, int64_t* ends, int cap) {
    const ParamTable& T = table();
    int n = 0;
    auto push = [&](int64_t b, int64_t e) { if (n < cap) { begins[n] = b; ends[n] = e; } ++n; };
    push(T.enc_begin, T.enc_end);
    if (stage == 1 || stage == 2) push(T.dur_begin, T.dur_end);
    if (stage == 3) { push(T.pitch_begin, T.pitch_end); push(T.pemb_begin, T.pemb_end); push(T.energy_begin, T.energy_end); }
    if (stage == 2 || stage == 3 || stage == 4) push(T.eemb_begin, T.eemb_end);
    if (stage == 3 || stage == 4) { push(T.dec_begin, T.dec_end); push(T.proj_begin, T.proj_end); }
    if (stage == 1) push(T.attn_begin, T.attn_end);
    return n;
}

extern "C" int64_t xva_fp_workspace_bytes(const xva_fp_dims* d) {
    Plan p;
    if (make_plan(d, &p) != XVA_OK) return -1;
    return p.total;
}

extern "C" int xva_fp_slot_offset(const xva_fp_dims* d, int slot, int64_t* off_bytes) {
    Plan p;
    XVA_TRY(make_plan(d, &p));
    XVA_CHECK_ARG(off_bytes, "slot_offset: null");
    switch (slot) {
        case XVA_FP_SLOT_MEL_OUT: *off_bytes = p.mel_out; break;
        case XVA_FP_SLOT_PITCH_PRED: *off_bytes = p.pitch.out; break;
        case XVA_FP_SLOT_ENERGY_PRED: *off_bytes = p.energy.out; break;
        case XVA_FP_SLOT_LOG_DUR_PRED: *off_bytes = p.dur.out; break;
        case XVA_FP_SLOT_DUR_PRED: *off_bytes = p.dur_pred; break;
        case XVA_FP_SLOT_PITCH_TGT: *off_bytes = p.ptgt; break;
        case XVA_FP_SLOT_ENERGY_TGT: *off_bytes = p.etgt; break;
        case XVA_FP_SLOT_DEC_LENS: *off_bytes = p.dec_lens; break;
        case XVA_FP_SLOT_LOSS_ACC: *off_bytes = p.acc; break;
        case XVA_FP_SLOT_LOSSES: *off_bytes = p.losses; break;
        case XVA_FP_SLOT_D_MEL: *off_bytes = p.d_mel; break;
        case XVA_FP_SLOT_D_PITCH: *off_bytes = p.d_pitch; break;
        case XVA_FP_SLOT_D_ENERGY: *off_bytes = p.d_energy; break;
        case XVA_FP_SLOT_D_LOGDUR: *off_bytes = p.d_logdur; break;
        case XVA_FP_SLOT_ENC_OUT: *off_bytes = p.enc_x[NL]; break;
        case XVA_FP_SLOT_DEC_OUT: *off_bytes = p.dec_x[NL]; break;
        case XVA_FP_SLOT_ENC_COND: *off_bytes = p.enc_c2; break;
        default:
            // 100 + l / 200 + l: input of encoder / decoder layer l (l = NL: the stack's output), (B, T + 2, 384) in the activation dtype —
            // what the layer-by-layer (teacher-forced) parity check of the bf16 schedule reads
            if (slot >= 100 && slot <= 100 + NL) { *off_bytes = p.enc_x[slot - 100]; break; }
            if (slot >= 200 && slot <= 200 + NL) { *off_bytes = p.dec_x[slot - 200]; break; }
            if (slot == 300) { *off_bytes = p.pa; break; }     // diagnostics: the temporal predictors' backward scratch (Re x 256 fp32)
            if (slot == 301) { *off_bytes = p.pb; break; }
            xva_set_error("slot_offset: unknown slot %d", slot); return XVA_ERR_ARG;
    }
    return XVA_OK;
}

extern "C" int xva_fp_forward(const xva_fp_dims* d, const float* params, const xva_fp_batch* bt, void* workspace,
                              int64_t workspace_bytes, void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, params, nullptr, workspace, workspace_bytes, stream));
    XVA_CHECK_ARG(bt && bt->text && bt->in_lens && bt->pos_table, "fastpitch_forward: null batch field");
    const Plan& pl = c.pl;
    const ParamTable& T = table();
    const int B = pl.B;
    // bf16 shadow of the parameters: the encoder's slice first; the rest (predictors, embeddings, decoder, aligner: 5/6 of the bytes) on the
    // predictor lane under the encoder's forward
    WgLane& wl0 = wg_lane();
    bool split_rest = false;
    const bool split_cast = c.compute && wl0.ok && !g_fp_serial && d->stage != 2 && T.enc_begin == 0 && T.enc_end % 8 == 0;
    if (c.compute) {
        if (split_cast) {
            XVA_TRY(xva_cast_f32(params, c.W + pl.wshadow, c.dt, T.enc_end, c.st));
            XVA_HIP_TRY(hipEventRecord(wl0.pfork, (hipStream_t)c.st));      // orders the side lane after whatever wrote `params` on the caller's stream
            XVA_HIP_TRY(hipStreamWaitEvent(wl0.sp, wl0.pfork, 0));
            XVA_TRY(xva_cast_f32(params + T.enc_end, c.W + pl.wshadow + T.enc_end * c.es, c.dt, T.total - T.enc_end, wl0.sp));
            XVA_HIP_TRY(hipEventRecord(wl0.pmid, wl0.sp));
            XVA_TRY(refresh_wt_c2(c, params, wl0.sp));                     // read by backward only: behind the shadow, off the forward's critical path
        } else {
            XVA_TRY(xva_cast_f32(params, c.W + pl.wshadow, c.dt, T.total, c.st));
            XVA_TRY(refresh_wt_c2(c, params, c.st));
        }
    } else if (planes_mode(c) && wl0.ok && !g_fp_serial && d->stage != 2 && T.enc_begin == 0 && T.enc_end % 8 == 0) {
        // split products: the parameter pairs the same way — the encoder's slice first, the rest and the transposed weights on the predictor lane
        split_rest = true;
        XVA_TRY(refresh_planes_range(c, params, 0, T.enc_end, c.st));
        XVA_HIP_TRY(hipEventRecord(wl0.pfork, (hipStream_t)c.st));
        XVA_HIP_TRY(hipStreamWaitEvent(wl0.sp, wl0.pfork, 0));
        XVA_TRY(refresh_planes_range(c, params, T.enc_end, T.total, wl0.sp));
        XVA_HIP_TRY(hipEventRecord(wl0.pmid, wl0.sp));
        XVA_TRY(refresh_planes_wt(c, params, wl0.sp));
    } else XVA_TRY(refresh_planes(c, params, c.st));
    // encoder                                                              (model.py:346)
    XVA_TRY(xva_fp_embed_fwd(bt->text, c.P + T.word_emb, bt->pos_table, c.A(pl.enc_x[0]), c.dt, B, pl.Tt, DM, c.st));
    XVA_TRY(layers_fwd(c, T.enc, pl.enc, pl.enc_x, pl.Re, pl.Ttp, pl.Tse, bt->in_lens, DS_ENC));
    if (split_cast || split_rest) XVA_HIP_TRY(hipStreamWaitEvent((hipStream_t)c.st, wl0.pmid, 0));          // the rest of the shadow / pairs is in place
    char* enc_out = c.A(pl.enc_x[NL]);
    if (d->stage == 2) {                                                    // model.py:367-373
        XVA_TRY(pred_fwd(c, T.dur, pl.dur, enc_out, pl.pin_a, bt->in_lens, DS_PRED + 0));
        XVA_TRY(xva_fp_dur_from_log(c.F(pl.dur.out), c.F(pl.dur_pred), (int)pl.Re, 75.f, c.st));
        return XVA_OK;
    }
    XVA_CHECK_ARG(bt->durs && bt->pitch && bt->energy, "fastpitch_forward: stage 3/4 needs durations, pitch and energy");
    int32_t* dec_lens = (int32_t*)c.A(pl.dec_lens);
    // pitch / energy conditioning                                          (model.py:394-423)
    // The decoder input uses the TARGET pitch / energy (training): the two predictors only feed the loss, and run on the predictor lane.
    WgLane& wl = wg_lane();
    Ctx cp = c;
    const bool pl_on = wl.ok && !g_fp_serial;
    if (pl_on) {
        cp.st = wl.sp; cp.lane = 2;
        XVA_HIP_TRY(hipEventRecord(wl.pfork, (hipStream_t)c.st));
        XVA_HIP_TRY(hipStreamWaitEvent(wl.sp, wl.pfork, 0));
    }
    XVA_TRY(pred_fwd(cp, T.pitch, pl.pitch, enc_out, pl.pin_a, bt->in_lens, DS_PRED + 2));
    XVA_TRY(xva_fp_avg_pitch(bt->pitch, bt->durs, c.F(pl.ptgt), B, pl.Tt, pl.Tm, 0, c.st));
    XVA_TRY(xva_fp_cond_add_fwd(enc_out, c.F(pl.ptgt), c.P + T.pitch_emb_w, c.P + T.pitch_emb_b, c.A(pl.enc_c1), c.dt, bt->in_lens, B,
                                pl.Ttp, DM, c.st));
    if (pl_on) {   // the energy predictor reads enc_c1
        XVA_HIP_TRY(hipEventRecord(wl.pmid, (hipStream_t)c.st));
        XVA_HIP_TRY(hipStreamWaitEvent(wl.sp, wl.pmid, 0));
    }
    XVA_TRY(pred_fwd(cp, T.energy, pl.energy, c.A(pl.enc_c1), pl.pin_b, bt->in_lens, DS_PRED + 4));
    if (pl_on) XVA_HIP_TRY(hipEventRecord(wl.pjoin, wl.sp));
    XVA_TRY(xva_fp_avg_pitch(bt->energy, bt->durs, c.F(pl.etgt), B, pl.Tt, pl.Tm, 1, c.st));
    XVA_TRY(xva_fp_cond_add_fwd(c.A(pl.enc_c1), c.F(pl.etgt), c.P + T.energy_emb_w, c.P + T.energy_emb_b, c.A(pl.enc_c2), c.dt,
                                bt->in_lens, B, pl.Ttp, DM, c.st));
    // length regulation + decoder + projection                             (model.py:381-386)
    XVA_TRY(xva_fp_lenreg_map(bt->durs, (int32_t*)c.A(pl.tok), (int32_t*)c.A(pl.tstart), dec_lens, B, pl.Tt, pl.Tm, 1.0f, c.st));
    XVA_TRY(xva_fp_lenreg_fwd(c.A(pl.enc_c2), (int32_t*)c.A(pl.tok), dec_lens, bt->pos_table, c.A(pl.dec_x[0]), c.dt, B, pl.Tt, pl.Tm,
                              DM, c.st));
    XVA_TRY(layers_fwd(c, T.dec, pl.dec, pl.dec_x, pl.Rd, pl.Tmp, pl.Tsd, dec_lens, DS_DEC));
    XVA_TRY(linear_fwd(c, c.A(pl.dec_x[NL]), pl.Rd, DM, DM, T.proj_w, c.P + T.proj_b, c.A(pl.mel_out), NMEL, NMEL, nullptr, 0,
                       XVA_MASK_PAD, dec_lens, pl.Tmp));
    if (pl_on) XVA_HIP_TRY(hipStreamWaitEvent((hipStream_t)c.st, wl.pjoin, 0));     // join the predictor lane
    return XVA_OK;
}

// ---- training stage 1: the aligner ------------------------------------------------------------------------------------------
namespace {
constexpr int NATT = 80, NKH = 2 * DM, NQH = 2 * NMEL;   // attention dim, key / query hidden widths (768, 160)
struct AlignPlan {
    int B, Tt, Tm, Ttp, Tmp, ld;
    int64_t Rt, Rm;
    int64_t temb, k1, kenc, melT, q1, q2, qenc, qn, kn, S, logprob, soft, lse1, lse2, alpha, beta, G, colsum, choice, durs, loss;
    int64_t dqenc, dq2, dq1, dkenc, dk1, dtemb;
    int64_t total;
};
int make_align_plan(const xva_fp_dims* d, AlignPlan* p) {
    XVA_CHECK_ARG(d && d->B > 0 && d->Tt > 0 && d->Tm > 0 && d->Tt <= 2046 && d->Tm <= 8190, "align: bad dims");
    p->B = d->B; p->Tt = d->Tt; p->Tm = d->Tm; p->Ttp = d->Tt + 2; p->Tmp = d->Tm + 2; p->ld = (d->Tt + 3) & ~3;
    p->Rt = (int64_t)d->B * p->Ttp; p->Rm = (int64_t)d->B * p->Tmp;
    Bump b;
    const int64_t map = (int64_t)d->B * d->Tm * p->ld * 4;
    p->temb = b.seq(p->Rt, DM, 4); p->k1 = b.seq(p->Rt, NKH, 4); p->kenc = b.seq(p->Rt, NATT, 4);
    p->melT = b.seq(p->Rm, NMEL, 4); p->q1 = b.seq(p->Rm, NQH, 4); p->q2 = b.seq(p->Rm, NMEL, 4); p->qenc = b.seq(p->Rm, NATT, 4);
    p->qn = b.take(p->Rm * 4); p->kn = b.take(p->Rt * 4);
    p->S = b.take(map); p->logprob = b.take(map); p->soft = b.take(map);
    p->lse1 = b.take((int64_t)d->B * d->Tm * 4); p->lse2 = b.take((int64_t)d->B * d->Tm * 4);
    p->alpha = b.take((int64_t)d->B * d->Tm * (2 * d->Tt + 1) * 4); p->beta = b.take((int64_t)d->B * d->Tm * (2 * d->Tt + 1) * 4);
    p->G = b.take(map); p->colsum = b.take((int64_t)d->B * d->Tt * 4);
    p->choice = b.take((int64_t)d->B * d->Tm * d->Tt); p->durs = b.take((int64_t)d->B * d->Tt * 4); p->loss = b.take(16);
    p->dqenc = b.seq(p->Rm, NATT, 4); p->dq2 = b.seq(p->Rm, NMEL, 4); p->dq1 = b.seq(p->Rm, NQH, 4);
    p->dkenc = b.seq(p->Rt, NATT, 4); p->dk1 = b.seq(p->Rt, NKH, 4); p->dtemb = b.seq(p->Rt, DM, 4);
    p->total = b.cur;
    return XVA_OK;
}
struct ACtx {
    AlignPlan pl; const float* P; float* G; char* W; void* st; int compute;
    float* F(int64_t off) const { return (float*)(W + off); }
};
xva_gemm_params agp(const ACtx& c) {
    xva_gemm_params g;
    memset(&g, 0, sizeof(g));
    g.batch = 1; g.batch2 = 1; g.alpha = 1.f; g.beta = 1.f; g.splitk = 1; g.compute = c.compute; g.mask_pad = 1; g.mask_mul = 1;
    g.a_dtype = g.b_dtype = g.c_dtype = XVA_F32; g.r_dtype = g.g_dtype = XVA_F32;
    return g;
}
// Y[rows, N] = act(Xcat W^T + b): conv k (1 or 3) over a padded token-major fp32 sequence; W tap-major [N][k*Cin]
int a_conv_fwd(ACtx& c, const float* X, int64_t rows, int Cin, int k, int64_t w_off, int64_t b_off, float* Y, int N, int relu) {
    xva_gemm_params g = agp(c);
    g.layout = XVA_GEMM_NT; g.A = X - (k == 3 ? Cin : 0); g.B = c.P + w_off; g.C = Y; g.M = (int)rows; g.N = N; g.K = k * Cin;
    g.lda = Cin; g.ldb = k * Cin; g.ldc = N; g.bias = c.P + b_off; g.act = relu ? XVA_ACT_RELU : XVA_ACT_NONE;
    return xva_gemm(&g, c.st);
}
// dX = (dY (*) W^T-conv) gated by Gate > 0 (ReLU backward) — k = 1: dX = dY W ; k = 3: tap-reversed segments
int a_conv_bwd_data(ACtx& c, const float* dY, int64_t rows, int N, int k, int64_t w_off, int Cin, float* dX, const float* Gate) {
    xva_gemm_params g = agp(c);
    g.layout = XVA_GEMM_NN; g.A = dY - (k == 3 ? N : 0); g.B = c.P + w_off; g.C = dX; g.M = (int)rows; g.N = Cin; g.K = k * N;
    g.lda = N; g.ldb = k * Cin; g.ldc = Cin;
    if (k == 3) { g.seglen = N; g.seg0 = 2 * Cin; g.segstride = -Cin; }
    g.G = Gate; g.ldg = Cin;
    return xva_gemm(&g, c.st);
}
// dW[N][k*Cin] += scale-free dY^T Xcat ; db[N] += colsum(dY)
int a_conv_bwd_weight(ACtx& c, const float* dY, int64_t rows, int N, int k, const float* X, int Cin, int64_t w_off, int64_t b_off) {
    xva_gemm_params g = agp(c);
    g.layout = XVA_GEMM_TN; g.A = dY; g.B = X - (k == 3 ? Cin : 0); g.C = c.G + w_off; g.M = N; g.N = k * Cin; g.K = (int)rows;
    g.lda = N; g.ldb = Cin; g.ldc = k * Cin; g.accumulate = 1; g.splitk = 0;
    XVA_TRY(xva_gemm(&g, c.st));
    return xva_fp_colsum(dY, XVA_F32, c.G + b_off, rows, N, N, c.st);
}
int make_actx(ACtx& c, const xva_fp_dims* d, const float* params, float* grads, void* ws, int64_t ws_bytes, void* st) {
    XVA_TRY(make_align_plan(d, &c.pl));
    XVA_CHECK_ARG(params && ws && ((uintptr_t)ws % 256) == 0 && ((uintptr_t)params % 16) == 0, "align: null / misaligned params or workspace");
    XVA_CHECK_ARG(ws_bytes >= c.pl.total, "align: workspace too small (%ld < %ld bytes)", (long)ws_bytes, (long)c.pl.total);
    c.P = params; c.G = grads; c.W = (char*)ws; c.st = st; c.compute = d->compute == 2 ? 2 : (d->compute ? 1 : 0);      // fp32-stored; 2: split-bf16 products
    return XVA_OK;
}
}  // namespace

extern "C" int64_t xva_fp_align_workspace_bytes(const xva_fp_dims* d) {
    AlignPlan p;
    if (make_align_plan(d, &p) != XVA_OK) return -1;
    return p.total;
}

extern "C" int xva_fp_align_forward(const xva_fp_dims* d, const float* params, const xva_fp_align_batch* bt, void* workspace, int64_t workspace_bytes,
                                    float* attn_soft_out, float* attn_logprob_out, int32_t* durs_out, float* loss_out, void* stream) {
    ACtx c;
    XVA_TRY(make_actx(c, d, params, nullptr, workspace, workspace_bytes, stream));
    XVA_CHECK_ARG(bt && bt->text && bt->in_lens && bt->mel && bt->mel_lens && bt->attn_prior && durs_out && loss_out, "align_forward: null");
    const AlignPlan& pl = c.pl;
    const ParamTable& T = table();
    const int B = pl.B, Tt = pl.Tt, Tm = pl.Tm, ld = pl.ld;
    hipStream_t hs = (hipStream_t)stream;
    // keys: word embeddings -> key_proj                                                          (model.py:300; attention.py:189)
    XVA_TRY(xva_al_embed(bt->text, c.P + T.word_emb, c.F(pl.temb), B, Tt, DM, stream));
    XVA_TRY(a_conv_fwd(c, c.F(pl.temb), pl.Rt, DM, 3, T.at.k0_w, T.at.k0_b, c.F(pl.k1), NKH, 1));
    XVA_TRY(a_conv_fwd(c, c.F(pl.k1), pl.Rt, NKH, 1, T.at.k2_w, T.at.k2_b, c.F(pl.kenc), NATT, 0));
    // queries: target mel -> query_proj                                                          (attention.py:192-196)
    XVA_TRY(xva_al_mel_to_tm(bt->mel, c.F(pl.melT), B, NMEL, Tm, stream));
    XVA_TRY(a_conv_fwd(c, c.F(pl.melT), pl.Rm, NMEL, 3, T.at.q0_w, T.at.q0_b, c.F(pl.q1), NQH, 1));
    XVA_TRY(a_conv_fwd(c, c.F(pl.q1), pl.Rm, NQH, 1, T.at.q2_w, T.at.q2_b, c.F(pl.q2), NMEL, 1));
    XVA_TRY(a_conv_fwd(c, c.F(pl.q2), pl.Rm, NMEL, 1, T.at.q4_w, T.at.q4_b, c.F(pl.qenc), NATT, 0));
    XVA_TRY(xva_al_sqnorm(c.F(pl.qenc), c.F(pl.qn), pl.Rm, NATT, stream));
    XVA_TRY(xva_al_sqnorm(c.F(pl.kenc), c.F(pl.kn), pl.Rt, NATT, stream));
    {   // S = 0.001 q.k per item: -0.0005 ||q - k||^2 = S - 0.0005 (|q|^2 + |k|^2)                (attention.py:206-208)
        xva_gemm_params g = agp(c);
        g.layout = XVA_GEMM_NT; g.A = c.F(pl.qenc) + NATT; g.B = c.F(pl.kenc) + NATT; g.C = c.F(pl.S); g.M = Tm; g.N = Tt; g.K = NATT;
        g.lda = NATT; g.ldb = NATT; g.ldc = ld; g.batch = B; g.sA = (int64_t)pl.Tmp * NATT; g.sB = (int64_t)pl.Ttp * NATT; g.sC = (int64_t)Tm * ld;
        g.alpha = 0.001f;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    XVA_TRY(xva_al_attn_rows(c.F(pl.S), c.F(pl.qn), c.F(pl.kn), bt->attn_prior, bt->in_lens, c.F(pl.logprob), c.F(pl.soft), c.F(pl.lse1),
                             c.F(pl.lse2), B, Tm, Tt, ld, stream));                                                  // :209-219
    XVA_TRY(xva_al_mas(c.F(pl.soft), bt->in_lens, bt->mel_lens, (uint8_t*)(c.W + pl.choice), (int32_t*)(c.W + pl.durs), B, Tm, Tt, ld, stream));   // model.py:315-318
    // forward-sum loss; its gradient w.r.t. attn_logprob is produced in the same pass (scaled in backward)   (attn_loss_function.py:27-44)
    if (hipMemsetAsync(c.W + pl.loss, 0, 16, hs) != hipSuccess) { xva_set_error("align_forward: memset failed"); return XVA_ERR_HIP; }
    XVA_TRY(xva_al_ctc(c.F(pl.logprob), c.F(pl.lse2), bt->in_lens, bt->mel_lens, c.F(pl.alpha), c.F(pl.beta), c.F(pl.G), c.F(pl.loss), B, Tm, Tt, ld,
                       1.f, stream));
    bool ok = hipMemcpyAsync(loss_out, c.W + pl.loss, 4, hipMemcpyDeviceToDevice, hs) == hipSuccess &&
              hipMemcpyAsync(durs_out, c.W + pl.durs, (size_t)B * Tt * 4, hipMemcpyDeviceToDevice, hs) == hipSuccess;
    if (attn_soft_out) ok = ok && hipMemcpy2DAsync(attn_soft_out, (size_t)Tt * 4, c.W + pl.soft, (size_t)ld * 4, (size_t)Tt * 4, (size_t)B * Tm, hipMemcpyDeviceToDevice, hs) == hipSuccess;
    if (attn_logprob_out) ok = ok && hipMemcpy2DAsync(attn_logprob_out, (size_t)Tt * 4, c.W + pl.logprob, (size_t)ld * 4, (size_t)Tt * 4, (size_t)B * Tm, hipMemcpyDeviceToDevice, hs) == hipSuccess;
    if (!ok) { xva_set_error("align_forward: output copy failed"); return XVA_ERR_HIP; }
    return XVA_OK;
}

extern "C" int xva_fp_align_backward(const xva_fp_dims* d, const float* params, float* grads, const xva_fp_align_batch* bt, void* workspace,
                                     int64_t workspace_bytes, float grad_scale, void* stream) {
    ACtx c;
    XVA_TRY(make_actx(c, d, params, grads, workspace, workspace_bytes, stream));
    XVA_CHECK_ARG(grads && bt && bt->text && bt->in_lens, "align_backward: null");
    const AlignPlan& pl = c.pl;
    const ParamTable& T = table();
    const int B = pl.B, Tt = pl.Tt, Tm = pl.Tm, ld = pl.ld;
    hipStream_t hs = (hipStream_t)stream;
    // G holds d(loss)/d(attn_logprob) from the forward (unit scale): through the first log-softmax -> d/d(-0.0005 ||q-k||^2)
    if (hipMemsetAsync(c.W + pl.colsum, 0, (size_t)B * Tt * 4, hs) != hipSuccess) { xva_set_error("align_backward: memset failed"); return XVA_ERR_HIP; }
    XVA_TRY(xva_al_logsoftmax_bwd(c.F(pl.S), c.F(pl.qn), c.F(pl.kn), c.F(pl.lse1), c.F(pl.G), c.F(pl.colsum), B, Tm, Tt, ld, stream));
    {   // d q = 0.001 G K   (the -|q|^2 term cancels: every row of G sums to zero)
        xva_gemm_params g = agp(c);
        g.layout = XVA_GEMM_NN; g.A = c.F(pl.G); g.B = c.F(pl.kenc) + NATT; g.C = c.F(pl.dqenc) + NATT; g.M = Tm; g.N = NATT; g.K = Tt;
        g.lda = ld; g.ldb = NATT; g.ldc = NATT; g.batch = B; g.sA = (int64_t)Tm * ld; g.sB = (int64_t)pl.Ttp * NATT; g.sC = (int64_t)pl.Tmp * NATT;
        g.alpha = 0.001f * grad_scale;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    {   // d k = 0.001 (G^T Q - colsum(G) o k)
        xva_gemm_params g = agp(c);
        g.layout = XVA_GEMM_TN; g.A = c.F(pl.G); g.B = c.F(pl.qenc) + NATT; g.C = c.F(pl.dkenc) + NATT; g.M = Tt; g.N = NATT; g.K = Tm;
        g.lda = ld; g.ldb = NATT; g.ldc = NATT; g.batch = B; g.sA = (int64_t)Tm * ld; g.sB = (int64_t)pl.Tmp * NATT; g.sC = (int64_t)pl.Ttp * NATT;
        g.alpha = 0.001f * grad_scale;
        XVA_TRY(xva_gemm(&g, c.st));
    }
    XVA_TRY(xva_al_dk_fix(c.F(pl.dkenc), c.F(pl.kenc), c.F(pl.colsum), B, Tt, NATT, grad_scale, stream));
    // query projections (weights only: the mel is data)                                          (attention.py:118-133)
    XVA_TRY(a_conv_bwd_weight(c, c.F(pl.dqenc), pl.Rm, NATT, 1, c.F(pl.q2), NMEL, T.at.q4_w, T.at.q4_b));
    XVA_TRY(a_conv_bwd_data(c, c.F(pl.dqenc), pl.Rm, NATT, 1, T.at.q4_w, NMEL, c.F(pl.dq2), c.F(pl.q2)));
    XVA_TRY(a_conv_bwd_weight(c, c.F(pl.dq2), pl.Rm, NMEL, 1, c.F(pl.q1), NQH, T.at.q2_w, T.at.q2_b));
    XVA_TRY(a_conv_bwd_data(c, c.F(pl.dq2), pl.Rm, NMEL, 1, T.at.q2_w, NQH, c.F(pl.dq1), c.F(pl.q1)));
    XVA_TRY(a_conv_bwd_weight(c, c.F(pl.dq1), pl.Rm, NQH, 3, c.F(pl.melT), NMEL, T.at.q0_w, T.at.q0_b));
    // key projections and the word embedding                                                     (attention.py:103-113; model.py:300)
    XVA_TRY(a_conv_bwd_weight(c, c.F(pl.dkenc), pl.Rt, NATT, 1, c.F(pl.k1), NKH, T.at.k2_w, T.at.k2_b));
    XVA_TRY(a_conv_bwd_data(c, c.F(pl.dkenc), pl.Rt, NATT, 1, T.at.k2_w, NKH, c.F(pl.dk1), c.F(pl.k1)));
    XVA_TRY(a_conv_bwd_weight(c, c.F(pl.dk1), pl.Rt, NKH, 3, c.F(pl.temb), DM, T.at.k0_w, T.at.k0_b));
    XVA_TRY(a_conv_bwd_data(c, c.F(pl.dk1), pl.Rt, NKH, 3, T.at.k0_w, DM, c.F(pl.dtemb), nullptr));
    XVA_TRY(xva_fp_embed_bwd(bt->text, c.F(pl.dtemb), XVA_F32, c.G + T.word_emb, B, Tt, DM, stream));
    return XVA_OK;
}

// ---- inference (model.py:426-481) ------------------------------------------------------------------------------------------
extern "C" int xva_fp_infer_encode(const xva_fp_dims* d0, const float* params, const xva_fp_batch* bt, float pace, float max_duration,
                                   void* workspace, int64_t workspace_bytes, void* enc_cond_out, int32_t* durs_out, int32_t* dec_lens_out,
                                   float* dur_pred_out, float* pitch_pred_out, float* energy_pred_out, void* stream) {
    XVA_CHECK_ARG(d0 && bt && bt->text && bt->in_lens && bt->pos_table && enc_cond_out && durs_out && dec_lens_out, "infer_encode: null");
    xva_fp_dims d = *d0;
    d.stage = 3; d.Tm = 1; d.p_dropout = 0.f;
    Ctx c;
    XVA_TRY(make_ctx(c, &d, params, nullptr, workspace, workspace_bytes, stream));
    const Plan& pl = c.pl;
    const ParamTable& T = table();
    const int B = pl.B;
    if (c.compute) XVA_TRY(xva_cast_f32(params, c.W + pl.wshadow, c.dt, T.total, c.st));
    else XVA_TRY(refresh_planes(c, params, c.st));
    XVA_TRY(xva_fp_embed_fwd(bt->text, c.P + T.word_emb, bt->pos_table, c.A(pl.enc_x[0]), c.dt, B, pl.Tt, DM, c.st));
    XVA_TRY(layers_fwd(c, T.enc, pl.enc, pl.enc_x, pl.Re, pl.Ttp, pl.Tse, bt->in_lens, DS_ENC));
    char* enc_out = c.A(pl.enc_x[NL]);
    XVA_TRY(pred_fwd(c, T.dur, pl.dur, enc_out, pl.pin_a, bt->in_lens, DS_PRED + 0));                          // :439-440
    XVA_TRY(xva_fp_dur_from_log(c.F(pl.dur.out), c.F(pl.dur_pred), (int)pl.Re, max_duration, c.st));
    XVA_TRY(pred_fwd(c, T.pitch, pl.pitch, enc_out, pl.pin_a, bt->in_lens, DS_PRED + 2));                      // :443
    XVA_TRY(xva_fp_cond_add_fwd(enc_out, c.F(pl.pitch.out), c.P + T.pitch_emb_w, c.P + T.pitch_emb_b, c.A(pl.enc_c1), c.dt, bt->in_lens, B,
                                pl.Ttp, DM, c.st));                                                             // :454-459
    XVA_TRY(pred_fwd(c, T.energy, pl.energy, c.A(pl.enc_c1), pl.pin_b, bt->in_lens, DS_PRED + 4));             // :465
    XVA_TRY(xva_fp_cond_add_fwd(c.A(pl.enc_c1), c.F(pl.energy.out), c.P + T.energy_emb_w, c.P + T.energy_emb_b, c.A(pl.enc_c2), c.dt,
                                bt->in_lens, B, pl.Ttp, DM, c.st));                                             // :466-470
    XVA_TRY(xva_fp_infer_finish(c.F(pl.dur_pred), c.F(pl.pitch.out), c.F(pl.energy.out), bt->in_lens, pace, B, pl.Tt, durs_out, dec_lens_out,
                                dur_pred_out, pitch_pred_out, energy_pred_out, c.st));
    if (hipMemcpyAsync(enc_cond_out, c.A(pl.enc_c2), pl.Re * DM * c.es, hipMemcpyDeviceToDevice, (hipStream_t)c.st) != hipSuccess) {
        xva_set_error("infer_encode: memcpy failed");
        return XVA_ERR_HIP;
    }
    return XVA_OK;
}

extern "C" int xva_fp_infer_decode(const xva_fp_dims* d0, const float* params, const void* enc_cond, const int32_t* durs, const float* pos_table,
                                   void* workspace, int64_t workspace_bytes, float* mel_out, void* stream) {
    XVA_CHECK_ARG(d0 && enc_cond && durs && pos_table && mel_out, "infer_decode: null");
    xva_fp_dims d = *d0;
    d.stage = 3; d.p_dropout = 0.f;
    Ctx c;
    XVA_TRY(make_ctx(c, &d, params, nullptr, workspace, workspace_bytes, stream));
    const Plan& pl = c.pl;
    const ParamTable& T = table();
    const int B = pl.B;
    if (c.compute) XVA_TRY(xva_cast_f32(params, c.W + pl.wshadow, c.dt, T.total, c.st));
    else XVA_TRY(refresh_planes(c, params, c.st));
    int32_t* dec_lens = (int32_t*)c.A(pl.dec_lens);
    XVA_TRY(xva_fp_lenreg_map(durs, (int32_t*)c.A(pl.tok), (int32_t*)c.A(pl.tstart), dec_lens, B, pl.Tt, pl.Tm, 1.0f, c.st));   // :472-474
    XVA_TRY(xva_fp_lenreg_fwd(enc_cond, (int32_t*)c.A(pl.tok), dec_lens, pos_table, c.A(pl.dec_x[0]), c.dt, B, pl.Tt, pl.Tm, DM, c.st));
    XVA_TRY(layers_fwd(c, T.dec, pl.dec, pl.dec_x, pl.Rd, pl.Tmp, pl.Tsd, dec_lens, DS_DEC));                   // :476
    XVA_TRY(linear_fwd(c, c.A(pl.dec_x[NL]), pl.Rd, DM, DM, T.proj_w, c.P + T.proj_b, c.A(pl.mel_out), NMEL, NMEL, nullptr, 0, XVA_MASK_PAD,
                       dec_lens, pl.Tmp));                                                                       // :477
    XVA_TRY(xva_fp_mel_to_bct(c.A(pl.mel_out), c.dt, mel_out, B, pl.Tm, NMEL, c.st));                           // :479
    return XVA_OK;
}

// Gradient buckets in the order backward completes them (each a contiguous [begin, end) range of the flat buffer):
//   0..5   decoder layers 5..0 (bucket 0 also holds proj)      6   predictors + pitch/energy embeddings
//   7..12  encoder layers 5..0 (bucket 12 also holds word_emb)
extern "C" int xva_fp_num_buckets(void) { return 2 * NL + 1; }
extern "C" int xva_fp_bucket_range(int i, int64_t* begin, int64_t* end) {
    const ParamTable& T = table();
    XVA_CHECK_ARG(i >= 0 && i < 2 * NL + 1 && begin && end, "bucket_range: bad index");
    if (i < NL) {
        int l = NL - 1 - i;
        *begin = T.dec[l].qkv_w;
        *end = (l == NL - 1) ? T.proj_end : T.dec[l + 1].qkv_w;
    } else if (i == NL) {
        *begin = T.dur_begin; *end = T.dec_begin;
    } else {
        int l = NL - 1 - (i - NL - 1);
        *begin = (l == 0) ? T.enc_begin : T.enc[l].qkv_w;
        *end = (l == NL - 1) ? T.enc_end : T.enc[l + 1].qkv_w;
    }
    return XVA_OK;
}

extern "C" int xva_fp_backward_ex(const xva_fp_dims* d, const float* params, float* grads, const xva_fp_batch* bt, void* workspace,
                                  int64_t workspace_bytes, void* const* bucket_events, void* stream);
extern "C" int xva_fp_backward(const xva_fp_dims* d, const float* params, float* grads, const xva_fp_batch* bt, void* workspace,
                               int64_t workspace_bytes, void* stream) {
    return xva_fp_backward_ex(d, params, grads, bt, workspace, workspace_bytes, nullptr, stream);
}
// Gradients of the loss w.r.t. the model outputs must already sit in the workspace slots D_MEL / D_PITCH / D_ENERGY /
// D_LOGDUR (xva_fp_loss_grads writes them there).  Accumulates into `grads` (the caller zeroes it when a new optimizer step
// starts: gradient accumulation across micro-batches is the reference's "GAM").  Records bucket_events[i] (hipEvent_t,
// xva_fp_num_buckets() of them, entries may be null) on `stream` as soon as bucket i's gradients are complete.
extern "C" int xva_fp_backward_ex(const xva_fp_dims* d, const float* params, float* grads, const xva_fp_batch* bt, void* workspace,
                                  int64_t workspace_bytes, void* const* bucket_events, void* stream) {
    Ctx c;
    XVA_TRY(make_ctx(c, d, params, grads, workspace, workspace_bytes, stream));
    c.events = bucket_events;
    XVA_CHECK_ARG(grads && ((uintptr_t)grads % 16) == 0, "fastpitch_backward: grads null or misaligned");
    XVA_CHECK_ARG(bt && bt->text && bt->in_lens, "fastpitch_backward: null batch field");
    const Plan& pl = c.pl;
    const ParamTable& T = table();
    const int B = pl.B;
    char* gA = c.A(pl.gA);
    char* gE = c.A(pl.gE);
    char* enc_out = c.A(pl.enc_x[NL]);
    if (d->stage == 2) {
        for (int i = 0; i < NL; ++i) XVA_TRY(c.record(i));   // decoder buckets carry no gradient in stage 2
        XVA_TRY(pred_bwd(c, T.dur, pl.dur, enc_out, pl.pin_a, c.F(pl.d_logdur), gA, 0, bt->in_lens, DS_PRED + 0));
    } else {
        int32_t* dec_lens = (int32_t*)c.A(pl.dec_lens);
        char* d_mel = c.A(pl.d_mel);
        // stage 3: the predictors' backward (their inputs are the loss gradients and stored activations) on the predictor lane, under the
        // decoder's; each writes its d(encoder output) contribution to its own buffer, added into gE below
        WgLane& wl = wg_lane();
        const bool plane = wl.ok && !g_fp_serial && d->stage == 3;
        if (plane) {
            Ctx cp = c; cp.st = wl.sp; cp.lane = 2;
            XVA_HIP_TRY(hipEventRecord(wl.pfork, (hipStream_t)c.st));
            XVA_HIP_TRY(hipStreamWaitEvent(wl.sp, wl.pfork, 0));
            XVA_TRY(pred_bwd(cp, T.energy, pl.energy, c.A(pl.enc_c1), pl.pin_b, c.F(pl.d_energy), c.A(pl.gX1), 0, bt->in_lens, DS_PRED + 4));
            XVA_TRY(pred_bwd(cp, T.pitch, pl.pitch, enc_out, pl.pin_a, c.F(pl.d_pitch), c.A(pl.gX2), 0, bt->in_lens, DS_PRED + 2));
            XVA_HIP_TRY(hipEventRecord(wl.pjoin, wl.sp));
        }
        // proj backward                                                    (model.py:386)
        XVA_TRY(linear_bwd_data(c, d_mel, pl.Rd, NMEL, NMEL, T.proj_w, DM, gA, DM, nullptr, 0, XVA_MASK_NONE, nullptr, 0));
        XVA_TRY(linear_bwd_weight(c, d_mel, pl.Rd, NMEL, NMEL, c.A(pl.dec_x[NL]), DM, DM, c.G + T.proj_w));
        XVA_TRY(xva_fp_colsum(d_mel, c.dt, c.G + T.proj_b, pl.Rd, NMEL, NMEL, c.st));
        c.ev_base = 0;
        XVA_TRY(layers_bwd(c, T.dec, pl.dec, pl.dec_x, pl.Rd, pl.Tmp, pl.Tsd, dec_lens, false, DS_DEC));
        // regulate_len backward: segmented sum of frame grads per token -> gE = d enc_c2 = d enc_c1
        XVA_TRY(xva_fp_lenreg_bwd(gA, (int32_t*)c.A(pl.tstart), dec_lens, gE, c.dt, B, pl.Tt, pl.Tm, DM, 0, c.st));
        XVA_TRY(xva_fp_cond_add_bwd(gE, c.dt, c.F(pl.etgt), c.G + T.energy_emb_w, c.G + T.energy_emb_b, bt->in_lens, B, pl.Ttp, DM, c.st));
        if (d->stage == 3) {
            if (plane) {
                XVA_HIP_TRY(hipStreamWaitEvent((hipStream_t)c.st, wl.pjoin, 0));
                XVA_TRY(xva_fp_add_act(gE, c.A(pl.gX1), c.dt, pl.Re * DM, c.st));
            } else XVA_TRY(pred_bwd(c, T.energy, pl.energy, c.A(pl.enc_c1), pl.pin_b, c.F(pl.d_energy), gE, 1, bt->in_lens, DS_PRED + 4));
            XVA_TRY(xva_fp_cond_add_bwd(gE, c.dt, c.F(pl.ptgt), c.G + T.pitch_emb_w, c.G + T.pitch_emb_b, bt->in_lens, B, pl.Ttp, DM, c.st));
            if (plane) XVA_TRY(xva_fp_add_act(gE, c.A(pl.gX2), c.dt, pl.Re * DM, c.st));
            else XVA_TRY(pred_bwd(c, T.pitch, pl.pitch, enc_out, pl.pin_a, c.F(pl.d_pitch), gE, 1, bt->in_lens, DS_PRED + 2));
        }
        // hand over to the encoder stack: gA <- gE
        if (hipMemcpyAsync(gA, gE, pl.Re * DM * c.es, hipMemcpyDeviceToDevice, (hipStream_t)c.st) != hipSuccess) {
            xva_set_error("fastpitch_backward: memcpy failed");
            return XVA_ERR_HIP;
        }
    }
    XVA_TRY(c.record(NL));          // predictors + conditioning embeddings bucket
    c.ev_base = NL + 1;
    XVA_TRY(layers_bwd(c, T.enc, pl.enc, pl.enc_x, pl.Re, pl.Ttp, pl.Tse, bt->in_lens, true, DS_ENC));
    XVA_TRY(xva_fp_embed_bwd(bt->text, gA, c.dt, c.G + T.word_emb, B, pl.Tt, DM, c.st));
    XVA_TRY(c.record(2 * NL));      // encoder layer 0 + word embedding bucket
    return XVA_OK;
}
