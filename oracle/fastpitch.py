"""CPU oracle for the FastPitch1.1 training step (floating point: torch fp32/fp64 on CPU, autograd for grads).

A functional restatement driven by a state_dict with the REFERENCE's keys and layouts, so the same
checkpoint feeds the reference, this oracle and the HIP path.  Restates:
  FFTransformer / TransformerLayer / MultiHeadAttn / PositionwiseConvFF / PositionalEmbedding
                                      python/fastpitch1_1/fastpitch/transformer.py:21-243
  TemporalPredictor, ConvReLUNorm     python/fastpitch1_1/fastpitch/model.py:103-122, common/layers.py:85-97
  regulate_len, average_pitch         python/fastpitch1_1/fastpitch/model.py:59-100
  FastPitch.forward (stages 2, 3, 4)  python/fastpitch1_1/fastpitch/model.py:325-423
  FastPitchLoss.forward               python/fastpitch1_1/fastpitch/loss_function.py:63-154
  Lamb.step                           python/fastpitch1_1/lamb.py:40-106
  adjust_learning_rate                python/fastpitch1_1/xva_train.py:1252-1261
Dropout: the goldens are taken with model.eval() (torch's dropout RNG stream cannot be reproduced by another implementation;
SURVEY.md §7 "hard parts").  For the training-mode check the oracle applies nn.Dropout's arithmetic (x * m / (1 - p)) at the
reference's sites (transformer.py:51,127,139; common/layers.py:97) with the masks supplied by a callable — HashDropout below
restates the HIP path's stateless mask function, so both sides drop the same elements.
Stage 1 (ConvAttention + MAS) is a "next" row and is not restated yet.
"""
import math

import torch
import torch.nn.functional as F

N_LAYERS = 6
D_MODEL = 384
D_HEAD = 64
N_MEL = 80
N_SYMBOLS = 148


# ---- storage="bf16": the throughput mode's rounding points --------------------------------------------------------------------
# The HIP engine's bf16 mode (csrc/fastpitch_engine.hip: make_plan / layers_fwd / layers_bwd) STORES sequence activations and their
# gradients in bf16, feeds the MFMAs a bf16 shadow of the fp32 master weights, accumulates in fp32 and runs every epilogue
# (bias, residual, dropout, ReLU, LayerNorm statistics, softmax, losses, LAMB) in fp32.  The three temporal predictors keep fp32
# storage but their GEMM operands are rounded to bf16 in flight.  Restating exactly those roundings on top of the fp32 graph gives
# an oracle the bf16 engine can be held to tightly (tests/test_fastpitch_gpu.py: 2e-3 outputs / loss, 1e-2 per-tensor gradients)
# instead of the loose bf16-vs-fp32 bounds:
#   s(x)   a STORED tensor: value rounded in forward, its gradient rounded in backward (both live in bf16 buffers)
#   q(x)   a GEMM OPERAND rounded on its way into the MFMA (weight shadow, fp32-stored predictor tensors): forward only
#   gq(x)  the upstream gradient rounded on ITS way into the backward GEMMs of an fp32-stored tensor: backward only
class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd, dt=torch.bfloat16):
        ctx.bwd, ctx.dt = bwd, dt
        return x.to(dt).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(ctx.dt).to(g.dtype) if ctx.bwd else g), None, None, None


class Fp32Storage:
    name = "fp32"
    flash = False
    resid32 = True
    s = q = gq = r = o = staticmethod(lambda x: x)


def make_storage(name, dt, resid32=False):
    """A 16-bit storage model.  s / q / gq as above, in dtype `dt` (bf16: 8 mantissa bits, the throughput mode; fp16: 11 bits, the width the
    reference's own GPU path computes in under autocast — python/fastpitch1_1/xva_train.py:350,787).
    resid32 = the RESIDUAL STREAM (embedding sums, o_net / FFN output + residual, LayerNorm inputs and outputs) stays fp32 and is rounded only
    on its way into a product:  r(x) stores a residual-stream tensor (identity when resid32), o(x) reads one as a GEMM operand."""
    s = staticmethod(lambda x: _Round.apply(x, True, True, dt))
    q = staticmethod(lambda x: _Round.apply(x, True, False, dt))
    gq = staticmethod(lambda x: _Round.apply(x, False, True, dt))
    ident = staticmethod(lambda x: x)
    return type(name, (), dict(name=name, flash=True, resid32=resid32, dtype=dt, s=s, q=q, gq=gq,
                               r=ident if resid32 else s, o=q if resid32 else ident))


Bf16Storage = make_storage("bf16", torch.bfloat16)
F16Storage = make_storage("f16", torch.float16)
Bf16Resid32Storage = make_storage("bf16_r32", torch.bfloat16, True)
F16Resid32Storage = make_storage("f16_r32", torch.float16, True)
_STORAGES = {"bf16": Bf16Storage, "f16": F16Storage, "bf16_r32": Bf16Resid32Storage, "f16_r32": F16Resid32Storage}


def _storage(storage):
    if storage is None or storage == "fp32":
        return Fp32Storage
    if isinstance(storage, str):
        return _STORAGES[storage]
    return storage


def mask_from_lens(lens, max_len):
    ids = torch.arange(0, max_len, device=lens.device, dtype=lens.dtype)
    return torch.lt(ids, lens.unsqueeze(1))


def positional_embedding(T, demb, dtype):
    inv_freq = 1 / (10000 ** (torch.arange(0.0, demb, 2.0) / demb))
    pos_seq = torch.arange(T).to(dtype)
    sinusoid = torch.matmul(pos_seq.unsqueeze(-1), inv_freq.to(dtype).unsqueeze(0))
    return torch.cat([sinusoid.sin(), sinusoid.cos()], dim=1)[None]


class HashDropout:
    """Mask source equal to xva_dropout_scale (xva-trainer_amd/csrc/xva_common.h): element idx of dropout site `stream` is dropped
    iff (keyed 32-bit xorshift-multiply hash(seed, stream, idx) >> 8) * 2^-24 < p; kept elements are scaled by 1/(1-p).  Indices follow the HIP
    path's padded layouts: activations (B, T+2, C) row-major, attention probabilities rows of T+2 columns."""

    def __init__(self, p, seed):
        import numpy as np
        self.np = np
        self.p = np.float32(p)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    @staticmethod
    def _mix32(x):
        import numpy as np
        x = np.asarray(x, dtype=np.uint32).copy()
        with np.errstate(over="ignore"):
            x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
            x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
            x ^= x >> np.uint32(16)
        return x

    def _mult(self, stream, idx):
        np = self.np
        with np.errstate(over="ignore"):
            inner = self._mix32(np.uint32(((self.seed >> 32) + 0x9E3779B9 * (stream + 1)) & 0xFFFFFFFF))
            k1 = self._mix32(np.uint32(self.seed & 0xFFFFFFFF) ^ inner)
            k2 = self._mix32(k1 + np.uint32(0x85ebca6b))
            idx = idx.astype(np.uint64)
            x = ((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32) + (idx >> np.uint64(32)).astype(np.uint32)) ^ k1
            x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
            x ^= k2
            x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
            x ^= x >> np.uint32(16)
        h = x >> np.uint32(8)
        u = h.astype(np.float32) * np.float32(1.0 / 16777216.0)
        keep = np.float32(1.0) / (np.float32(1.0) - self.p)
        return np.where(u < self.p, np.float32(0.0), keep).astype(np.float32)

    def act(self, stream, x):
        """x: (B, T, C) activation; padded row index = b * (T + 2) + t + 1."""
        if self.p <= 0:
            return x
        np = self.np
        B, T, C = x.shape
        rows = (np.arange(B, dtype=np.uint64)[:, None] * np.uint64(T + 2) + np.arange(1, T + 1, dtype=np.uint64)[None, :])
        idx = rows[:, :, None] * np.uint64(C) + np.arange(C, dtype=np.uint64)[None, None, :]
        return x * torch.from_numpy(self._mult(stream, idx)).to(x.dtype)

    def prob(self, stream, pr):
        """pr: (B, T, T) attention probabilities; index = (b * Tp + i + 1) * Tp + j + 1 with Tp = T + 2."""
        if self.p <= 0:
            return pr
        np = self.np
        B, T, _ = pr.shape
        Tp = np.uint64(T + 2)
        rows = (np.arange(B, dtype=np.uint64)[:, None] * Tp + np.arange(1, T + 1, dtype=np.uint64)[None, :])
        idx = rows[:, :, None] * Tp + np.arange(1, T + 1, dtype=np.uint64)[None, None, :]
        return pr * torch.from_numpy(self._mult(stream, idx)).to(pr.dtype)


def _flash_pv(score, v, mult, st):
    """softmax(score) @ v the way csrc/attention.hip:attn_fwd_kernel evaluates it: keys in blocks of 64 PADDED positions (key k sits at
    padded index k + 1), a running row maximum, UNNORMALISED probabilities p = exp(s - m) (times the dropout multiplier) rounded to bf16 as
    the MFMA operand, the fp32 accumulator rescaled when the maximum moves, one division by the unrounded row sum at the end.
    Mathematically softmax(score) @ v; numerically the roundings land where the kernel's do."""
    B, Tq, Tk = score.shape
    m = torch.full((B, Tq, 1), -float("inf"), dtype=score.dtype)
    lsum = torch.zeros(B, Tq, 1, dtype=score.dtype)
    o = torch.zeros(B, Tq, v.size(2), dtype=score.dtype)
    for j0 in range(0, Tk + 1, 64):
        k0, k1 = max(j0 - 1, 0), min(j0 + 63, Tk)             # keys whose padded index lies in [j0, j0 + 63]
        if k1 <= k0:
            continue
        sb = score[:, :, k0:k1]
        m_new = torch.maximum(m, sb.max(dim=2, keepdim=True).values)
        m_use = torch.where(torch.isinf(m_new), torch.zeros_like(m_new), m_new)
        alpha = torch.exp(m - m_use)
        pb = torch.exp(sb - m_use)
        lsum = lsum * alpha + pb.sum(dim=2, keepdim=True)
        if mult is not None:
            pb = pb * mult[:, :, k0:k1]
        o = o * alpha + torch.bmm(st.q(pb), v[:, k0:k1])
        m = m_new
    return o / torch.where(lsum > 0, lsum, torch.ones_like(lsum)) * (lsum > 0)


def _mha(sd, pre, inp, key_pad_mask, drop=None, site=0, st=Fp32Storage):
    qkv = st.s(F.linear(st.o(inp), st.q(sd[pre + "qkv_net.weight"]), sd[pre + "qkv_net.bias"]))
    q, k, v = torch.chunk(qkv, 3, dim=2)
    score = st.gq(torch.bmm(q, k.transpose(1, 2)) * (1 / (D_HEAD ** 0.5)))   # dS is rounded on its way into the dQ / dK products
    score = score.masked_fill(key_pad_mask.unsqueeze(1), -float("inf"))
    if st.flash:                                              # the fused flash-style attention of the throughput mode
        mult = drop.prob(site + 0, torch.ones_like(score)) if drop is not None else None
        vec = st.s(_flash_pv(score, v, mult, st))
    else:
        prob = F.softmax(score, dim=2)
        if drop is not None:
            prob = drop.prob(site + 0, prob)                  # dropatt, transformer.py:127
        vec = torch.bmm(prob, v)
    out = F.linear(vec, st.q(sd[pre + "o_net.weight"]))
    if drop is not None:
        out = drop.act(site + 1, out)                         # drop, transformer.py:139
    return st.r(F.layer_norm(st.r(inp + out), (D_MODEL,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"]))


def _conv_ff(sd, pre, inp, drop=None, site=0, st=Fp32Storage):
    core = st.o(inp).transpose(1, 2)
    core = F.conv1d(core, st.q(sd[pre + "CoreNet.0.weight"]), sd[pre + "CoreNet.0.bias"], padding=1)
    core = st.s(F.relu(core))
    core = F.conv1d(core, st.q(sd[pre + "CoreNet.2.weight"]), sd[pre + "CoreNet.2.bias"], padding=1)
    core = core.transpose(1, 2)
    if drop is not None:
        core = drop.act(site + 2, core)                       # CoreNet's trailing nn.Dropout, transformer.py:51
    return st.r(F.layer_norm(st.r(inp + core), (D_MODEL,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"]))


def fft_transformer(sd, pre, dec_inp, seq_lens=None, embed=False, taps=None, drop=None, site=0, st=Fp32Storage):
    if embed:
        inp = F.embedding(dec_inp, sd[pre + "word_emb.weight"], padding_idx=0)
        mask = (dec_inp != 0).unsqueeze(2)
    else:
        inp = dec_inp
        mask = mask_from_lens(seq_lens, inp.size(1)).unsqueeze(2)
    pos = positional_embedding(inp.size(1), D_MODEL, inp.dtype) * mask
    out = st.r(inp + pos)
    if taps is not None:
        taps[pre + "in"] = out
    for i in range(N_LAYERS):
        lp = "%slayers.%d." % (pre, i)
        out = _mha(sd, lp + "dec_attn.", out, ~mask.squeeze(2), drop, site + 4 * i, st)
        out = out * mask
        out = _conv_ff(sd, lp + "pos_ff.", out, drop, site + 4 * i, st)
        out = out * mask
        if taps is not None:
            taps[lp + "out"] = out
    return out, mask


def temporal_predictor(sd, pre, enc_out, enc_mask, drop=None, site=0, st=Fp32Storage):
    out = (enc_out * enc_mask).transpose(1, 2)
    for i in range(2):
        lp = "%slayers.%d." % (pre, i)
        # fp32-stored in both modes; in bf16 mode the conv operands (and, in backward, the pre-activation gradient) are rounded in flight
        out = F.relu(st.gq(F.conv1d(st.q(out), st.q(sd[lp + "conv.weight"]), sd[lp + "conv.bias"], padding=1)))
        C = out.size(1)
        out = F.layer_norm(out.transpose(1, 2), (C,), sd[lp + "norm.weight"], sd[lp + "norm.bias"])
        if drop is not None:
            out = drop.act(site + i, out)                     # ConvReLUNorm dropout, common/layers.py:97
        out = out.transpose(1, 2)
    out = out.transpose(1, 2)
    return F.linear(st.q(out), st.q(sd[pre + "fc.weight"]), sd[pre + "fc.bias"]) * enc_mask


def regulate_len(durations, enc_out, pace=1.0, mel_max_len=None):
    reps = (durations.float() * pace + 0.5).long()
    dec_lens = reps.sum(dim=1)
    max_len = int(dec_lens.max())
    cum = torch.cumsum(F.pad(reps, (1, 0, 0, 0), value=0), dim=1)[:, None, :].to(enc_out.dtype)
    rng = torch.arange(max_len)[None, :, None]
    mult = ((cum[:, :, :-1] <= rng) & (cum[:, :, 1:] > rng)).to(enc_out.dtype)
    enc_rep = torch.matmul(mult, enc_out)
    if mel_max_len is not None:
        enc_rep = enc_rep[:, :mel_max_len]
        dec_lens = torch.clamp_max(dec_lens, mel_max_len)
    return enc_rep, dec_lens


def average_pitch(pitch, durs):
    ends = torch.cumsum(durs, dim=1).long()
    starts = F.pad(ends[:, :-1], (1, 0))
    nz_cums = F.pad(torch.cumsum(pitch != 0.0, dim=2), (1, 0))
    p_cums = F.pad(torch.cumsum(pitch, dim=2), (1, 0))
    bs, l = ends.size()
    nf = pitch.size(1)
    dcs = starts[:, None, :].expand(bs, nf, l)
    dce = ends[:, None, :].expand(bs, nf, l)
    sums = (torch.gather(p_cums, 2, dce) - torch.gather(p_cums, 2, dcs)).float()
    nel = (torch.gather(nz_cums, 2, dce) - torch.gather(nz_cums, 2, dcs)).float()
    return torch.where(nel == 0.0, nel, sums / nel).to(pitch.dtype)


DS_ENC, DS_DEC, DS_PRED = 0, 100, 200      # dropout site (stream) numbering shared with fastpitch_engine.hip


def forward(sd, batch, stage, taps=None, drop=None, storage=None):
    """batch: dict(text (B,Tt) int64, in_lens, mel_tgt (B,80,Tm), mel_lens, pitch (B,1,Tm), energy (B,Tm),
    durs (B,Tt) int).  Returns the reference's 13-slot output list (model.py:388-390).
    storage: None / "fp32" = the reference's arithmetic; "bf16" = the same graph with the HIP throughput mode's rounding points."""
    st = _storage(storage)
    text, mel_lens = batch["text"], batch["mel_lens"]
    mel_max_len = int(mel_lens.max())
    enc_out, enc_mask = fft_transformer(sd, "encoder.", text, embed=True, taps=taps, drop=drop, site=DS_ENC, st=st)
    dur_tgt = batch["durs"]
    if stage == 2:
        log_dur_pred = temporal_predictor(sd, "duration_predictor.", enc_out, enc_mask, drop, DS_PRED + 0, st).squeeze(-1)
        dur_pred = torch.clamp(torch.exp(log_dur_pred) - 1, 0, 75)
        return [None, None, dur_pred, log_dur_pred, None, None, None, None, None, None, dur_tgt, None, batch["in_lens"]]
    pitch_pred = temporal_predictor(sd, "pitch_predictor.", enc_out, enc_mask, drop, DS_PRED + 2, st).permute(0, 2, 1)
    pitch_tgt = average_pitch(batch["pitch"], dur_tgt)
    pitch_emb = F.conv1d(pitch_tgt, sd["pitch_emb.weight"], sd["pitch_emb.bias"], padding=1)
    enc_out = st.r(enc_out + pitch_emb.transpose(1, 2))
    energy_pred = temporal_predictor(sd, "energy_predictor.", enc_out, enc_mask, drop, DS_PRED + 4, st).squeeze(-1)
    energy_tgt = torch.log(1.0 + average_pitch(batch["energy"].unsqueeze(1), dur_tgt))
    energy_emb = F.conv1d(energy_tgt, sd["energy_emb.weight"], sd["energy_emb.bias"], padding=1)
    energy_tgt = energy_tgt.squeeze(1)
    enc_out = st.r(enc_out + energy_emb.transpose(1, 2))
    if taps is not None:
        taps["enc_cond"] = enc_out
    len_regulated, dec_lens = regulate_len(dur_tgt, enc_out, 1.0, mel_max_len)
    dec_out, dec_mask = fft_transformer(sd, "decoder.", len_regulated, seq_lens=dec_lens, taps=taps, drop=drop, site=DS_DEC, st=st)
    mel_out = st.r(F.linear(st.o(dec_out), st.q(sd["proj.weight"]), sd["proj.bias"]))
    return [mel_out, dec_mask, None, None, pitch_pred, pitch_tgt, energy_pred, energy_tgt, None, None, dur_tgt, None,
            batch["in_lens"]]


# ---- stage 1: the aligner (ConvAttention + monotonic alignment search + forward-sum / CTC loss) -------------------------------
def beta_binomial_prior(phoneme_count, mel_count, scaling=1.0):
    """data_function.py:84-94: the (mel_count, phoneme_count) beta-binomial attention prior of one utterance."""
    import numpy as np
    from scipy.stats import betabinom
    x = np.arange(0, phoneme_count)
    rows = [betabinom(phoneme_count, scaling * i, scaling * (mel_count + 1 - i)).pmf(x) for i in range(1, mel_count + 1)]
    return torch.tensor(np.array(rows))


def attn_prior_batch(in_lens, mel_lens, dtype=torch.float32):
    """TTSCollate (data_function.py:600-609): zero-padded (B, max_mel, max_text) stack of the per-item priors."""
    B = in_lens.numel()
    out = torch.zeros(B, int(mel_lens.max()), int(in_lens.max()), dtype=dtype)
    for b in range(B):
        L, M = int(in_lens[b]), int(mel_lens[b])
        out[b, :M, :L] = beta_binomial_prior(L, M).to(dtype)
    return out


def conv_attention(sd, queries, keys, key_pad_mask, attn_prior, pre="attention."):
    """ConvAttention.forward (attention.py:171-220), align_query_enc_type '3xconv'.  queries (B, 80, T1) mel, keys (B, 384, T2)
    text embeddings, key_pad_mask (B, T2) True on padding.  Returns attn (B, 1, T1, T2) and attn_logprob."""
    k = F.conv1d(keys, sd[pre + "key_proj.0.conv.weight"], sd[pre + "key_proj.0.conv.bias"], padding=1)
    k = F.conv1d(F.relu(k), sd[pre + "key_proj.2.conv.weight"], sd[pre + "key_proj.2.conv.bias"])
    q = F.conv1d(queries, sd[pre + "query_proj.0.conv.weight"], sd[pre + "query_proj.0.conv.bias"], padding=1)
    q = F.conv1d(F.relu(q), sd[pre + "query_proj.2.conv.weight"], sd[pre + "query_proj.2.conv.bias"])
    q = F.conv1d(F.relu(q), sd[pre + "query_proj.4.conv.weight"], sd[pre + "query_proj.4.conv.bias"])
    attn = (q[:, :, :, None] - k[:, :, None]) ** 2
    attn = -0.0005 * attn.sum(1, keepdim=True)
    attn = F.log_softmax(attn, dim=3) + torch.log(attn_prior[:, None] + 1e-8)
    attn_logprob = attn.clone()
    attn = attn.masked_fill(key_pad_mask[:, None, None, :], -float("inf"))
    return F.softmax(attn, dim=3), attn_logprob


def mas_width1(attn_map):
    """alignment.py:76-104 (numpy): monotonic alignment search over a (mel, text) probability map; ties prefer the diagonal."""
    import numpy as np
    opt = np.zeros_like(attn_map)
    with np.errstate(divide="ignore"):
        lm = np.log(attn_map)
    lm[0, 1:] = -np.inf
    log_p = np.zeros_like(lm)
    log_p[0, :] = lm[0, :]
    prev_ind = np.zeros(lm.shape, dtype=np.int64)
    for i in range(1, lm.shape[0]):
        prev = log_p[i - 1]
        shifted = np.concatenate([[-np.inf], prev[:-1]])
        take_diag = shifted >= prev
        take_diag[0] = False
        log_p[i] = lm[i] + np.where(take_diag, shifted, prev)
        prev_ind[i] = np.arange(lm.shape[1]) - take_diag.astype(np.int64)
    cur = lm.shape[1] - 1
    for i in range(lm.shape[0] - 1, -1, -1):
        opt[i, cur] = 1
        cur = prev_ind[i, cur]
    opt[0, cur] = 1
    return opt


def binarize_attention(attn, in_lens, out_lens):
    """model.py:283-295 (b_mas): hard alignment of every item, zero outside its (out_len, in_len) block."""
    a = attn.detach().cpu().numpy()
    import numpy as np
    out = np.zeros_like(a)
    for b in range(a.shape[0]):
        L, M = int(in_lens[b]), int(out_lens[b])
        out[b, 0, :M, :L] = mas_width1(a[b, 0, :M, :L])
    return torch.from_numpy(out)


def attention_ctc_loss(attn_logprob, in_lens, out_lens, blank_logprob=-1.0):
    """AttentionCTCLoss (attn_loss_function.py:20-44): per item, log-softmax over [blank, keys 1..L] and nn.CTCLoss(mean) against the
    identity target 1..L; averaged over the batch."""
    padded = F.pad(attn_logprob, (1, 0, 0, 0, 0, 0, 0, 0), value=blank_logprob)
    total = 0.0
    for b in range(attn_logprob.shape[0]):
        L, M = int(in_lens[b]), int(out_lens[b])
        target = torch.arange(1, L + 1).unsqueeze(0)
        lp = padded[b].permute(1, 0, 2)[:M, :, :L + 1]
        lp = F.log_softmax(lp[None], dim=3)[0]
        total = total + F.ctc_loss(lp, target, input_lengths=torch.tensor([M]), target_lengths=torch.tensor([L]), zero_infinity=True)
    return total / attn_logprob.shape[0]


def forward_stage1(sd, batch):
    """FastPitch.forward with training_stage == 1 (model.py:346-360) -> (attn_hard_dur, attn_soft, attn_hard, attn_logprob)."""
    text = batch["text"]
    text_emb = F.embedding(text, sd["encoder.word_emb.weight"], padding_idx=0)
    key_pad = ~mask_from_lens(batch["in_lens"], text.size(1))
    attn_soft, attn_logprob = conv_attention(sd, batch["mel_tgt"], text_emb.permute(0, 2, 1), key_pad, batch["attn_prior"])
    attn_hard = binarize_attention(attn_soft, batch["in_lens"], batch["mel_lens"])
    attn_hard_dur = attn_hard.sum(2)[:, 0, :]
    return attn_hard_dur, attn_soft, attn_hard, attn_logprob


def loss_stage1(attn_logprob, batch, attn_loss_scale=1.0):
    """FastPitchLoss.forward, training_stage == 1 (loss_function.py:73-81)."""
    return attention_ctc_loss(attn_logprob, batch["in_lens"], batch["mel_lens"]) * attn_loss_scale


def infer(sd, text, pace=1.0, max_duration=75):
    """FastPitch.infer (model.py:426-481) with predicted durations / pitch / energy, no speaker embedding, no pitch transform.
    text: (B, Tt) int64, zero-padded.  Returns (mel_out (B, 80, Tm), dec_lens, dur_pred, pitch_pred (B, 1, Tt), energy_pred)."""
    enc_out, enc_mask = fft_transformer(sd, "encoder.", text, embed=True)
    log_dur_pred = temporal_predictor(sd, "duration_predictor.", enc_out, enc_mask).squeeze(-1)
    dur_pred = torch.clamp(torch.exp(log_dur_pred) - 1, 0, max_duration)
    pitch_pred = temporal_predictor(sd, "pitch_predictor.", enc_out, enc_mask).permute(0, 2, 1)
    enc_out = enc_out + F.conv1d(pitch_pred, sd["pitch_emb.weight"], sd["pitch_emb.bias"], padding=1).transpose(1, 2)
    energy_pred = temporal_predictor(sd, "energy_predictor.", enc_out, enc_mask).squeeze(-1)
    enc_out = enc_out + F.conv1d(energy_pred.unsqueeze(1), sd["energy_emb.weight"], sd["energy_emb.bias"], padding=1).transpose(1, 2)
    len_regulated, dec_lens = regulate_len(dur_pred, enc_out, pace, None)
    dec_out, dec_mask = fft_transformer(sd, "decoder.", len_regulated, seq_lens=dec_lens)
    mel_out = F.linear(dec_out, sd["proj.weight"], sd["proj.bias"]).permute(0, 2, 1)
    return mel_out, dec_lens, dur_pred, pitch_pred, energy_pred


def loss(model_out, batch, stage, dur_scale=0.1, pitch_scale=0.1, energy_scale=0.1):
    """FastPitchLoss.forward with the trainer's scales (xva_train.py:702-704; energy default loss_function.py:54).
    Returns (loss, dict of component losses)."""
    (mel_out, _, _, log_dur_pred, pitch_pred, pitch_tgt, energy_pred, energy_tgt, _, _, dur_tgt, _, in_lens) = model_out
    mel_tgt = batch["mel_tgt"]
    max_inp = int(batch["text"].size(1))
    dur_mask = mask_from_lens(in_lens, max_inp)
    z = torch.zeros((), dtype=mel_tgt.dtype)
    dur_l = mel_l = pitch_l = energy_l = z
    if stage == 2:
        log_dur_tgt = torch.log(dur_tgt.to(mel_tgt.dtype) + 1)
        dl = F.mse_loss(log_dur_pred, log_dur_tgt, reduction="none")
        dur_l = (dl * dur_mask).sum() / dur_mask.sum()
    if stage in (3, 4):
        mt = mel_tgt.transpose(1, 2)
        ldiff = mt.size(1) - mel_out.size(1)
        mo = F.pad(mel_out, (0, 0, 0, ldiff, 0, 0), value=0.0)
        mm = mt.ne(0).to(mt.dtype)
        mel_l = (F.mse_loss(mo, mt, reduction="none") * mm).sum() / mm.sum()
        if stage == 3:
            pl = F.mse_loss(pitch_tgt, pitch_pred, reduction="none")
            pitch_l = (pl * dur_mask.unsqueeze(1)).sum() / dur_mask.sum()
            el = F.mse_loss(energy_tgt, energy_pred, reduction="none")
            energy_l = (el * dur_mask).sum() / dur_mask.sum()
    total = mel_l + dur_l * dur_scale + pitch_l * pitch_scale + energy_l * energy_scale
    return total, {"mel": mel_l, "dur": dur_l, "pitch": pitch_l, "energy": energy_l}


def trainable_names(sd_keys, stage):
    """Which parameters train in each stage (freezing in xva_train.py:589-672, summarised SURVEY.md Appendix A)."""
    def frozen(k):
        grp = k.split(".")[0]
        if stage == 2:
            return grp in ("attention", "decoder", "pitch_predictor", "pitch_emb", "energy_predictor", "proj")
        if stage == 3:
            return grp in ("attention", "duration_predictor")
        if stage == 4:
            return grp in ("attention", "duration_predictor", "pitch_predictor", "pitch_emb", "energy_predictor")
        return False
    buffers = ("pitch_mean", "pitch_std")
    return [k for k in sd_keys if k not in buffers and not k.endswith("inv_freq") and not frozen(k)]


def adjust_learning_rate(total_iter, learning_rate=0.1, warmup_iters=1000):
    if warmup_iters == 0:
        scale = 1.0
    elif total_iter > warmup_iters:
        scale = 1.0 / (total_iter ** 0.5)
    else:
        scale = total_iter / (warmup_iters ** 1.5)
    return learning_rate * scale


def clip_grad_norm_(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (xva_train.py:857,861): global L2 norm, coef = max_norm/(norm+1e-6) clamped to 1."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total


def lamb_step(params, grads, state, lr, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6):
    """Lamb.step (lamb.py:40-106): no bias correction, weight-norm clamp 10, trust ratio 1 when either norm is 0.
    params/grads/state are dicts keyed by name; state[name] = dict(exp_avg, exp_avg_sq, step)."""
    b1, b2 = betas
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        st = state.setdefault(k, {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)})
        st["step"] += 1
        st["exp_avg"].mul_(b1).add_(g, alpha=1 - b1)
        st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1 - b2)
        weight_norm = p.pow(2).sum().sqrt().clamp(0, 10)
        adam_step = st["exp_avg"] / st["exp_avg_sq"].sqrt().add(eps)
        if weight_decay != 0:
            adam_step.add_(p, alpha=weight_decay)
        adam_norm = adam_step.pow(2).sum().sqrt()
        trust = 1.0 if (weight_norm == 0 or adam_norm == 0) else (weight_norm / adam_norm).item()
        st["weight_norm"], st["adam_norm"], st["trust_ratio"] = weight_norm, adam_norm, trust
        p.add_(adam_step, alpha=-lr * trust)


# ---------------------------------------------------------------- synthetic inputs ----
def synth_batch(B, T_text, T_mel, seed, ragged=True, dtype=torch.float32):
    """Seeded FastPitch batch shaped like TTSCollate's output (data_function.py:565-695): text sorted by length desc,
    zero padding, durations summing to each item's mel length, pitch with unvoiced zeros, energy = ||mel||_2."""
    g = torch.Generator().manual_seed(seed)
    if ragged and B > 1:
        in_lens = torch.sort(torch.randint(max(2, T_text // 3), T_text + 1, (B,), generator=g), descending=True).values
        in_lens[0] = T_text
    else:
        in_lens = torch.full((B,), T_text, dtype=torch.long)
    text = torch.zeros(B, T_text, dtype=torch.long)
    durs = torch.zeros(B, T_text, dtype=torch.long)
    mel_lens = torch.zeros(B, dtype=torch.long)
    for b in range(B):
        L = int(in_lens[b])
        text[b, :L] = torch.randint(1, N_SYMBOLS, (L,), generator=g)
        tm = T_mel if b == 0 else max(L, int(T_mel * L / T_text))
        tm = max(tm, L)
        extra = torch.multinomial(torch.ones(L), tm - L, replacement=True, generator=g) if tm > L else torch.zeros(0, dtype=torch.long)
        d = torch.ones(L, dtype=torch.long)
        d.scatter_add_(0, extra, torch.ones_like(extra))
        durs[b, :L] = d
        mel_lens[b] = tm
    Tm = int(mel_lens.max())
    mel = torch.zeros(B, N_MEL, Tm, dtype=dtype)
    pitch = torch.zeros(B, 1, Tm, dtype=dtype)
    for b in range(B):
        tm = int(mel_lens[b])
        mel[b, :, :tm] = torch.clamp(torch.randn(N_MEL, tm, generator=g) * 2 - 5, math.log(1e-5), 2.0).to(dtype)
        p = torch.randn(tm, generator=g)
        p[torch.rand(tm, generator=g) < 0.3] = 0.0
        pitch[b, 0, :tm] = p.to(dtype)
    energy = torch.norm(mel.float(), dim=1, p=2).to(dtype)
    for b in range(B):
        energy[b, int(mel_lens[b]):] = 0
    return {"text": text, "in_lens": in_lens, "mel_tgt": mel, "mel_lens": mel_lens, "pitch": pitch, "energy": energy,
            "durs": durs}


def init_state_dict(seed, dtype=torch.float32):
    """Random-init state_dict with the reference's 185 keys/shapes (FastPitch.__init__, model.py:125-265), using torch's
    default initialisers' scale (values only need to be plausible: there are no checkpoints offline)."""
    g = torch.Generator().manual_seed(seed)

    def U(shape, fan_in):
        bound = 1.0 / math.sqrt(fan_in)
        return ((torch.rand(shape, generator=g) * 2 - 1) * bound).to(dtype)

    sd = {"pitch_mean": torch.zeros(1, dtype=dtype), "pitch_std": torch.zeros(1, dtype=dtype)}
    emb = torch.randn(N_SYMBOLS, D_MODEL, generator=g).to(dtype)
    emb[0] = 0
    for name in ("encoder", "decoder"):
        if name == "encoder":
            sd["encoder.word_emb.weight"] = emb
        sd[name + ".pos_emb.inv_freq"] = (1 / (10000 ** (torch.arange(0.0, D_MODEL, 2.0) / D_MODEL))).to(dtype)
        for i in range(N_LAYERS):
            p = "%s.layers.%d." % (name, i)
            sd[p + "dec_attn.qkv_net.weight"] = U((192, 384), 384)
            sd[p + "dec_attn.qkv_net.bias"] = U((192,), 384)
            sd[p + "dec_attn.o_net.weight"] = U((384, 64), 64)
            sd[p + "dec_attn.layer_norm.weight"] = (1 + 0.1 * torch.randn(384, generator=g)).to(dtype)
            sd[p + "dec_attn.layer_norm.bias"] = (0.1 * torch.randn(384, generator=g)).to(dtype)
            sd[p + "pos_ff.CoreNet.0.weight"] = U((1536, 384, 3), 384 * 3)
            sd[p + "pos_ff.CoreNet.0.bias"] = U((1536,), 384 * 3)
            sd[p + "pos_ff.CoreNet.2.weight"] = U((384, 1536, 3), 1536 * 3)
            sd[p + "pos_ff.CoreNet.2.bias"] = U((384,), 1536 * 3)
            sd[p + "pos_ff.layer_norm.weight"] = (1 + 0.1 * torch.randn(384, generator=g)).to(dtype)
            sd[p + "pos_ff.layer_norm.bias"] = (0.1 * torch.randn(384, generator=g)).to(dtype)
        if name == "encoder":
            _init_predictor(sd, "duration_predictor.", U, g, dtype)
    _init_predictor(sd, "pitch_predictor.", U, g, dtype)
    sd["pitch_emb.weight"] = U((384, 1, 3), 3)
    sd["pitch_emb.bias"] = U((384,), 3)
    _init_predictor(sd, "energy_predictor.", U, g, dtype)
    sd["energy_emb.weight"] = U((384, 1, 3), 3)
    sd["energy_emb.bias"] = U((384,), 3)
    sd["proj.weight"] = U((80, 384), 384)
    sd["proj.bias"] = U((80,), 384)
    sd["attention.query_proj.0.conv.weight"] = U((160, 80, 3), 240)
    sd["attention.query_proj.0.conv.bias"] = U((160,), 240)
    sd["attention.query_proj.2.conv.weight"] = U((80, 160, 1), 160)
    sd["attention.query_proj.2.conv.bias"] = U((80,), 160)
    sd["attention.query_proj.4.conv.weight"] = U((80, 80, 1), 80)
    sd["attention.query_proj.4.conv.bias"] = U((80,), 80)
    sd["attention.attn_proj.weight"] = U((1, 80, 1, 1), 80)
    sd["attention.attn_proj.bias"] = U((1,), 80)
    sd["attention.key_proj.0.conv.weight"] = U((768, 384, 3), 1152)
    sd["attention.key_proj.0.conv.bias"] = U((768,), 1152)
    sd["attention.key_proj.2.conv.weight"] = U((80, 768, 1), 768)
    sd["attention.key_proj.2.conv.bias"] = U((80,), 768)
    return sd


def _init_predictor(sd, p, U, g, dtype):
    sd[p + "layers.0.conv.weight"] = U((256, 384, 3), 1152)
    sd[p + "layers.0.conv.bias"] = U((256,), 1152)
    sd[p + "layers.0.norm.weight"] = (1 + 0.1 * torch.randn(256, generator=g)).to(dtype)
    sd[p + "layers.0.norm.bias"] = (0.1 * torch.randn(256, generator=g)).to(dtype)
    sd[p + "layers.1.conv.weight"] = U((256, 256, 3), 768)
    sd[p + "layers.1.conv.bias"] = U((256,), 768)
    sd[p + "layers.1.norm.weight"] = (1 + 0.1 * torch.randn(256, generator=g)).to(dtype)
    sd[p + "layers.1.norm.bias"] = (0.1 * torch.randn(256, generator=g)).to(dtype)
    sd[p + "fc.weight"] = U((1, 256), 256)
    sd[p + "fc.bias"] = U((1,), 256)


def train_step(sd, batch, stage, opt_state, total_iter, grad_clip=1000.0, gam=1):
    """One reference optimizer step (xva_train.py:780-862, AMP off): fwd, loss/gam, bwd, clip 1000, LAMB.
    Mutates sd (trainable tensors) and opt_state in place.  Returns (loss value, components, grads dict)."""
    names = trainable_names(sd.keys(), stage)
    leaves = {k: sd[k].detach().clone().requires_grad_(True) for k in names}
    work = dict(sd)
    work.update(leaves)
    out = forward(work, batch, stage)
    total, comps = loss(out, batch, stage)
    (total / gam).backward()
    grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    clip_grad_norm_(list(grads.values()), grad_clip)
    lr = adjust_learning_rate(total_iter)
    with torch.no_grad():
        params = {k: sd[k] for k in grads}
        lamb_step(params, grads, opt_state, lr)
    return float(total), {k: float(v) for k, v in comps.items()}, grads
