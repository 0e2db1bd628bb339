"""The acoustic half of xVAPitch's generator step on libxvahip — `xVAPitch.train_step` (python/xvapitch/model.py:681-870) up to, and without,
the waveform decoder, followed by the KL and duration terms of VitsGeneratorLoss.forward (python/xvapitch/losses.py:213-220):

    language / symbol embeddings -> TextEncoder (RelativePositionTransformer + `proj` statistics, model.py:1140-1170)
    linear spectrogram -> PosteriorEncoder -> z ; ResidualCouplingBlocks: z -> z_p
    monotonic alignment search over the prior log-likelihoods (:763-776) -> durations -> StochasticDurationPredictor NLL (:792-814)
    prior expansion along the path (:846-847) ; kl_loss(z_p, logs_q, m_p, logs_p)

Built for the reference's default switches (xva_train.py:1098-1120: --energy / --flc / --ow_flow / --mltts_rc 0; detach_dp_input True,
model.py:52; lang_w 1), with --pitch 0 (the argparse default) or 1 (what the shipped trainer sets, xva_train.py:1421-1425: pitch_emb subtracted
from z_p :752-755, average_pitch targets :817-829, the pitch predictor :836, the pitch term of losses.py:224-241).  Dropout: `dropout_p` is the
text encoder's and the pitch predictor's (0.1: model.py:88,166), `sdp_dropout_p` the duration predictor's (0.5: model.py:128) — at the
reference's sites (transformer.py / sdp.py), masks from a keyed hash under a seed that advances with every training forward; both default to 0
(the parity goldens run the reference in eval mode), the trainer passes the reference's values.  The waveform decoder + discriminator branch (:852-853) is the HiFi-GAN path (xva-trainer_amd/hifigan); its speaker-conditioned
generator variant is not built, so this class returns z and the acoustic losses, not the full generator loss.

state_dict keys are the reference's (`emb_l.weight`, `text_encoder.*`, `posterior_encoder.*`, `flow.flows.i.*`, `duration_predictor.*`).
Every matrix product, convolution, normalisation, spline, MAS and KL step is a libxvahip call (through the block classes of wn.py /
transformer.py / sdp.py / ops.py, plus the two batched GEMM forms below); torch supplies the embedding gathers, the speaker-vector
normalisation, concatenations / transposes and the elementwise preparation of the MAS operands."""
import math
import os

import torch
import torch.nn.functional as F

from .. import _lib
from . import ops
from .sdp import Conv1x1, Mask, StochasticDurationPredictor, _param
from .transformer import RelativePositionTransformer
from .wn import PosteriorEncoder, ResidualCouplingBlocks


_lib.lib.xva_fp_avg_pitch.restype = __import__("ctypes").c_int32
_lib.lib.xva_fp_avg_pitch.argtypes = [__import__("ctypes").c_void_p] * 3 + [__import__("ctypes").c_int32] * 4 + [__import__("ctypes").c_void_p]


def _pad4(n):
    return (n + 3) // 4 * 4


def prior_logp(stats, z_p, Cc):
    """The four log-likelihood terms of model.py:765-771 as ONE batched xva_gemm: with s = exp(-2 logs_p),
    logp[b, i, j] = [s | m s](b, i, :) . [-z^2 / 2 | z](b, j, :) + sum_c(-log(2 pi) / 2 - logs_p - m^2 s / 2)(b, i).
    stats (B, Tt, 2C) = [m_p | logs_p] time-major, z_p (B, C, Ty).  Returns (B, Tt, Ty)."""
    B, Tt, _ = stats.shape
    Ty = z_p.size(2)
    Typ = _pad4(Ty)
    m_p, logs_p = stats[..., :Cc], stats[..., Cc:]
    s = torch.exp(-2.0 * logs_p)
    A = torch.cat([s, m_p * s], -1).contiguous()
    zt = z_p.transpose(1, 2)
    Bm = torch.zeros(B, Typ, 2 * Cc, device=stats.device)
    Bm[:, :Ty, :Cc] = -0.5 * zt * zt
    Bm[:, :Ty, Cc:] = zt
    row = (-0.5 * math.log(2 * math.pi) - logs_p - 0.5 * m_p * m_p * s).sum(-1)
    logp = torch.empty(B, Tt, Typ, device=stats.device)
    _lib.gemm(A, Bm, logp, Tt, Typ, 2 * Cc, 2 * Cc, 2 * Cc, Typ, layout=_lib.GEMM_NT, compute=0, batch=B, sA=Tt * 2 * Cc, sB=Typ * 2 * Cc, sC=Tt * Typ)
    return logp[..., :Ty] + row.unsqueeze(-1)


class _Expand(torch.autograd.Function):
    """model.py:846-847, both einsums at once: out[b, j, :] = sum_i attn[b, i, j] stats[b, i, :] (the path is constant).  attn (B, Tt, Typ) with
    zero columns past Ty; stats (B, Tt, 2C); out (B, Typ, 2C)."""
    @staticmethod
    def forward(ctx, stats, attn):
        B, Tt, W = stats.shape
        Typ = attn.size(2)
        stats = stats.contiguous()
        out = torch.empty(B, Typ, W, device=stats.device)
        _lib.gemm(attn, stats, out, Typ, W, Tt, Typ, W, W, layout=_lib.GEMM_TN, compute=0, batch=B, sA=Tt * Typ, sB=Tt * W, sC=Typ * W)
        ctx.save_for_backward(attn)
        return out

    @staticmethod
    def backward(ctx, d_out):
        (attn,) = ctx.saved_tensors
        B, Tt, Typ = attn.shape
        d_out = d_out.contiguous()
        W = d_out.size(2)
        d_stats = torch.empty(B, Tt, W, device=d_out.device)
        _lib.gemm(attn, d_out, d_stats, Tt, W, Typ, Typ, W, W, layout=_lib.GEMM_NN, compute=0, batch=B, sA=Tt * Typ, sB=Typ * W, sC=Tt * W)
        return d_stats, None


def _cross(stream, *tensors):
    """Tensors that were allocated on one stream and are about to be read on `stream`: tell the caching allocator, so that a block freed while that
    stream's reader is still queued is not handed out again before it has run."""
    for t in tensors:
        if t is not None and t.is_cuda:
            t.record_stream(stream)


class _JoinStreams(torch.autograd.Function):
    """Identity on the loss.  Its backward is the first node of the backward pass: it queues the end-of-pass join — the modules' backward passes run on the
    streams of their forward passes and most of them write their parameter gradients themselves (no AccumulateGrad the engine would wait for), so the
    caller's stream waits for every one of those streams when backward() returns."""
    @staticmethod
    def forward(ctx, loss, streams):
        ctx.streams = streams
        return loss.view_as(loss)

    @staticmethod
    def backward(ctx, g):
        streams = ctx.streams

        def join():
            cur = torch.cuda.current_stream()
            for s in streams:
                cur.wait_stream(s)
        torch.autograd.Variable._execution_engine.queue_callback(join)
        return g, None


def _sub_compute(compute, env):
    """Throughput mode of the two RelativePositionTransformers (text encoder, pitch predictor): "split" — fp32 storage, every product three bf16 MFMAs on
    hi + lo split operands (xva_gemm compute 2, ~1e-5 per product).  With plain bf16-rounded operands ("mixed", rounds 2 - 4) the pitch loss sat at 9.2e-4
    of the reference's against the 1e-3 bound; measured on the reference golden in round 5 (bench.py xvapitch_c5.parity): text encoder split 5.9e-4, both
    split 5.1e-6 (loss_kl 3.8e-4 -> 1.6e-4, loss_duration 5.5e-5 -> 2.4e-7) for +0.2 ms of a 24 ms iteration — both stacks run on side streams.
    XVA_C5_TEXT_COMPUTE / XVA_C5_PITCH_COMPUTE = mixed | split | fp32 override."""
    if compute != "bf16":
        return "fp32"
    return os.environ.get(env, "split")


class AcousticTrainPath:
    """Constructor arguments follow model.py:55-135 (defaults = the reference's non-`big` model).  compute: "fp32" = exact-fp32 products everywhere
    (the parity mode) ; "bf16" = the throughput mode: WaveNet stacks (posterior encoder, flow) bf16-stored with bf16 MFMA, the two transformers' projections
    and feed-forward convolutions bf16 MFMA on fp32-stored tensors; attention, LayerNorm, the duration predictor, MAS and the losses stay fp32."""

    def __init__(self, n_vocab, num_languages, latent_size=192, embedded_language_dim=4, d_vector_dim=512, hidden_channels_ffn=768, num_heads=2,
                 text_layers=10, posterior_layers=16, flow_layers=4, num_flows=4, spec_bins=513, pitch=False, pe_scaling=0.1, device="cuda", compute="fp32",
                 seed=0, dropout_p=0.0, sdp_dropout_p=0.0):
        Cc, L = latent_size, embedded_language_dim
        self.training, self.drop_seed, self._drop_calls = True, (int(seed) * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF, 0
        self.C, self.L = Cc, L
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(seed)
        self.p = {"emb_l.weight": _param(torch.randn(num_languages, L, generator=gen), self.device),
                  "text_encoder.emb.weight": _param(torch.randn(n_vocab, Cc, generator=gen) * Cc ** -0.5, self.device),                 # model.py:1120
                  "text_encoder.proj.weight": _param((torch.rand(2 * Cc, Cc + L, 1, generator=gen) * 2 - 1) * (Cc + L) ** -0.5, self.device),
                  "text_encoder.proj.bias": _param((torch.rand(2 * Cc, generator=gen) * 2 - 1) * (Cc + L) ** -0.5, self.device)}
        self.encoder = RelativePositionTransformer(Cc + L, Cc + L, Cc + L, hidden_channels_ffn, num_heads, text_layers, kernel_size=3, dropout_p=dropout_p,
                                                   layer_norm_type="2", rel_attn_window_size=4, device=device, seed=seed + 1,
                                                   compute=_sub_compute(compute, "XVA_C5_TEXT_COMPUTE"), dropout_site_base=1000)
        self.posterior_encoder = PosteriorEncoder(spec_bins, Cc, Cc, 5, 1, posterior_layers, cond_channels=d_vector_dim, device=device, compute=compute,
                                                  seed=seed + 2)
        self.flow = ResidualCouplingBlocks(Cc, Cc, 5, 1, flow_layers, num_flows=num_flows, cond_channels=d_vector_dim, device=device, compute=compute,
                                           seed=seed + 3)
        self.duration_predictor = StochasticDurationPredictor(Cc, Cc, 3, sdp_dropout_p, 4, cond_channels=d_vector_dim, language_emb_dim=L, device=device,
                                                              seed=seed + 4, dropout_site_base=3000)
        self._subs = [("text_encoder.encoder.", self.encoder), ("posterior_encoder.", self.posterior_encoder), ("flow.", self.flow),
                      ("duration_predictor.", self.duration_predictor)]
        # --pitch 1, what the shipped trainer sets (xva_train.py:1421-1425): model.py:153-176
        self.pitch, self.pe_scaling, self.Dv = bool(pitch), float(pe_scaling), d_vector_dim
        if self.pitch:
            hid = Cc + L + d_vector_dim                                                                                                   # model.py:1283-1284
            self.pitch_predictor = RelativePositionTransformer(hid, 1, hid, hidden_channels_ffn, num_heads, 3, kernel_size=3, dropout_p=dropout_p,
                                                               layer_norm_type="2", rel_attn_window_size=4, device=device, seed=seed + 5,
                                                               compute=_sub_compute(compute, "XVA_C5_PITCH_COMPUTE"), dropout_site_base=2000)
            self._subs.append(("pitch_predictor.encoder.", self.pitch_predictor))
            self.p["pitch_emb.weight"] = _param((torch.rand(Cc, 1, 3, generator=gen) * 2 - 1) * 3 ** -0.5, self.device)
            self.p["pitch_emb.bias"] = _param((torch.rand(Cc, generator=gen) * 2 - 1) * 3 ** -0.5, self.device)

    def _streams(self, device):
        """(text encoder, duration predictor, pitch predictor) streams; XVA_C5_STREAMS=0: the current stream three times (one code path, no concurrency)."""
        cur = torch.cuda.current_stream(device)
        if os.environ.get("XVA_C5_STREAMS", "1") == "0":
            return cur, cur, cur
        key = torch.device(device).index or 0
        pool = self.__dict__.setdefault("_stream_pool", {})
        if key not in pool:
            pool[key] = tuple(torch.cuda.Stream(device) for _ in range(3))
        return pool[key]

    # ---- dropout (nn.Module.train / eval; the masks' seed) ----
    def _droppers(self):
        return [self.encoder, self.duration_predictor] + ([self.pitch_predictor] if self.pitch else [])

    def train(self, mode=True):
        self.training = bool(mode)
        for m in self._droppers():
            m.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def set_dropout_seed(self, seed):
        """Base seed of the dropout masks (the trainer derives it from its run seed); every training forward uses base + its call count, so
        iterations differ and a resumed run can be replayed."""
        self.drop_seed, self._drop_calls = int(seed) & 0xFFFFFFFFFFFFFFFF, 0

    # ---- reference state_dict ----
    def state_dict(self):
        sd = {k: v.detach().clone() for k, v in self.p.items()}
        for pre, m in self._subs:
            sd.update({pre + k: v for k, v in m.state_dict().items()})
        return sd

    def load_state_dict(self, sd):
        """Reference keys; buffers the reference registers beside the parameters (none are read by this path) are ignored."""
        with torch.no_grad():
            for k, t in self.p.items():
                t.copy_(sd[k].to(device=self.device, dtype=torch.float32))
        for pre, m in self._subs:
            own = set(m.state_dict())
            m.load_state_dict({k[len(pre):]: v.to(self.device) for k, v in sd.items() if k.startswith(pre) and k[len(pre):] in own})

    def grads(self):
        g = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in self.p.items()}
        for pre, m in self._subs:
            if hasattr(m, "grads"):
                g.update({pre + k: v for k, v in m.grads().items()})
            else:
                g.update({pre + k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in m.p.items()})
        return g

    def param_grad_pairs(self):
        """[(parameter tensor, its gradient tensor)] over the tensors the modules actually hold (not the clones of state_dict()): what an in-place
        optimiser steps.  Leaf parameters without a gradient yet are paired with None."""
        def conv(c):
            return [(c.p[n], c.g[n]) for n in c.p]

        def wn(w):
            return [pg for _, c in w._named() for pg in conv(c)]
        out = [(v, v.grad) for v in self.p.values()]
        for _, m in self._subs:
            if isinstance(m, RelativePositionTransformer):
                pd, gd = dict(m._named("p")), dict(m._named("g"))
                # out_channels == 1: the last layer's feed-forward network and second LayerNorm never reach the output (transformer.py): no gradient, like
                # the reference's p.grad None
                dead = ("ffn_layers.%d." % (m.L - 1), "norm_layers_2.%d." % (m.L - 1)) if m.Co == 1 else ()
                out += [(pd[k], None if k.startswith(dead) and dead else gd[k]) for k in pd]
            elif isinstance(m, PosteriorEncoder):
                out += conv(m.pre) + wn(m.enc) + conv(m.proj)
            elif isinstance(m, ResidualCouplingBlocks):
                for f in m.flows:
                    out += conv(f.pre) + wn(f.enc) + conv(f.post)
            else:
                out += [(v, v.grad) for v in m.p.values()]
        return out

    def named_param_grads(self):
        """[(reference state_dict key, parameter tensor as the module holds it, callable -> its current gradient or None)] in module order: what
        an optimiser with per-parameter state and a torch-format state_dict needs (train_step.FlatGroupAdamW).  The posterior encoder's input
        convolution is held zero-padded to a multiple of 4 bins (its key's checkpoint shape is narrower; see state_dict())."""
        def conv(pre, c):
            return [(pre + n, c.p[n], (lambda g=c.g[n]: g)) for n in c.p]

        def wn(pre, w):
            return [e for sub, c in w._named() for e in conv(pre + sub, c)]
        out = [(k, v, (lambda v=v: v.grad)) for k, v in self.p.items()]
        for pre, m in self._subs:
            if isinstance(m, RelativePositionTransformer):
                pd, gd = dict(m._named("p")), dict(m._named("g"))
                dead = ("ffn_layers.%d." % (m.L - 1), "norm_layers_2.%d." % (m.L - 1)) if m.Co == 1 else ()
                out += [(pre + k, pd[k], ((lambda: None) if (dead and k.startswith(dead)) else (lambda g=gd[k]: g))) for k in pd]
            elif isinstance(m, PosteriorEncoder):
                out += conv(pre + "pre.", m.pre) + wn(pre + "enc.", m.enc) + conv(pre + "proj.", m.proj)
            elif isinstance(m, ResidualCouplingBlocks):
                for i, f in enumerate(m.flows):
                    out += conv("%sflows.%d.pre." % (pre, i), f.pre) + wn("%sflows.%d.enc." % (pre, i), f.enc) + conv("%sflows.%d.post." % (pre, i), f.post)
            else:
                out += [(pre + k, v, (lambda v=v: v.grad)) for k, v in m.p.items()]
        return out

    def zero_grad(self):
        if getattr(self, "_flat_g", None) is not None:                 # train_step.FlatGroupAdamW moved every gradient into one arena: one memset, and the
            self._flat_g.zero_()                                       # leaf parameters keep their .grad views (autograd accumulates into them in place)
            return
        for v in self.p.values():
            v.grad = None
        for _, m in self._subs:
            if hasattr(m, "zero_grad"):
                m.zero_grad()
            else:
                for v in m.p.values():
                    v.grad = None

    def _pitch_emb(self, pitch):
        """pitch_emb = Conv1d(1, C, 3, padding 1) (model.py:176) as a GEMM over the three taps of each frame (a 4th zero column keeps rows 16 bytes
        wide).  pitch (B, 1, Ty) -> (B, C, Ty)"""
        p, Cc = self.p, self.C
        B, _, Ty = pitch.shape
        pp = F.pad(pitch.float().reshape(B, Ty), (1, 1))
        cols = torch.stack([pp[:, 0:Ty], pp[:, 1:Ty + 1], pp[:, 2:Ty + 2], torch.zeros(B, Ty, device=pitch.device)], -1).contiguous()
        w4 = torch.cat([p["pitch_emb.weight"].reshape(Cc, 3), torch.zeros(Cc, 1, device=pitch.device)], 1).reshape(Cc, 4, 1)
        return Conv1x1.apply(cols, w4, p["pitch_emb.bias"]).transpose(1, 2)

    # ---- model.py:417-599 ----
    @torch.no_grad()
    def infer(self, tokens, d_vector, language_id, decoder, pacing=1.0, durs_only=False, noise=None, noise_scale_dp=0.333, length_scale=1.0,
              max_inference_len=None):
        """`xVAPitch.infer` on the switches xVAPitchModel sets (xva_train.py:1424-1428: --pitch 1, pe_scaling 0.1, --energy / --ow_flow /
        --expanded_flow 0; flc 0, lang_w 1) for ONE utterance: tokens (1, Tt) int64, d_vector (d_vector_dim,), language_id scalar tensor / int.
        text encoder (:438-439) -> duration predictor sampled in reverse with noise_scale_dp (:443, model.py:73) -> w_ceil = ceil(exp(logw) *
        length_scale * pacing) (:445-447; returned as is when durs_only) -> the prior means expanded along the path (:455-458) -> + pitch_emb(the
        pitch prediction expanded by the durations) * pe_scaling (:498-515) -> z_p = m_p (the reference sets inference_noise_scale to 0, :549) ->
        flow in reverse (:592) -> waveform decoder (:597).  noise (1, 2, Tt): the duration predictor's N(0, 1) draw (torch.randn when None).
        Returns the waveform (1, 1, Ty * 256).  Runs in eval mode (no dropout) whatever the training flag says, as the reference's model.eval()."""
        if not self.pitch:
            raise ValueError("AcousticTrainPath.infer: built without the pitch branch (xVAPitchModel sets --pitch 1)")
        _lib.require_cuda(tokens, d_vector)
        p, Cc, L = self.p, self.C, self.L
        B, Tt = tokens.shape
        if B != 1:
            raise ValueError("infer: one utterance at a time (the reference's infer unsqueezes ONE embedding, model.py:420)")
        was = self.training
        self.eval()
        try:
            dev = tokens.device
            g = F.normalize(d_vector.float().reshape(1, -1)).unsqueeze(-1)                                  # _set_cond_input :918
            lid = torch.as_tensor(language_id, device=dev).reshape(1).long()
            lang = F.embedding(lid, p["emb_l.weight"])                                                      # (1, L) :431-432
            x_emb = F.embedding(tokens, p["text_encoder.emb.weight"]) * math.sqrt(Cc)
            x_in = torch.cat([x_emb, lang.unsqueeze(1).expand(B, Tt, L)], -1).transpose(1, 2)
            x_mask = torch.ones(B, 1, Tt, device=dev)
            x = self.encoder(x_in * x_mask, x_mask)                                                         # :438
            stats = Conv1x1.apply(x.transpose(1, 2).contiguous(), p["text_encoder.proj.weight"], p["text_encoder.proj.bias"])   # (1, Tt, 2C) :439
            logw = self.duration_predictor.infer(x, x_mask, g=g, lang_emb=lang.unsqueeze(-1), noise_scale=noise_scale_dp, noise=noise)   # :443
            w_ceil = torch.ceil(torch.exp(logw) * x_mask * length_scale * pacing)                           # :445-447
            if durs_only:
                return w_ceil
            reps = w_ceil.reshape(Tt).long()
            Ty = max(int(reps.sum()), 1)                                                                    # :452
            idx = torch.repeat_interleave(torch.arange(Tt, device=dev), reps)                               # generate_path + matmul :455-458 = a gather
            m_p = torch.zeros(B, Cc, Ty, device=dev)
            m_p[:, :, :idx.numel()] = stats[0, :, :Cc].index_select(0, idx).t().unsqueeze(0)
            pin = torch.cat([x, g.expand(B, self.Dv, Tt)], 1)                                               # :498, model.py:1338-1340
            pitch_pred = self.pitch_predictor(pin * x_mask, x_mask)                                         # (1, 1, Tt)
            pitch_exp = torch.zeros(B, 1, Ty, device=dev)
            pitch_exp[0, 0, :idx.numel()] = pitch_pred.reshape(Tt).index_select(0, idx)                     # expand_pitch_energy :935-958
            m_p = m_p + self._pitch_emb(pitch_exp) * self.pe_scaling                                        # :511-515
            y_mask = torch.ones(B, 1, Ty, device=dev)
            z = self.flow(m_p.contiguous(), y_mask, g=g, reverse=True)                                      # :549-550, :592
            z = (z * y_mask)[:, :, :max_inference_len]
            self.last_infer = {"w_ceil": w_ceil, "z": z, "m_p": m_p, "logw": logw, "pitch_pred": pitch_pred}
            return decoder(z.contiguous(), g)                                                               # :597
        finally:
            self.train(was)

    # ---- model.py:681-870 ----
    def __call__(self, tokens, x_lengths, y, y_lengths, d_vectors, language_ids, eps=None, noise=None, pitch_padded=None, after_posterior=None):
        """tokens (B, Tt) int64, y (B, spec_bins, Ty) linear spectrogram, d_vectors (B, d_vector_dim), language_ids (B,).  eps (B, C, Ty) / noise
        (B, 2, Tt): the N(0, 1) draws of the posterior encoder (model.py:1472) and the duration predictor (sdp.py:281), drawn here when None.
        after_posterior(z): called as soon as the posterior latent has been enqueued (generator_pass.py: the vocoder branch).
        pitch_padded (B, 1, Ty): frame-level pitch (0 = unvoiced), required when built with pitch=True.
        Returns the tensors train_step hands to the loss plus `attn`, `loss_kl`, `loss_duration` (`loss_pitch`, `pitch_tgt`, `pitch_pred`), `loss`."""
        if self.pitch and pitch_padded is None:
            raise ValueError("AcousticTrainPath(pitch=True): pitch_padded is required")
        _lib.require_cuda(y, d_vectors)
        if self.training:
            self._drop_calls += 1
            for m in self._droppers():
                m.set_dropout_seed(self.drop_seed + 0x9E3779B97F4A7C15 * self._drop_calls)
        p, Cc, L = self.p, self.C, self.L
        B, Tt = tokens.shape
        Ty = y.size(2)
        g = F.normalize(d_vectors.float()).unsqueeze(-1)                                                  # _set_cond_input, model.py:918
        lang = F.embedding(language_ids, p["emb_l.weight"])                                               # (B, L) :695-696
        # Four streams (see _streams): the text encoder needs neither the recording nor z, the two predictors read x DETACHED (their backward passes touch
        # nothing but their own parameters) — each is a chain of hundreds of small launches that leaves most of the device idle when it runs alone.
        main = torch.cuda.current_stream(y.device)
        s_text, s_dur, s_pitch = self._streams(y.device)
        forked = s_text != main
        ev_in = main.record_event() if forked else None
        # the text encoder first: the longest chain ahead of the alignment (13 transformer layers of small products), and it needs nothing but the tokens
        if forked:
            s_text.wait_event(ev_in)
            _cross(s_text, tokens, x_lengths, lang, g, pitch_padded)
        with torch.cuda.stream(s_text):
            x_emb = F.embedding(tokens, p["text_encoder.emb.weight"]) * math.sqrt(Cc)                      # :1152
            x_in = torch.cat([x_emb, lang.unsqueeze(1).expand(B, Tt, L)], -1).transpose(1, 2)              # :1158-1163
            x_lens = x_lengths.to(device=y.device, dtype=torch.int32).contiguous()
            x_mask = (torch.arange(Tt, device=y.device)[None, :] < x_lens[:, None]).float().unsqueeze(1)
            xin = x_in * x_mask
            started = self.encoder.start(xin, x_mask)                                                      # forward issued here; its autograd node is created below, AFTER the flow's
        z, m_q, logs_q, y_mask = self.posterior_encoder(y, y_lengths, g=g, eps=eps)                        # :698
        if after_posterior is not None:          # the waveform decoder needs nothing but z: GeneratorPass starts it here, next to the rest of this path
            after_posterior(z)
        z_p = self.flow(z, y_mask, g=g)                                                                    # :723
        with torch.cuda.stream(s_text):
            # autograd issues backward nodes in reverse order of creation: created here, the text encoder's backward (one engine call) is issued BEFORE the
            # flow's and the posterior encoder's and runs next to them instead of after them
            x = self.encoder.attach(xin, x_mask, started)                                                  # (B, C + L, Tt) :1166
            stats = Mask.apply(Conv1x1.apply(x.transpose(1, 2).contiguous(), p["text_encoder.proj.weight"], p["text_encoder.proj.bias"]), x_lens)   # :1148
        if self.pitch:
            if forked:
                s_pitch.wait_stream(s_text)
                s_pitch.wait_event(ev_in)
                _cross(s_pitch, x, x_mask, g)
            with torch.cuda.stream(s_pitch):
                pin = torch.cat([x.detach(), g.expand(B, self.Dv, Tt)], 1)                                 # :836, model.py:1338-1340
                pitch_pred = self.pitch_predictor(pin * x_mask, x_mask)                                    # (B, 1, Tt)
        if self.pitch:                                                                                     # :752-755  z_p -= pitch_emb(pitch) * pe_scaling
            z_p = z_p - self._pitch_emb(pitch_padded) * self.pe_scaling
        if forked:
            main.wait_stream(s_text)
            _cross(main, x, x_mask, x_lens, stats)
        with torch.no_grad():                                                                              # :763-776
            logp = prior_logp(stats.detach(), z_p.detach(), Cc)
            attn_mask = x_mask.squeeze(1).unsqueeze(-1) * y_mask.squeeze(1).unsqueeze(1)
            attn = ops.maximum_path(logp, attn_mask)
            attn_pad = F.pad(attn, (0, _pad4(Ty) - Ty)).contiguous()
        dr = attn.sum(2).unsqueeze(1)                                                                      # :792
        if forked:
            s_dur.wait_stream(main)
            _cross(s_dur, x, x_mask, dr, g, lang)
        with torch.cuda.stream(s_dur):
            nll = self.duration_predictor(x.detach(), x_mask, dr, g=g, lang_emb=lang.detach().unsqueeze(-1), noise=noise)     # :795-803, :722
            loss_duration = (nll / x_mask.sum()).sum()                                                     # :814, losses.py:220
        ex = _Expand.apply(stats, attn_pad)[:, :Ty].transpose(1, 2)                                        # (B, 2C, Ty) :846-847
        m_p, logs_p = ex[:, :Cc].contiguous(), ex[:, Cc:].contiguous()
        loss_kl, _ = ops.kl_loss(z_p, logs_q, m_p, logs_p, y_mask)                                          # losses.py:213
        if self.pitch:
            if forked:
                s_pitch.wait_stream(main)
                _cross(s_pitch, dr, pitch_padded)
            with torch.cuda.stream(s_pitch):
                with torch.no_grad():                                                                      # :817-829 (ceil of a 0 / 1 path sum = the sum)
                    durs = (dr.squeeze(1) * x_mask.squeeze(1)).ceil().to(torch.int32).contiguous()
                    tgt_pad = torch.empty(B, Tt + 2, device=y.device)                                     # the kernel writes FastPitch's padded token rows: [0 | Tt values | 0]
                    _lib.check(_lib.lib.xva_fp_avg_pitch(_lib.ptr(pitch_padded.float().reshape(B, Ty).contiguous()), _lib.ptr(durs), _lib.ptr(tgt_pad), B, Tt, Ty, 0,
                                                         _lib.stream_ptr()), "xva_fp_avg_pitch")
                    pitch_tgt = tgt_pad[:, 1:Tt + 1].contiguous()
                # losses.py:224-241: the reference's mask broadcast makes the "masked mean" the plain sum of squared errors; / B, x 0.1 (:55)
                err = pitch_pred.reshape(B, Tt) - pitch_tgt
                loss_pitch = (err * err).sum() / B * 0.1
        if forked:
            main.wait_stream(s_dur)
            _cross(main, loss_duration)
            if self.pitch:
                main.wait_stream(s_pitch)
                _cross(main, loss_pitch, pitch_tgt, pitch_pred)
        loss = loss_kl + loss_duration
        if self.pitch:
            loss = loss + loss_pitch
        if forked and loss.requires_grad:
            loss = _JoinStreams.apply(loss, (s_text, s_dur, s_pitch))
        out = {"z": z, "m_q": m_q, "logs_q": logs_q, "x": x, "x_mask": x_mask, "y_mask": y_mask, "z_p": z_p, "m_p": m_p, "logs_p": logs_p, "attn": attn,
               "loss_kl": loss_kl, "loss_duration": loss_duration, "loss": loss}
        if self.pitch:
            out.update({"pitch_tgt": pitch_tgt.unsqueeze(1), "pitch_pred": pitch_pred, "loss_pitch": loss_pitch})
        return out
