"""FastPitch — drop-in for python/fastpitch1_1/fastpitch/model.py:125-482 (class FastPitch), stages 1-4, on libxvahip.

Same constructor, same `forward(inputs_x)` 12-tuple in / 13-slot list out, same `training_stage` attribute, same 185-entry
state_dict (keys, shapes, dtypes) — so reference checkpoints load and our checkpoints load into the reference.  Inside,
all 181 parameters are views of ONE flat nn.Parameter (`flat`) that the HIP engine consumes directly; `forward` and
`backward` are one C call each (training stages 1-4 and `infer`).
"""
from collections import OrderedDict

import torch
import torch.nn as nn

from .. import _lib
from . import engine as E
from . import params as P


class FpOutputs(list):
    """The reference's 13-slot output list, plus the handles the fused loss needs (engine workspace + device batch)."""
    engine = None
    batch = None
    stage = None


class _FastPitchFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, module, batch, stage):
        eng = module._get_engine()
        eng.forward(flat.detach(), batch, stage)
        ctx.module, ctx.batch, ctx.stage = module, batch, stage
        o = eng.outputs(batch, stage)
        if stage == 2:
            return o["log_dur_pred"].clone(), o["dur_pred"].clone()
        return o["mel_out"].to(torch.float32, copy=True), o["pitch_pred"].clone(), o["energy_pred"].clone()

    @staticmethod
    def backward(ctx, *gouts):
        module, b, stage = ctx.module, ctx.batch, ctx.stage
        eng = module._get_engine()
        eng._prepare(b.B, b.Tt, b.Tm, stage)

        def put(slot, shape, g, sl):
            d = eng.slot(slot, shape)
            d.zero_()
            if g is not None:
                d[sl] = g.reshape(d[sl].shape) * eng.loss_scale      # ("f16": the fp16 gradient buffers want the loss-scaled seed; 1 otherwise)

        if stage == 2:
            put("D_LOGDUR", (b.B, b.Tt + 2), gouts[0], (slice(None), slice(1, b.Tt + 1)))
        else:
            put("D_MEL", (b.B, b.Tm + 2, 80), gouts[0], (slice(None), slice(1, b.Tm + 1)))
            put("D_PITCH", (b.B, b.Tt + 2), gouts[1], (slice(None), slice(1, b.Tt + 1)))
            put("D_ENERGY", (b.B, b.Tt + 2), gouts[2], (slice(None), slice(1, b.Tt + 1)))
        g = torch.zeros_like(module.flat)
        eng.backward(module.flat.detach(), g, b, stage)
        if eng.loss_scale != 1.0:
            g.mul_(eng.grad_inv_scale)
        return g, None, None, None


class _AlignFn(torch.autograd.Function):
    """Training stage 1 (model.py:346-360): forward = ConvAttention + MAS + the forward-sum loss in one C call; backward = one C call that
    accumulates d(loss)/d(attention.*, encoder.word_emb) into the flat gradient."""

    @staticmethod
    def forward(ctx, flat, module, inputs, input_lens, mel_tgt, mel_lens, attn_prior):
        eng = module._get_engine()
        loss, durs, soft, logp = eng.align_forward(flat.detach(), inputs, input_lens, mel_tgt, mel_lens, attn_prior)
        ctx.module = module
        ctx.mark_non_differentiable(durs, soft, logp)
        return loss, durs, soft, logp

    @staticmethod
    def backward(ctx, g_loss, *unused):
        module = ctx.module
        g = torch.zeros_like(module.flat)
        module._get_engine().align_backward(module.flat.detach(), g, float(g_loss.reshape(-1)[0].item()))
        return g, None, None, None, None, None, None


class FastPitch(nn.Module):
    def __init__(self, logger=None, compute="bf16", p_dropout=0.1):
        """p_dropout: every dropout site of the reference model (p_in_fft_dropout, p_in_fft_dropatt, dur/pitch/energy predictor
        dropout, ... model.py:248-263) is 0.1; it is applied in train() mode and off in eval() mode like nn.Dropout."""
        super().__init__()
        self.logger = logger
        self.compute = compute
        self.p_dropout = float(p_dropout)
        self._table = E.tensor_table()
        total = int(_lib.lib.xva_fp_param_floats())
        self.flat = nn.Parameter(torch.zeros(total))
        P.default_init_(self.flat.data, self._table)
        self.register_buffer("pitch_mean", torch.zeros(1))
        self.register_buffer("pitch_std", torch.zeros(1))
        self.full_train_epochs = torch.tensor(-1)
        self.training_stage = torch.tensor(1)
        self.energy_conditioning = True
        self.speaker_emb = None
        self._engine = None
        self.seed = 1234                 # dropout-mask seed of the engine (the trainer sets 1234 + rank, xva_train.py:294-295)

    # ---- engine plumbing ----
    def _get_engine(self):
        if self._engine is None or self._engine.device != self.flat.device:
            self._engine = E.FastPitchEngine(self.flat.device, self.compute, seed=self.seed)
        self._engine.p_dropout = self.p_dropout if self.training else 0.0
        return self._engine

    def named_tensors(self):
        """Reference-named views (reference shapes; conv weights as permuted views) of the flat parameter."""
        out = OrderedDict()
        for name, off, n, shape, kind in self._table:
            v = self.flat.data[off:off + n]
            out[name] = v.view(shape[0], shape[2], shape[1]).permute(0, 2, 1) if kind == 1 else v.view(shape)
        return out

    # ---- checkpoint format (xva_train.py:1001-1016, 1054-1081) ----
    def state_dict(self, *args, destination=None, prefix="", keep_vars=False):
        sd = OrderedDict() if destination is None else destination
        t = P.from_flat(self.flat.data, self._table)
        inv_freq = (1 / (10000 ** (torch.arange(0.0, 384, 2.0) / 384))).to(self.flat.device)   # buffer is built on the CPU in the reference
        sd[prefix + "pitch_mean"] = self.pitch_mean.detach().clone()
        sd[prefix + "pitch_std"] = self.pitch_std.detach().clone()
        for name in P.reference_param_order(self._table):
            if name == "encoder.layers.0.dec_attn.qkv_net.weight":
                sd[prefix + "encoder.pos_emb.inv_freq"] = inv_freq.clone()
            if name == "decoder.layers.0.dec_attn.qkv_net.weight":
                sd[prefix + "decoder.pos_emb.inv_freq"] = inv_freq.clone()
            sd[prefix + name] = t[name]
        return sd

    def load_state_dict(self, state_dict, strict=True):
        sd = {(k[7:] if k.startswith("module.") else k): v for k, v in state_dict.items()}
        names = {t[0] for t in self._table}
        missing = [n for n in names if n not in sd]
        unexpected = [k for k in sd if k not in names and k not in P.BUFFER_KEYS]
        if strict and (missing or unexpected):
            raise RuntimeError("Error(s) in loading state_dict for FastPitch: missing %s unexpected %s" % (missing, unexpected))
        P.to_flat(sd, [t for t in self._table if t[0] in sd], self.flat.data)
        with torch.no_grad():
            for k in ("pitch_mean", "pitch_std"):
                if k in sd:
                    getattr(self, k).copy_(sd[k].to(getattr(self, k)))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ---- inference (model.py:426-481) ----
    def infer(self, inputs, pace=1.0, dur_tgt=None, pitch_tgt=None, energy_tgt=None, pitch_transform=None, max_duration=75, speaker=0):
        """Same signature and return tuple as the reference: (mel_out (B, 80, T), dec_lens, dur_pred, pitch_pred, energy_pred)."""
        if dur_tgt is not None or pitch_tgt is not None or energy_tgt is not None or pitch_transform is not None or self.speaker_emb is not None:
            raise NotImplementedError("infer supports predicted durations / pitch / energy, single speaker, no pitch transform")
        _lib.require_cuda(inputs, self.flat)
        in_lens = (inputs != 0).sum(dim=1)                     # encoder mask = non-padding symbols (transformer.py:216)
        with torch.no_grad():
            return self._get_engine().infer(self.flat.detach(), inputs, in_lens, pace, float(max_duration))

    # ---- forward (model.py:325-390) ----
    def forward(self, inputs_x, use_gt_pitch=True, use_dur_tgt=False, pace=1.0, max_duration=75):
        (inputs, input_lens, mel_tgt, mel_lens, pitch_dense, energy_dense, speaker, attn_prior, durs_padded, max_inp_lengths,
         max_mel_lengths, audiopaths) = inputs_x
        stage = int(self.training_stage)
        if stage not in (1, 2, 3, 4):
            raise NotImplementedError("training_stage %d is not built on the HIP path (stages 1-4 are)" % stage)
        if not use_gt_pitch or pace != 1.0 or speaker is not None:
            raise NotImplementedError("training forward supports use_gt_pitch=True, pace=1.0, single speaker (the trainer's settings)")
        _lib.require_cuda(inputs, self.flat)
        if stage == 1:   # the aligner: [.., attn_soft, attn_hard, attn_hard_dur, attn_logprob, input_lens] (model.py:357-360)
            if attn_prior is None:
                raise ValueError("training stage 1 needs the beta-binomial attn_prior of the batch (data_function.py:84-94)")
            text = inputs.text if isinstance(inputs, E.DeviceBatch) else inputs
            loss, durs, soft, logp = _AlignFn.apply(self.flat, self, text, input_lens, mel_tgt, mel_lens, attn_prior.to(self.flat.device))
            res = FpOutputs()
            res.engine, res.batch, res.stage = self._get_engine(), None, 1
            res.align_loss = loss
            # attn_hard (the (B, 1, Tm, Tt) 0/1 map) is not materialised: every consumer on the training path uses its column sums
            res.extend([None, None, None, None, None, None, None, None, soft, None, durs.float(), logp, input_lens])
            return res
        b = inputs if isinstance(inputs, E.DeviceBatch) else E.DeviceBatch(inputs, input_lens, mel_tgt, mel_lens, pitch_dense, energy_dense, durs_padded)
        outs = _FastPitchFn.apply(self.flat, self, b, stage)
        eng = self._get_engine()
        o = eng.outputs(b, stage)
        res = FpOutputs()
        res.engine, res.batch, res.stage = eng, b, stage
        if stage == 2:
            log_dur_pred, dur_pred = outs
            res.extend([None, None, dur_pred.detach(), log_dur_pred, None, None, None, None, None, None, durs_padded, None, input_lens])
            return res
        mel_out, pitch_pred, energy_pred = outs
        dec_lens = o["dec_lens"].long()
        dec_mask = (torch.arange(b.Tm, device=dec_lens.device)[None, :] < dec_lens[:, None]).unsqueeze(2)
        res.extend([mel_out, dec_mask, None, None, pitch_pred, o["pitch_tgt"].clone(), energy_pred, o["energy_tgt"].clone(), None, None,
                    durs_padded, None, input_lens])
        return res

