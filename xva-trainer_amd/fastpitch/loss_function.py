"""FastPitchLoss — drop-in for python/fastpitch1_1/fastpitch/loss_function.py:52-154 (stages 1-4) on libxvahip.

Same constructor and `forward(model_out, targets, is_training, meta_agg, training_stage)` -> (loss, meta, [mel, dur, pitch,
energy]) contract.  The masked-MSE terms and their gradients are two fused HIP kernels per term reading the engine's
workspace directly; the four `.item()` syncs of the reference (loss_function.py:154) collapse into one 8-float read-back.
"""
import torch
from torch import nn

from . import model as M


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, outs, weights, *preds):
        eng, b, stage = outs.engine, outs.batch, outs.stage
        dur_w, pitch_w, energy_w = weights
        eng.loss_partials(b, stage)
        losses = eng.loss_grads(b, stage, 1.0, dur_w, pitch_w, energy_w)
        ctx.h = (eng, b, stage)
        return losses.clone()

    @staticmethod
    def backward(ctx, g):
        eng, b, stage = ctx.h
        s = g[0] * eng.grad_inv_scale          # the slots hold loss_scale x seed ("f16" mode); autograd carries the plain gradient
        if stage == 2:
            d = eng.slot("D_LOGDUR", (b.B, b.Tt + 2))[:, 1:b.Tt + 1]
            return None, None, d * s
        dm = eng.slot("D_MEL", (b.B, b.Tm + 2, 80))[:, 1:b.Tm + 1].float()
        dp = eng.slot("D_PITCH", (b.B, b.Tt + 2))[:, 1:b.Tt + 1].unsqueeze(1)
        de = eng.slot("D_ENERGY", (b.B, b.Tt + 2))[:, 1:b.Tt + 1]
        if stage == 4:
            return None, None, dm * s, None, None
        return None, None, dm * s, dp * s, de * s


class FastPitchLoss(nn.Module):
    def __init__(self, dur_predictor_loss_scale=1.0, pitch_predictor_loss_scale=1.0, attn_loss_scale=1.0,
                 energy_predictor_loss_scale=0.1, gpus=[0]):
        super().__init__()
        self.gpus = gpus
        self.dur_predictor_loss_scale = dur_predictor_loss_scale
        self.pitch_predictor_loss_scale = pitch_predictor_loss_scale
        self.energy_predictor_loss_scale = energy_predictor_loss_scale
        self.attn_loss_scale = attn_loss_scale

    def forward(self, model_out, targets, is_training=True, meta_agg="mean", training_stage=1):
        stage = int(training_stage)
        if not isinstance(model_out, M.FpOutputs):
            raise TypeError("FastPitchLoss needs the output list of xva_trainer_amd FastPitch.forward (it reads the engine workspace)")
        if stage == 1:   # loss_function.py:73-81: the forward-sum loss was evaluated with the alignment (one fused pass)
            attn_loss = model_out.align_loss.reshape(())
            loss = attn_loss * self.attn_loss_scale
            meta = {"loss": loss.clone().detach(), "attn_loss": attn_loss.clone().detach()}
            return loss, meta, [None, None, None, None]
        w = (self.dur_predictor_loss_scale, self.pitch_predictor_loss_scale, self.energy_predictor_loss_scale)
        if stage == 2:
            preds = (model_out[3],)
        else:
            preds = (model_out[0], model_out[4], model_out[6])
        losses = _LossFn.apply(model_out, w, *preds)
        loss = losses[0]
        host = losses.detach().cpu()            # one sync for all components
        mel_l, dur_l, pitch_l, energy_l = (float(host[i]) for i in (1, 2, 3, 4))
        z = torch.zeros((), device=loss.device)
        meta = {"loss": loss.detach().clone(), "mel_loss": losses[1].detach(), "duration_predictor_loss": losses[2].detach(),
                "pitch_loss": losses[3].detach(), "attn_loss": z, "energy_loss": losses[4].detach()}
        if meta_agg == "sum":
            bsz = model_out.batch.B
            meta = {k: v * bsz for k, v in meta.items()}
        return loss, meta, [mel_l, dur_l, pitch_l, energy_l]
