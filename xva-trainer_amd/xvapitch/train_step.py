"""The two passes of one xVAPitch training iteration on libxvahip (BASELINE config C5; the optimiser updates are the caller's) —
`xVAPitch.forward(batch, optimizer_idx, ...)`, python/xvapitch/model.py:272-384:

  generator pass (optimizer_idx 0, :273-364)   train_step (generator_pass.py) -> VitsDiscriminator on (generated, real) segments (:313-315) ->
                                               VitsGeneratorLoss.forward's total (python/xvapitch/losses.py:187-300):
                                               loss_kl + loss_feat + loss_mel + loss_gen + loss_duration + loss_pitch
  discriminator pass (optimizer_idx 1, :366-384)  VitsDiscriminator on the cached (generated.detach(), real) segments -> discriminator_loss

The feature loss is evaluated as the reference evaluates it — feature_loss(feats_disc_fake, feats_disc_real) (losses.py:196) against the signature
feature_loss(feats_real, feats_generated) (:64-72) puts the .detach() on the GENERATED features — so it contributes its value to the total but no
gradient to the generator; only loss_gen sends a gradient through the discriminator into the waveform.

    step = XVAPitchStep(GeneratorPass(acoustic, decoder, 32), VitsDiscriminator())
    out = step.generator_pass(tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=...)
    out["loss"].backward()                               # generator gradients: step.gen.acoustic.grads(), step.gen.decoder.grads()
    loss_disc = step.discriminator_pass(out["model_outputs"].detach(), out["waveform_seg"])      # discriminator gradients: step.disc.grads()
"""
import torch


class _Adversarial(torch.autograd.Function):
    """(generated segment, real segment) -> (loss_gen, loss_feat); backward: d loss_gen / d generated (see the module docstring for loss_feat)."""
    @staticmethod
    def forward(ctx, o, wav_seg, disc):
        loss_gen, loss_feat, d_wav = disc.g_pass(wav_seg, o, feature_grad=False)
        ctx.save_for_backward(d_wav)
        ctx.shape = tuple(o.shape)
        return loss_gen, loss_feat

    @staticmethod
    def backward(ctx, g_gen, g_feat):
        (d_wav,) = ctx.saved_tensors
        return (d_wav * g_gen).view(ctx.shape), None, None


class XVAPitchStep:
    def __init__(self, generator_pass, discriminator):
        self.gen, self.disc = generator_pass, discriminator

    def generator_pass(self, tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=None, eps=None, noise=None, slice_ids=None):
        out = self.gen(tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=pitch_padded, eps=eps, noise=noise, slice_ids=slice_ids)
        loss_gen, loss_feat = _Adversarial.apply(out["model_outputs"], out["waveform_seg"], self.disc)          # model.py:313-315, losses.py:195-196
        out.update({"loss_gen": loss_gen, "loss_feat": loss_feat, "loss": out["loss"] + loss_gen + loss_feat})  # losses.py:300
        return out

    def discriminator_pass(self, y_disc_cache, wav_seg_disc_cache):
        """model.py:366-384 + VitsDiscriminatorLoss (losses.py:331-351); the parameter gradients accumulate in self.disc.grads()."""
        return self.disc.d_pass(wav_seg_disc_cache, y_disc_cache)
