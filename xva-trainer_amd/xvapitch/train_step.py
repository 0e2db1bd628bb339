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
import ctypes as C

import torch

from .. import _lib

_lib.lib.xva_adamw_step.restype = C.c_int32
_lib.lib.xva_adamw_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_float, C.c_void_p]


def allreduce_mean_(tensors, group=None):
    """Data-parallel gradient reduction of a list of tensors: ONE flat SUM all-reduce (RCCL on GPU ranks, gloo in the CPU tests) and a division by
    the world size, written back in place.  The mean is nn.DataParallel's semantics in the reference trainer (python/xvapitch/xva_train.py:77-82,
    427-428 wrap the model; each replica normalises its losses over its own sub-batch and `loss_dict["loss"].mean()` averages the replicas, :664-668)."""
    import torch.distributed as dist
    ts = [t for t in tensors if t is not None]
    if not ts or not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    flat = torch.cat([t.detach().reshape(-1) for t in ts])
    dist.all_reduce(flat, group=group)
    flat.mul_(1.0 / world)
    views, off = [], 0
    for t in ts:
        views.append(flat[off:off + t.numel()].view(t.shape))
        off += t.numel()
    with torch.no_grad():
        torch._foreach_copy_([t.detach() for t in ts], views)


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

    # ---- the two torch.optim.AdamW of python/xvapitch/training_util.py:56-57 (betas 0.8 / 0.99, eps 1e-9, weight decay 0.01; lr args.lr / 2e-4) ----
    def _adamw(self, key, flat_p, flat_g, lr, betas, eps, weight_decay):
        st = self._opt.setdefault(key, {"m": torch.zeros_like(flat_p), "v": torch.zeros_like(flat_p)})
        _lib.check(_lib.lib.xva_adamw_step(_lib.ptr(flat_p), _lib.ptr(flat_g), _lib.ptr(st["m"]), _lib.ptr(st["v"]), flat_p.numel(), self._opt_step, lr,
                                           betas[0], betas[1], eps, weight_decay, _lib.stream_ptr()), "xva_adamw_step")

    def optimizer_step(self, lr=2e-4, lr_disc=2e-4, betas=(0.8, 0.99), eps=1e-9, weight_decay=0.01):
        """One step of both optimisers on the gradients the two passes left (xva_train.py:722-735: both step after the iteration's backward
        passes).  Generator group = every module make_optim chains (emb_l, text encoder, duration predictor, flow, posterior encoder, waveform
        decoder, pitch predictor, pitch_emb): one AdamW, so the acoustic parameters are stepped as one flat vector (gathered from / scattered
        back to the modules' tensors), the decoder and the discriminator in their own flat buffers — one xva_adamw_step each."""
        if not hasattr(self, "_opt"):
            self._opt, self._opt_step = {}, 0
        self._opt_step += 1
        ac, dec, D = self.gen.acoustic, self.gen.decoder, self.disc
        pairs = [(p_, g_) for p_, g_ in ac.param_grad_pairs() if g_ is not None]          # parameters no loss term reached are not stepped (torch skips p.grad is None)
        ps = [p_.detach() for p_, _ in pairs]
        key = "acoustic:%d" % len(ps)
        flat_p = torch.cat([t.reshape(-1) for t in ps])                                    # gather -> ONE xva_adamw_step -> scatter (torch copies: plumbing)
        flat_g = torch.cat([g_.detach().reshape(-1) for _, g_ in pairs])
        self._adamw(key, flat_p, flat_g, lr, betas, eps, weight_decay)
        views, off = [], 0
        for t in ps:
            views.append(flat_p[off:off + t.numel()].view(t.shape))
            off += t.numel()
        with torch.no_grad():
            torch._foreach_copy_(ps, views)
        self._adamw("decoder", dec.params, dec.grad, lr, betas, eps, weight_decay)
        self._adamw("disc", D.params, D.grad, lr_disc, betas, eps, weight_decay)

    def sync_gradients(self, group=None):
        """One process per GPU (torch.distributed over RCCL): average the iteration's gradients across the ranks — the generator group (acoustic
        modules + decoder) and the discriminator — before optimizer_step.  No-op without an initialised process group."""
        ac, dec, D = self.gen.acoustic, self.gen.decoder, self.disc
        allreduce_mean_([g_ for _, g_ in ac.param_grad_pairs() if g_ is not None] + [dec.grad], group)
        allreduce_mean_([D.grad], group)

    def discriminator_pass(self, y_disc_cache, wav_seg_disc_cache):
        """model.py:366-384 + VitsDiscriminatorLoss (losses.py:331-351); the parameter gradients accumulate in self.disc.grads()."""
        return self.disc.d_pass(wav_seg_disc_cache, y_disc_cache)
