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


GEN_GROUP_ORDER = ("emb_l.", "text_encoder.", "duration_predictor.", "flow.", "posterior_encoder.", "waveform_decoder.", "pitch_predictor.", "pitch_emb.")


class FlatGroupAdamW:
    """torch.optim.AdamW (betas (0.8, 0.99), eps 1e-9, weight decay 0.01: python/xvapitch/training_util.py:56-57) over one optimiser group of the
    xVAPitch trainer, stepped by xva_adamw_step and (de)serialised in torch's own state_dict format over the REFERENCE's parameter order
    (make_optim chains emb_l, text_encoder, duration_predictor, flow, posterior_encoder, waveform_decoder, pitch_predictor, pitch_emb; the
    discriminator group is model.disc.parameters()), so `xVAPitch_*.pt` checkpoints carry optimiser states either trainer can resume from.
    Entries are either named tensors of the acoustic modules (moments in ONE flat buffer with fixed offsets — a parameter keeps its moments
    whatever the other parameters' gradients do; parameters without a gradient are not stepped, like torch) or a whole engine's flat parameter
    buffer (waveform decoder, discriminator: one launch).  One step counter for the group (torch's per-parameter counters only differ for
    parameters that are never reached, and those have no state)."""

    def __init__(self, named, flats, lr, betas=(0.8, 0.99), eps=1e-9, weight_decay=0.01):
        self.named = list(named)                       # [(key, tensor, grad getter, checkpoint shape)]
        self.flats = list(flats)                       # [(key prefix, engine with .params / .grad / .table, position in the parameter order)]
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, initial_lr=lr)]
        self.step_count = 0
        dev = (self.named[0][1] if self.named else self.flats[0][1].params).device
        self.offs, n = [], 0
        for _, t, _, _ in self.named:
            self.offs.append(n)
            n += t.numel()
        self.m = torch.zeros(max(n, 1), device=dev)
        self.v = torch.zeros(max(n, 1), device=dev)
        self.fm = [torch.zeros_like(e.params) for _, e, _ in self.flats]
        self.fv = [torch.zeros_like(e.params) for _, e, _ in self.flats]

    @classmethod
    def for_generator(cls, acoustic, decoder, lr):
        ref_shapes = {k: tuple(v.shape) for k, v in acoustic.state_dict().items()}
        ent = [(k, t, g, ref_shapes[k]) for k, t, g in acoustic.named_param_grads()]
        rank = lambda k: next(i for i, pre in enumerate(GEN_GROUP_ORDER) if k.startswith(pre))
        ent.sort(key=lambda e: rank(e[0]))              # stable: module order inside a group
        return cls(ent, [("waveform_decoder.", decoder, GEN_GROUP_ORDER.index("waveform_decoder."))], lr)

    @classmethod
    def for_discriminator(cls, disc, lr):
        return cls([], [("disc.", disc, 0)], lr)

    def order(self):
        """[(key, checkpoint shape)] in the reference optimiser's parameter order: make_optim's chain of modules, inside a module torch's
        parameters() walk (param_order.order_key: the registration order of siblings recorded from the reference's classes)."""
        from .param_order import order_key
        ent = [(k, shp) for k, _, _, shp in self.named]
        for pre, e, _ in self.flats:
            ent += [(pre + n, tuple(shape)) for n, (off, numel, shape) in e.table.items()]
        grp = lambda k: next((i for i, pre in enumerate(GEN_GROUP_ORDER) if k.startswith(pre)), len(GEN_GROUP_ORDER))
        return sorted(ent, key=lambda e: (grp(e[0]), order_key(e[0])))

    def _launch(self, p, g, m, v):
        h = self.param_groups[0]
        _lib.check(_lib.lib.xva_adamw_step(_lib.ptr(p), _lib.ptr(g), _lib.ptr(m), _lib.ptr(v), p.numel(), self.step_count, h["lr"], h["betas"][0],
                                           h["betas"][1], h["eps"], h["weight_decay"], _lib.stream_ptr()), "xva_adamw_step")

    def step(self):
        self.step_count += 1
        live = [(i, t, g()) for i, (_, t, g, _) in enumerate(self.named)]
        live = [(i, t, g) for i, t, g in live if g is not None]
        if live:
            ps = [t.detach() for _, t, _ in live]
            sl = [slice(self.offs[i], self.offs[i] + t.numel()) for i, t, _ in live]
            flat_p = torch.cat([t.reshape(-1) for t in ps])                      # gather -> ONE xva_adamw_step -> scatter (torch copies: plumbing)
            flat_g = torch.cat([g.detach().reshape(-1) for _, _, g in live])
            flat_m = torch.cat([self.m[s_] for s_ in sl])
            flat_v = torch.cat([self.v[s_] for s_ in sl])
            self._launch(flat_p, flat_g, flat_m, flat_v)
            off = 0
            views = []
            for (i, t, _), s_ in zip(live, sl):
                n = t.numel()
                views.append(flat_p[off:off + n].view(t.shape))
                self.m[s_] = flat_m[off:off + n]
                self.v[s_] = flat_v[off:off + n]
                off += n
            with torch.no_grad():
                torch._foreach_copy_(ps, views)
        for (pre, e, _), m, v in zip(self.flats, self.fm, self.fv):
            self._launch(e.params, e.grad, m, v)

    # ---- torch.optim.AdamW's state_dict format over the reference parameter order ----
    def _state_views(self):
        """key -> (exp_avg view, exp_avg_sq view) in the tensors' own (possibly zero-padded) shapes"""
        out = {}
        for (k, t, _, _), o in zip(self.named, self.offs):
            out[k] = (self.m[o:o + t.numel()].view(t.shape), self.v[o:o + t.numel()].view(t.shape))
        for (pre, e, _), m, v in zip(self.flats, self.fm, self.fv):
            for n, (off, numel, shape) in e.table.items():
                out[pre + n] = (m[off:off + numel].view(shape), v[off:off + numel].view(shape))
        return out

    def state_dict(self):
        h = self.param_groups[0]
        order = self.order()
        group = {"lr": h["lr"], "betas": tuple(h["betas"]), "eps": h["eps"], "weight_decay": h["weight_decay"], "amsgrad": False, "foreach": None,
                 "maximize": False, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True,
                 "initial_lr": h["initial_lr"], "params": list(range(len(order)))}
        state = {}
        if self.step_count > 0:
            views = self._state_views()
            for i, (k, shp) in enumerate(order):
                m, v = views[k]
                idx = tuple(slice(0, d) for d in shp)                            # zero-padded tensors are stored at their checkpoint shape
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": m[idx].detach().cpu().clone(), "exp_avg_sq": v[idx].detach().cpu().clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        order = self.order()
        g = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or len(g["params"]) != len(order):
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
        views = self._state_views()
        for idx, st in sd["state"].items():
            k, shp = order[int(idx)]
            if tuple(st["exp_avg"].shape) != tuple(shp):
                raise ValueError("optimizer state %s (%s): shape %s != %s" % (idx, k, tuple(st["exp_avg"].shape), tuple(shp)))
        self.m.zero_(); self.v.zero_()
        for m, v in zip(self.fm, self.fv):
            m.zero_(); v.zero_()
        self.step_count = 0
        for idx, st in sd["state"].items():
            k, shp = order[int(idx)]
            m, v = views[k]
            sl = tuple(slice(0, d) for d in shp)
            m[sl].copy_(st["exp_avg"].to(m)); v[sl].copy_(st["exp_avg_sq"].to(v))
            self.step_count = max(self.step_count, int(float(st["step"])))
        self.param_groups[0].update(lr=float(g["lr"]), betas=tuple(g["betas"]), eps=float(g["eps"]), weight_decay=float(g["weight_decay"]),
                                    initial_lr=float(g.get("initial_lr", g["lr"])))


class BucketedSync:
    """Data-parallel mean of one iteration's gradients, bucketed and overlapped with the rest of the iteration (one process per GPU, RCCL over
    xGMI; the reference wraps the model in nn.DataParallel, python/xvapitch/xva_train.py:77-82,427-428).  The generator group goes out in
    buckets on a side stream as soon as the generator backward has been enqueued — the waveform decoder's flat gradient and the acoustic
    modules' gradients group by group (posterior encoder, flow, text encoder, ...: ~15 - 60 MB each, sized for per-link-bound xGMI) — so the
    exchange runs UNDER the discriminator pass; the discriminator's flat gradient goes out after its pass and runs under the generator
    optimiser's update.  finish_*() make the compute stream wait for a group's buckets right before that group's optimiser step."""

    def __init__(self, step, group=None):
        self.step, self.group = step, group
        self.comm = torch.cuda.Stream()
        self.pending = {"gen": [], "disc": []}

    def _launch(self, which, tensors):
        import torch.distributed as dist
        ts = [t for t in tensors if t is not None]
        if not ts:
            return
        world = dist.get_world_size(self.group)
        self.comm.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.comm):
            flat = torch.cat([t.detach().reshape(-1) for t in ts]) if len(ts) > 1 else ts[0].detach().reshape(-1)
            dist.all_reduce(flat, group=self.group)
            flat.mul_(1.0 / world)
            if len(ts) > 1:
                off, views = 0, []
                for t in ts:
                    views.append(flat[off:off + t.numel()].view(t.shape))
                    off += t.numel()
                with torch.no_grad():
                    torch._foreach_copy_([t.detach() for t in ts], views)
            ev = torch.cuda.Event()
            ev.record(self.comm)
        for t in ts:
            t.record_stream(self.comm)
        self.pending[which].append(ev)

    def start_generator(self):
        ac, dec = self.step.gen.acoustic, self.step.gen.decoder
        self._launch("gen", [dec.grad])                                          # final first (the decoder is the head of the backward pass)
        buckets = {}
        for k, _, g in ac.named_param_grads():
            buckets.setdefault(k.split(".")[0], []).append(g())
        for name in ("posterior_encoder", "flow", "duration_predictor", "pitch_predictor", "text_encoder", "pitch_emb", "emb_l"):   # backward order
            if name in buckets:
                self._launch("gen", buckets.pop(name))
        for rest in buckets.values():
            self._launch("gen", rest)

    def start_discriminator(self):
        self._launch("disc", [self.step.disc.grad])

    def finish(self, which):
        for ev in self.pending[which]:
            torch.cuda.current_stream().wait_event(ev)
        self.pending[which] = []

    def reduce(self):
        """blocking form: everything, then wait (tests; a trainer without overlap)"""
        self.start_generator(); self.start_discriminator(); self.finish("gen"); self.finish("disc")


class XVAPitchStep:
    def __init__(self, generator_pass, discriminator):
        self.gen, self.disc = generator_pass, discriminator

    def generator_pass(self, tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=None, eps=None, noise=None, slice_ids=None,
                       train=False):
        out = self.gen(tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=pitch_padded, eps=eps, noise=noise, slice_ids=slice_ids)
        loss_gen, loss_feat = _Adversarial.apply(out["model_outputs"], out["waveform_seg"], self.disc)          # model.py:313-315, losses.py:195-196
        out.update({"loss_gen": loss_gen, "loss_feat": loss_feat, "loss": out["loss"] + loss_gen + loss_feat})  # losses.py:300
        return out

    # ---- the two torch.optim.AdamW of python/xvapitch/training_util.py:56-57 (betas 0.8 / 0.99, eps 1e-9, weight decay 0.01; lr args.lr / 2e-4) ----
    def optimizer_step(self, lr=2e-4, lr_disc=2e-4, betas=(0.8, 0.99), eps=1e-9, weight_decay=0.01):
        """One step of both optimisers on the gradients the two passes left (xva_train.py:722-735: both step after the iteration's backward
        passes).  Generator group = every module make_optim chains (emb_l, text encoder, duration predictor, flow, posterior encoder, waveform
        decoder, pitch predictor, pitch_emb): one AdamW — FlatGroupAdamW keeps one moment slot per parameter at a fixed offset, steps the
        parameters that have a gradient as one flat vector and the decoder / the discriminator in their own flat buffers."""
        if not hasattr(self, "_optims"):
            self._optims = [FlatGroupAdamW.for_generator(self.gen.acoustic, self.gen.decoder, lr), FlatGroupAdamW.for_discriminator(self.disc, lr_disc)]
        for opt, l in zip(self._optims, (lr, lr_disc)):
            opt.param_groups[0].update(lr=l, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
            opt.step()

    def sync_gradients(self, group=None):
        """One process per GPU (torch.distributed over RCCL): average the iteration's gradients across the ranks — the generator group (acoustic
        modules + decoder) and the discriminator — before optimizer_step.  No-op without an initialised process group."""
        ac, dec, D = self.gen.acoustic, self.gen.decoder, self.disc
        allreduce_mean_([g_ for _, g_ in ac.param_grad_pairs() if g_ is not None] + [dec.grad], group)
        allreduce_mean_([D.grad], group)

    def discriminator_pass(self, y_disc_cache, wav_seg_disc_cache):
        """model.py:366-384 + VitsDiscriminatorLoss (losses.py:331-351); the parameter gradients accumulate in self.disc.grads()."""
        return self.disc.d_pass(wav_seg_disc_cache, y_disc_cache)
