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

import os

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


BACKWARD_BUCKETS = ("posterior_encoder", "flow", "duration_predictor", "pitch_predictor", "text_encoder", "pitch_emb", "emb_l")   # order the backward pass finishes them
GEN_GROUP_ORDER = ("emb_l.", "text_encoder.", "duration_predictor.", "flow.", "posterior_encoder.", "waveform_decoder.", "pitch_predictor.", "pitch_emb.")


class FlatGroupAdamW:
    """torch.optim.AdamW (betas (0.8, 0.99), eps 1e-9, weight decay 0.01: python/xvapitch/training_util.py:56-57) over one optimiser group of the
    xVAPitch trainer, stepped by xva_adamw_step and (de)serialised in torch's own state_dict format over the REFERENCE's parameter order
    (make_optim chains emb_l, text_encoder, duration_predictor, flow, posterior_encoder, waveform_decoder, pitch_predictor, pitch_emb; the
    discriminator group is model.disc.parameters()), so `xVAPitch_*.pt` checkpoints carry optimiser states either trainer can resume from.
    Entries are either named tensors of the acoustic modules — moved, at the first step, with their gradients and moments into flat arenas
    (_flatten): one launch over the parameters that have a gradient; the ones the loss never reaches sit in a tail that is not stepped, like
    torch skips p.grad None — or a whole engine's flat parameter buffer (waveform decoder, discriminator: one launch each).  One step counter for the group (torch's per-parameter counters only differ for
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
        self.flat_p = self.flat_g = self.owner = None
        self.n_live, self.bucket_slices = 0, {}
        self.fm = [torch.zeros_like(e.params) for _, e, _ in self.flats]
        self.fv = [torch.zeros_like(e.params) for _, e, _ in self.flats]

    @classmethod
    def for_generator(cls, acoustic, decoder, lr):
        ref_shapes = {k: tuple(v.shape) for k, v in acoustic.state_dict().items()}
        ent = [(k, t, g, ref_shapes[k]) for k, t, g in acoustic.named_param_grads()]
        rank = lambda k: next(i for i, pre in enumerate(GEN_GROUP_ORDER) if k.startswith(pre))
        ent.sort(key=lambda e: rank(e[0]))              # stable: module order inside a group
        opt = cls(ent, [("waveform_decoder.", decoder, GEN_GROUP_ORDER.index("waveform_decoder."))], lr)
        opt.owner = acoustic
        return opt

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

    def _flatten(self):
        """First step: move the named parameters, their gradients and their moments into flat arenas — live parameters (the ones that have a
        gradient now: the set is structural) first, bucket by bucket in the backward order BucketedSync sends them, the never-reached ones in a
        tail the update does not cover — and re-point the modules' own tensors (`.data`) at the arena.  From here on the group is ONE
        xva_adamw_step launch over [0, n_live), zero_grad is one memset, and a data-parallel bucket is a contiguous slice: no gather / scatter."""
        ent = [(k, t, g, shp, g()) for k, t, g, shp in self.named]
        bucket = lambda k: BACKWARD_BUCKETS.index(k.split(".")[0]) if k.split(".")[0] in BACKWARD_BUCKETS else len(BACKWARD_BUCKETS)
        ent.sort(key=lambda e: (e[4] is None, bucket(e[0])))                     # stable inside a bucket
        al = lambda c: (c + 15) // 16 * 16                                       # every tensor on a 64-byte boundary (GEMM operands want 16; the gaps stay zero)
        n = sum(al(e[1].numel()) for e in ent)
        dev = self.m.device
        P, G = torch.empty(max(n, 1), device=dev), torch.zeros(max(n, 1), device=dev)
        m, v = torch.zeros(max(n, 1), device=dev), torch.zeros(max(n, 1), device=dev)
        old = {k: o for (k, _, _, _), o in zip(self.named, self.offs)}
        offs, off, slices = [], 0, {}
        with torch.no_grad():
            for k, t, _, _, g in ent:
                c = t.numel()
                P[off:off + c].copy_(t.detach().reshape(-1))
                m[off:off + c].copy_(self.m[old[k]:old[k] + c]); v[off:off + c].copy_(self.v[old[k]:old[k] + c])
                t.data = P[off:off + c].view(t.shape)
                if g is not None:
                    G[off:off + c].copy_(g.detach().reshape(-1))
                    g.data = G[off:off + c].view(g.shape)
                    b = k.split(".")[0]
                    slices[b] = (slices.get(b, (off, off))[0], off + c)
                    self.n_live = off + c
                offs.append(off)
                off += al(c)
        self.named = [e[:4] for e in ent]
        self.offs, self.m, self.v, self.flat_p, self.flat_g, self.bucket_slices = offs, m, v, P, G, slices
        _lib.PARAM_EPOCH[0] += 1                                                # pointer tables of the engine calls (transformer.py) are stale now
        if self.owner is not None:
            self.owner._flat_g, self.owner._flat_buckets = G, slices

    def step(self):
        self.step_count += 1
        if self.named:
            if self.flat_p is None:
                self._flatten()
            if self.n_live:
                n = self.n_live
                self._launch(self.flat_p[:n], self.flat_g[:n], self.m[:n], self.v[:n])
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
        self._early = set()

    def attach(self, out):
        """Exchange what is FINAL while the generator backward is still running (north_star: "overlapped with the backward pass").  The backward visits the
        modules in reverse creation order — waveform decoder, pitch predictor, KL / expansion, duration predictor, flow, text encoder, posterior encoder —
        so two buckets are complete long before it ends: the decoder's flat gradient (one engine call; the largest tensor of the generator group, 14 M
        parameters in the reference's size) once d(loss) / d(z_slice) exists, and the flow's slice of the gradient arena once d(loss) / d(z) does.  A
        tensor hook on those two activations enqueues the bucket's all-reduce on the exchange stream at that moment — the host is still inside
        `loss.backward()` issuing the rest (the same reasoning as the engines' bucket callbacks, fastpitch/dp.py).  start_generator() sends the rest."""
        self._early = set()
        ac, dec = self.step.gen.acoustic, self.step.gen.decoder

        def on_decoder(_grad):
            if "waveform_decoder" not in self._early:
                self._early.add("waveform_decoder")
                self._launch("gen", [dec.grad])
        out["z_slice"].register_hook(on_decoder)
        flat, slices = getattr(ac, "_flat_g", None), getattr(ac, "_flat_buckets", None)
        if flat is not None and slices and "flow" in slices and out["z"].requires_grad:
            b, e = slices["flow"]

            def on_flow(_grad):
                if "flow" not in self._early:
                    self._early.add("flow")
                    self._launch("gen", [flat[b:e]])
            out["z"].register_hook(on_flow)

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
        """the buckets attach() has not already sent (all of them when it was not called)"""
        ac, dec = self.step.gen.acoustic, self.step.gen.decoder
        early, self._early = self._early, set()
        if "waveform_decoder" not in early:
            self._launch("gen", [dec.grad])                                      # final first (the decoder is the head of the backward pass)
        flat, slices = getattr(ac, "_flat_g", None), getattr(ac, "_flat_buckets", None)
        if flat is not None:                                                     # after the first optimiser step: a bucket is a slice of the gradient arena
            for name in list(BACKWARD_BUCKETS) + [n for n in slices if n not in BACKWARD_BUCKETS]:
                if name in slices and name not in early:
                    self._launch("gen", [flat[slices[name][0]:slices[name][1]]])
            return
        buckets = {}
        for k, _, g in ac.named_param_grads():
            buckets.setdefault(k.split(".")[0], []).append(g())
        for name in BACKWARD_BUCKETS:
            if name in buckets:
                self._launch("gen", buckets.pop(name))
        for rest in buckets.values():
            self._launch("gen", rest)

    def start_discriminator(self):
        side = self.step.gen.branch_stream(self.step.disc.grad.device)     # eager_disc: the discriminator pass ran on the vocoder branch's stream, which nobody has joined yet
        if side != torch.cuda.current_stream():
            self.comm.wait_stream(side)
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
                       train=False, eager_disc=False):
        """eager_disc: also run the DISCRIMINATOR pass (model.py:366-384) here, on the vocoder branch's stream right after the adversarial terms — it needs
        nothing but the decoder's detached output and the recording's segment, and the discriminator's parameters do not change before the optimiser steps
        at the end of the iteration, so its result is the one the reference computes after the generator's backward pass.  The caller zeroes the
        discriminator's gradients BEFORE this call (the reference zeroes them at the start of pass 1) and collects the loss with discriminator_pass().
        With it the pass also returns WITHOUT joining the branch's stream (GeneratorPass late_join): the branch's tensors and the loss VALUES in `out` are
        complete after out["loss"].backward() (which returns joined) or after self.gen.join()."""
        from .wn import seq_arena_begin
        seq_arena_begin(y.device)          # a new iteration: the previous one's sequences are dead, their slab is zeroed in one memset and reused
        self._eager = None

        def adversarial(o, wav_seg):                     # on the decoder's output, inside the vocoder branch (generator_pass.py)
            loss_gen, loss_feat = _Adversarial.apply(o, wav_seg, self.disc)                                     # model.py:313-315, losses.py:195-196
            if eager_disc:
                with torch.no_grad():
                    self._eager = (o, wav_seg, self.disc.d_pass(wav_seg, o.detach()))
            return {"loss_gen": loss_gen, "loss_feat": loss_feat}                                               # losses.py:300: summed into "loss"
        out = self.gen(tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=pitch_padded, eps=eps, noise=noise, slice_ids=slice_ids,
                       tail=adversarial, late_join=bool(eager_disc) and os.environ.get("XVA_C5_LATE_JOIN", "1") != "0")
        if self._eager is not None:
            self._eager[2].record_stream(torch.cuda.current_stream(y.device))
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
        eager, self._eager = getattr(self, "_eager", None), None
        if eager is not None and eager[0].data_ptr() == y_disc_cache.data_ptr() and eager[1].data_ptr() == wav_seg_disc_cache.data_ptr():
            return eager[2]                              # generator_pass(eager_disc=True) has already run it on these two tensors
        return self.disc.d_pass(wav_seg_disc_cache, y_disc_cache)
