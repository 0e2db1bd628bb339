"""One HiFi-GAN optimisation iteration on libxvahip — the body of HiFiTrainer.iteration (python/hifigan/xva_train.py:479-515):

    y_g_hat = generator(x)                                           G forward
    D step: mpd / msd on (y, y_g_hat.detach()) -> discriminator_loss -> backward -> optim_d.step()
    G step: 45 * L1(mel) + mpd / msd again (updated D) -> feature_loss * 2 + generator_loss -> backward -> optim_g.step()

AdamW (lr 2e-4, betas (0.8, 0.99), torch defaults eps 1e-8 / weight_decay 0.01) runs as one fused kernel per optimizer over
the flat buffers.  The parameter gradients of the discriminators during the G step are not computed: the reference computes
and then discards them (zero_grad at the next iteration), so skipping them changes no result.
With torch.distributed initialised, gradients are averaged over ranks (equal per-rank batches = the global-batch mean):
each gradient buffer is exchanged in the engine's buckets (the 8 discriminators; the generator's stages), every bucket's
all-reduce enqueued on a side stream behind the HIP event the backward records when that bucket is final, so the exchange
runs under the rest of the backward pass (RCCL over xGMI; same scheme as fastpitch/dp.py:GradSync)."""
import ctypes as C

import torch

from .. import _lib, mel as pmel
from . import engine as E

lib = _lib.lib
lib.xva_adamw_step.restype = C.c_int32
lib.xva_adamw_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, C.c_void_p]


class FlatAdamW:
    """torch.optim.AdamW over the trainable prefix of a flat buffer (python/hifigan/xva_train.py:298-300).
    `order` = [(name, offset, numel, shape)] in the REFERENCE optimizer's parameter order (generator.parameters(); for the
    discriminators itertools.chain(msd.parameters(), mpd.parameters())): state_dict() / load_state_dict() speak torch's own
    format over it, so `do_########` checkpoints are interchangeable with the reference's optim.load_state_dict (:583,303-304)."""

    def __init__(self, flat, n_trainable, lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01, order=None):
        self.flat, self.n = flat, int(n_trainable)
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, initial_lr=lr)]
        self.exp_avg = torch.zeros(self.n, device=flat.device)
        self.exp_avg_sq = torch.zeros(self.n, device=flat.device)
        self.step_count = 0
        self.order = order

    def step(self, grads):
        g = self.param_groups[0]
        self.step_count += 1
        _lib.check(lib.xva_adamw_step(_lib.ptr(self.flat), _lib.ptr(grads), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self.n,
                                      self.step_count, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], _lib.stream_ptr()),
                   "xva_adamw_step")

    def state_dict(self):
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False, "foreach": None,
                 "maximize": False, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": True,
                 "initial_lr": g["initial_lr"],
                 "params": list(range(len(self.order)))}
        state = {}
        if self.step_count > 0:
            for i, (name, off, n, shape) in enumerate(self.order):
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[off:off + n].view(shape).cpu().clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shape).cpu().clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.order):
            raise ValueError("loaded state dict contains a parameter group that doesn't match the size of optimizer's group")
        g = groups[0]
        self.param_groups[0].update(lr=float(g["lr"]), betas=tuple(g["betas"]), eps=float(g["eps"]), weight_decay=float(g["weight_decay"]),
                                    initial_lr=float(g.get("initial_lr", g["lr"])))
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        self.step_count = 0
        for idx, st in sd["state"].items():
            name, off, n, shape = self.order[g["params"].index(idx) if idx in g["params"] else int(idx)]
            if tuple(st["exp_avg"].shape) != tuple(shape):
                raise ValueError("optimizer state %d (%s): shape %s != %s" % (idx, name, tuple(st["exp_avg"].shape), tuple(shape)))
            self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1).to(self.exp_avg))
            self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(self.exp_avg_sq))
            self.step_count = max(self.step_count, int(float(st["step"])))


def optimizer_orders(table=None):
    """[(name, offset, numel, shape)] of the trainable tensors in the reference optimizers' parameter order:
    optim_g = AdamW(generator.parameters()), optim_d = AdamW(chain(msd.parameters(), mpd.parameters())) (python/hifigan/xva_train.py:298-300)."""
    table = table or {E.G: E.tensor_table(E.G), E.D: E.tensor_table(E.D)}
    tg = [(n_, o, c, sh) for n_, o, c, sh, k in table[E.G] if k == 0]
    td = [(n_, o, c, sh) for n_, o, c, sh, k in table[E.D] if k == 0]
    return tg, [t for t in td if t[0].startswith("msd.")] + [t for t in td if t[0].startswith("mpd.")]


_BUCKET_CB = C.CFUNCTYPE(None, C.c_int32, C.c_void_p)


class BucketSync:
    """Bucketed mean all-reduce of one flat gradient buffer, overlapped with the backward that produces it."""

    def __init__(self, which, grads, group=None):
        self.grads, self.group = grads, group
        self.ranges = E.bucket_ranges(which)
        n = len(self.ranges)
        lib.xva_event_create.restype = C.c_void_p
        lib.xva_event_destroy.argtypes = [C.c_void_p]
        lib.xva_stream_wait_event.restype = C.c_int32
        lib.xva_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]
        lib.xva_hg_set_bucket_callback.restype = None
        lib.xva_hg_set_bucket_callback.argtypes = [C.c_void_p, C.c_void_p]
        self.events = (C.c_void_p * n)(*[lib.xva_event_create() for _ in range(n)])
        self.comm = torch.cuda.Stream(device=grads.device)
        self.world = torch.distributed.get_world_size(group)
        self.avg = torch.distributed.get_backend(group) == "nccl"          # RCCL has ncclAvg; gloo needs sum + scale

    def __del__(self):
        try:
            for e in self.events:
                lib.xva_event_destroy(e)
        except Exception:
            pass

    def begin(self):
        """Call right BEFORE the *_ex backward that records self.events: registers the engine's bucket callback, so that each bucket's wait + all-reduce
        are enqueued on the exchange stream the moment its event has been recorded — while the host is still issuing the rest of the pass.  (Enqueued
        after the call had returned, the waits resolved only when the recording lane had drained: the exchange ran after backward, not under it.)"""
        dist = torch.distributed
        comm_ptr = C.c_void_p(self.comm.cuda_stream)
        self._works, self._left, self._err = [], set(range(len(self.ranges))), None

        def on_bucket(i, _user):
            try:
                if i in self._left:
                    self._left.discard(i)
                    self._enqueue(i, comm_ptr, dist)
            except BaseException as ex:                             # never unwind through the C frames: re-raised in reduce()
                self._err = ex
        self._cb = _BUCKET_CB(on_bucket)
        lib.xva_hg_set_bucket_callback(C.cast(self._cb, C.c_void_p), None)

    def _enqueue(self, i, comm_ptr, dist):
        b, e = self.ranges[i]
        _lib.check(lib.xva_stream_wait_event(comm_ptr, self.events[i]), "xva_stream_wait_event")
        with torch.cuda.stream(self.comm):
            self._works.append(dist.all_reduce(self.grads[b:e], op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, group=self.group,
                                               async_op=True))

    def reduce(self):
        """Call right after that backward returned: returns once the compute stream is ordered after every bucket's all-reduce (no host sync)."""
        dist = torch.distributed
        started = getattr(self, "_cb", None) is not None
        if started:
            lib.xva_hg_set_bucket_callback(None, None)
            self._cb = None
            if self._err is not None:
                raise self._err
        else:
            self._works, self._left = [], set(range(len(self.ranges)))
        comm_ptr = C.c_void_p(self.comm.cuda_stream)
        for i in sorted(self._left):                                # buckets the engine did not announce (or begin() was not called)
            self._enqueue(i, comm_ptr, dist)
        self._left = set()
        for w in self._works:
            w.wait()
        if not self.avg:
            for b, e in self.ranges:
                self.grads[b:e].mul_(1.0 / self.world)


class HifiganStep:
    def __init__(self, device, compute="bf16", lr=2e-4, betas=(0.8, 0.99), group=None):
        self.eng = E.HifiganEngine(device, compute)
        dev = self.eng.device
        self.flat_g = torch.zeros(self.eng.total[E.G], device=dev)
        self.flat_d = torch.zeros(self.eng.total[E.D], device=dev)
        self.grads_g = torch.zeros_like(self.flat_g)
        self.grads_d = torch.zeros_like(self.flat_d)
        order_g, order_d = optimizer_orders(self.eng.table)
        self.optim_g = FlatAdamW(self.flat_g, self.eng.trainable[E.G], lr, betas, order=order_g)
        self.optim_d = FlatAdamW(self.flat_d, self.eng.trainable[E.D], lr, betas, order=order_d)
        self.group = group
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        self.sync_d = BucketSync(E.D, self.grads_d, group) if self.world > 1 else None
        self.sync_g = BucketSync(E.G, self.grads_g, group) if self.world > 1 else None

    # ---- checkpoint tensors (python/hifigan/xva_train.py:570-601: {'generator': sd}, {'mpd': sd, 'msd': sd, ...}) ----
    def load_state_dicts(self, generator=None, mpd=None, msd=None):
        self._loads = getattr(self, "_loads", 0) + 1
        if generator is not None:
            E.to_flat(generator, self.eng.table[E.G], self.flat_g)
        if mpd is not None:
            E.to_flat(mpd, self.eng.table[E.D], self.flat_d, "mpd.")
        if msd is not None:
            E.to_flat(msd, self.eng.table[E.D], self.flat_d, "msd.")

    def state_dicts(self):
        return {"generator": E.from_flat(self.flat_g, self.eng.table[E.G]), "mpd": E.from_flat(self.flat_d, self.eng.table[E.D], "mpd."),
                "msd": E.from_flat(self.flat_d, self.eng.table[E.D], "msd.")}

    def _d_token(self):
        return (self.optim_d.step_count, getattr(self, "_loads", 0))

    def train_step(self, x_mel, y_wav, y_mel):
        """x_mel (B, 80, T) input mel (fmax 8000), y_wav (B, T*256) target audio, y_mel (B, 80, T) loss mel (fmax None).
        Returns device tensors: dict(loss_disc_all, loss_gen_all, loss_mel, loss_fm, loss_gen, y_g_hat)."""
        eng = self.eng
        y_g_hat = eng.generator_forward(self.flat_g, x_mel)
        # ---- discriminator step
        # (the D step's forward runs on the parameters the previous iteration's G-step forward prepared: the token tells the engine nothing wrote them since)
        ld = eng.disc_forward(self.flat_d, y_wav, y_g_hat, losses="d", weights_token=self._d_token())
        self.grads_d.zero_()       # (zeroing both gradient buffers on a side stream under the generator forward was measured: +0.5 ... +1.1 ms — one more stream
                                   # moves the engine's lanes onto other hardware queues; tools/hg_step_time.py)
        if self.sync_d:
            self.sync_d.begin()
        eng.disc_backward_d(self.flat_d, self.grads_d, self.sync_d.events if self.sync_d else None)
        if self.sync_d:
            self.sync_d.reduce()
        self.optim_d.step(self.grads_d)
        # ---- generator step (updated discriminators)
        lg = eng.disc_forward(self.flat_d, y_wav, y_g_hat, losses="g", weights_token=self._d_token())
        d_wav = eng.disc_backward_g(self.flat_d)
        loss_mel, _ = pmel.mel_l1_loss_backward(y_g_hat, y_mel, d_wav, scale=45.0, accumulate=True)
        self.grads_g.zero_()
        if self.sync_g:
            self.sync_g.begin()
        eng.generator_backward(self.flat_g, self.grads_g, d_wav, self.sync_g.events if self.sync_g else None)
        if self.sync_g:
            self.sync_g.reduce()
        self.optim_g.step(self.grads_g)
        return {"loss_disc_all": ld[0], "loss_gen": lg[1], "loss_fm": lg[2], "loss_mel": loss_mel[0],
                "loss_gen_all": lg[1] + lg[2] + loss_mel[0], "y_g_hat": y_g_hat}
