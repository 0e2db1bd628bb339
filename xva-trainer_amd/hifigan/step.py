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
    """torch.optim.AdamW over the trainable prefix of a flat buffer (python/hifigan/xva_train.py:298-300)."""

    def __init__(self, flat, n_trainable, lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01):
        self.flat, self.n = flat, int(n_trainable)
        self.param_groups = [dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)]
        self.exp_avg = torch.zeros(self.n, device=flat.device)
        self.exp_avg_sq = torch.zeros(self.n, device=flat.device)
        self.step_count = 0

    def step(self, grads):
        g = self.param_groups[0]
        self.step_count += 1
        _lib.check(lib.xva_adamw_step(_lib.ptr(self.flat), _lib.ptr(grads), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq), self.n,
                                      self.step_count, g["lr"], g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], _lib.stream_ptr()),
                   "xva_adamw_step")


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

    def reduce(self):
        """Call right after the *_ex backward was enqueued with self.events: returns once the compute stream is ordered after
        every bucket's all-reduce (no host sync)."""
        dist = torch.distributed
        comm_ptr = C.c_void_p(self.comm.cuda_stream)
        works = []
        for i, (b, e) in enumerate(self.ranges):
            _lib.check(lib.xva_stream_wait_event(comm_ptr, self.events[i]), "xva_stream_wait_event")
            with torch.cuda.stream(self.comm):
                works.append(dist.all_reduce(self.grads[b:e], op=dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM, group=self.group,
                                             async_op=True))
        for w in works:
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
        self.optim_g = FlatAdamW(self.flat_g, self.eng.trainable[E.G], lr, betas)
        self.optim_d = FlatAdamW(self.flat_d, self.eng.trainable[E.D], lr, betas)
        self.group = group
        self.world = torch.distributed.get_world_size(group) if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
        self.sync_d = BucketSync(E.D, self.grads_d, group) if self.world > 1 else None
        self.sync_g = BucketSync(E.G, self.grads_g, group) if self.world > 1 else None

    # ---- checkpoint tensors (python/hifigan/xva_train.py:570-601: {'generator': sd}, {'mpd': sd, 'msd': sd, ...}) ----
    def load_state_dicts(self, generator=None, mpd=None, msd=None):
        if generator is not None:
            E.to_flat(generator, self.eng.table[E.G], self.flat_g)
        if mpd is not None:
            E.to_flat(mpd, self.eng.table[E.D], self.flat_d, "mpd.")
        if msd is not None:
            E.to_flat(msd, self.eng.table[E.D], self.flat_d, "msd.")

    def state_dicts(self):
        return {"generator": E.from_flat(self.flat_g, self.eng.table[E.G]), "mpd": E.from_flat(self.flat_d, self.eng.table[E.D], "mpd."),
                "msd": E.from_flat(self.flat_d, self.eng.table[E.D], "msd.")}

    def train_step(self, x_mel, y_wav, y_mel):
        """x_mel (B, 80, T) input mel (fmax 8000), y_wav (B, T*256) target audio, y_mel (B, 80, T) loss mel (fmax None).
        Returns device tensors: dict(loss_disc_all, loss_gen_all, loss_mel, loss_fm, loss_gen, y_g_hat)."""
        eng = self.eng
        y_g_hat = eng.generator_forward(self.flat_g, x_mel)
        # ---- discriminator step
        ld = eng.disc_forward(self.flat_d, y_wav, y_g_hat)
        self.grads_d.zero_()
        eng.disc_backward_d(self.flat_d, self.grads_d, self.sync_d.events if self.sync_d else None)
        if self.sync_d:
            self.sync_d.reduce()
        self.optim_d.step(self.grads_d)
        # ---- generator step (updated discriminators)
        lg = eng.disc_forward(self.flat_d, y_wav, y_g_hat)
        d_wav = eng.disc_backward_g(self.flat_d)
        loss_mel, _ = pmel.mel_l1_loss_backward(y_g_hat, y_mel, d_wav, scale=45.0, accumulate=True)
        self.grads_g.zero_()
        eng.generator_backward(self.flat_g, self.grads_g, d_wav, self.sync_g.events if self.sync_g else None)
        if self.sync_g:
            self.sync_g.reduce()
        self.optim_g.step(self.grads_g)
        return {"loss_disc_all": ld[0], "loss_gen": lg[1], "loss_fm": lg[2], "loss_mel": loss_mel[0],
                "loss_gen_all": lg[1] + lg[2] + loss_mel[0], "y_g_hat": y_g_hat}
