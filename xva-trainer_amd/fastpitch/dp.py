"""Data-parallel FastPitch step: one process per GPU, RCCL over xGMI through torch.distributed.

Replaces the reference's single-process nn.DataParallel (python/fastpitch1_1/xva_train.py:48-53,465-466): replicate /
scatter / gather / reduce-to-GPU0 every step under the GIL.  Here every rank owns a full replica and a disjoint shard of
the minibatch; per optimizer step there are exactly two exchanges:
  1. a 2-float all-reduce of the masked-mean DENOMINATORS (valid mel bins, tokens: functions of the batch alone), issued before the
     forward and travelling under it, so the losses are normalised GLOBALLY — the semantics of the reference, which computes the loss
     on the gathered outputs (xva_train.py:788-790) — without an exchange between forward and backward; the numerators are reporting
     only: each rank's losses are its share of the global means, SUM-reduced on the side stream while backward runs;
  2. a SUM all-reduce of the gradients, bucketed per transformer layer in backward-completion order.  The engine records
     a HIP event per bucket while backward is still running; each bucket's all-reduce is enqueued on a side stream that
     waits on its event, so communication overlaps the rest of backward.  xGMI is point-to-point (per-link bound), so
     buckets are whole layers (~14.6 MB fp32) rather than NVSwitch-style tiny buckets.
With gradient accumulation only the last micro-batch synchronises (pass sync=False for the others).
"""
import ctypes as C
import os

import torch
import torch.distributed as dist

from .. import _lib
from . import engine as E

lib = _lib.lib
lib.xva_fp_num_buckets.restype = C.c_int32
lib.xva_fp_bucket_range.restype = C.c_int32
lib.xva_fp_bucket_range.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
lib.xva_fp_backward_ex.restype = C.c_int32
lib.xva_fp_backward_ex.argtypes = [C.POINTER(E.FpDims), C.c_void_p, C.c_void_p, C.POINTER(E.FpBatch), C.c_void_p, C.c_int64,
                                   C.POINTER(C.c_void_p), C.c_void_p]
_BUCKET_CB = C.CFUNCTYPE(None, C.c_int32, C.c_void_p)
lib.xva_fp_set_bucket_callback.restype = None
lib.xva_fp_set_bucket_callback.argtypes = [C.c_void_p, C.c_void_p]
lib.xva_event_create.restype = C.c_void_p
lib.xva_event_destroy.argtypes = [C.c_void_p]
lib.xva_stream_wait_event.restype = C.c_int32
lib.xva_stream_wait_event.argtypes = [C.c_void_p, C.c_void_p]


def bucket_ranges():
    out = []
    for i in range(lib.xva_fp_num_buckets()):
        b, e = C.c_int64(), C.c_int64()
        _lib.check(lib.xva_fp_bucket_range(i, C.byref(b), C.byref(e)), "xva_fp_bucket_range")
        out.append((b.value, e.value))
    return out


def buckets_for_stage(stage):
    """Indices of the buckets that carry gradients in `stage` (decoder buckets are empty in stage 2)."""
    tr = E.trainable_ranges(stage)
    keep = []
    for i, (b, e) in enumerate(bucket_ranges()):
        if any(b < te and tb < e for tb, te in tr):
            if stage == 2 and i < lib.xva_fp_num_buckets() // 2:
                continue
            keep.append(i)
    return keep


class GradSync:
    def __init__(self, eng, flat, grads, world, group=None):
        self.eng, self.flat, self.grads, self.world, self.group = eng, flat, grads, world, group
        self.ranges = bucket_ranges()
        n = len(self.ranges)
        self.events = (C.c_void_p * n)(*[lib.xva_event_create() for _ in range(n)])
        self.comm = torch.cuda.Stream(device=flat.device, priority=int(os.environ.get("XVA_DP_COMM_PRIO", "-1")))
        self._works = []
        self.diag = False          # True: the next fwd_loss_bwd(sync=True) stamps timing events (see diagnostics())
        self._diag = None

    def __del__(self):
        try:
            for e in self.events:
                lib.xva_event_destroy(e)
        except Exception:
            pass

    def fwd_loss_bwd(self, batch, stage, grad_scale=1.0, sync=True):
        eng = self.eng
        cur = torch.cuda.current_stream()
        den = eng.loss_denominators(batch, stage)
        self.comm.wait_stream(cur)
        with torch.cuda.stream(self.comm):
            wden = dist.all_reduce(den, group=self.group, async_op=True)      # 2 floats, under the forward pass
        eng.forward(self.flat, batch, stage)
        acc = eng.loss_partials(batch, stage)
        wden.wait()
        acc[1:4:2].copy_(den)                                                  # acc[1], acc[3]: the GLOBAL denominators
        losses = eng.loss_grads(batch, stage, grad_scale)                      # this rank's share of the global means
        self.comm.wait_stream(cur)
        with torch.cuda.stream(self.comm):
            wloss = dist.all_reduce(losses, group=self.group, async_op=True)   # reporting: SUM of the shares, under backward
        d = eng._prepare(batch.B, batch.Tt, batch.Tm, stage)
        self._works, self._cb_error = [], None
        tev = lambda: torch.cuda.Event(enable_timing=True)
        diag = {"bwd_start": tev(), "bwd_end": tev(), "joined": tev(), "start": {}, "end": {}} if (self.diag and sync) else None
        if diag is not None:
            diag["bwd_start"].record(cur)
        if sync:
            # The engine calls back right after it has recorded a bucket's event, while this thread is still inside xva_fp_backward_ex issuing the rest
            # of backward: the bucket's wait + all-reduce go onto the exchange stream at that moment.  (Enqueued after the call had returned, every wait
            # resolved when the recording lane had drained — the exchange ran after backward instead of under it: tools/dp_overlap_probe.py.)
            comm_ptr = C.c_void_p(self.comm.cuda_stream)
            live = set(buckets_for_stage(stage))

            def on_bucket(i, _user):
                try:
                    if i in live:
                        b, e = self.ranges[i]
                        _lib.check(lib.xva_stream_wait_event(comm_ptr, self.events[i]), "xva_stream_wait_event")
                        with torch.cuda.stream(self.comm):
                            if diag is not None:
                                diag["start"][i] = tev(); diag["start"][i].record(self.comm)      # the bucket's gradients are final: its exchange may start
                            self._works.append(dist.all_reduce(self.grads[b:e], group=self.group, async_op=True))
                            if diag is not None:
                                diag["end"][i] = tev(); diag["end"][i].record(self.comm)
                        live.discard(i)
                except BaseException as ex:                         # never unwind through the C frames: re-raised below
                    self._cb_error = ex
            cb = _BUCKET_CB(on_bucket)
            lib.xva_fp_set_bucket_callback(C.cast(cb, C.c_void_p), None)
        try:
            rc = lib.xva_fp_backward_ex(C.byref(d), _lib.ptr(self.flat), _lib.ptr(self.grads), C.byref(eng._abi), _lib.ptr(eng._ws),
                                        eng._ws.numel(), self.events if sync else None, _lib.stream_ptr())
        finally:
            if sync:
                lib.xva_fp_set_bucket_callback(None, None)
        _lib.check(rc, "xva_fp_backward_ex")
        if self._cb_error is not None:
            raise self._cb_error
        if sync:
            for i in sorted(live):                                  # a bucket the engine did not announce (none today): exchange it after the call
                b, e = self.ranges[i]
                _lib.check(lib.xva_stream_wait_event(comm_ptr, self.events[i]), "xva_stream_wait_event")
                with torch.cuda.stream(self.comm):
                    self._works.append(dist.all_reduce(self.grads[b:e], group=self.group, async_op=True))
            if diag is not None:
                diag["bwd_end"].record(cur)
            for w in self._works:
                w.wait()                                            # the compute stream waits for the reduced buckets
            if diag is not None:
                diag["joined"].record(cur)
                self._diag, self.diag = diag, False
        wloss.wait()
        return losses

    def diagnostics(self):
        """Timeline of the last fwd_loss_bwd that ran with `diag = True` (the first real multi-GPU run must explain itself):
        bucket_start_ms[i] — when bucket i's exchange could start, relative to the start of backward (event on the exchange stream behind the bucket's event);
        bucket_ms[i] — duration of its all-reduce; backward_ms; allreduce_ms_exposed — how long the compute stream waited for the exchange after backward's
        last kernel (0 = fully overlapped)."""
        g = self._diag
        if g is None:
            return None
        torch.cuda.synchronize()
        order = sorted(g["start"])
        return {"buckets": order, "bucket_bytes": [4 * (self.ranges[i][1] - self.ranges[i][0]) for i in order],
                "bucket_start_ms": [g["bwd_start"].elapsed_time(g["start"][i]) for i in order],
                "bucket_ms": [g["start"][i].elapsed_time(g["end"][i]) for i in order],
                "backward_ms": g["bwd_start"].elapsed_time(g["bwd_end"]), "allreduce_ms_exposed": g["bwd_end"].elapsed_time(g["joined"])}


def allreduce_flat_buckets(grads, ranges, group=None):
    """Backend-agnostic bucketed SUM all-reduce of a flat gradient buffer (used by the CPU/gloo tests and as the plain
    fallback when no event overlap is wanted)."""
    works = [dist.all_reduce(grads[b:e], group=group, async_op=True) for b, e in ranges if e > b]
    for w in works:
        w.wait()
