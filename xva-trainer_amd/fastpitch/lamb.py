"""Fused multi-tensor LAMB on libxvahip — mirror of python/fastpitch1_1/lamb.py (class Lamb).

Same hyper-parameters and semantics (no bias correction, weight-norm clamp 10, trust ratio 1 if a norm is zero, tensors
without a gradient are skipped), same state_dict layout (per-parameter step / exp_avg / exp_avg_sq / weight_norm /
adam_norm / trust_ratio, indexed in the reference's parameter order) — but one step is 4 kernel launches over flat
buffers instead of ~2k ATen calls.  clip_grad_norm_ (xva_train.py:857) is fused in front of it.
"""
import ctypes as C

import torch

from .. import _lib
from . import params as P

lib = _lib.lib
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
lib.xva_opt_build_chunks.restype = i64
lib.xva_opt_build_chunks.argtypes = [C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(i64), C.POINTER(i32), i64]
lib.xva_lamb_step.restype = i32
lib.xva_lamb_step.argtypes = [vp, vp, vp, vp, i64, vp, vp, vp, i64, i32, vp, vp, f32, f32, f32, f32, f32, f32, f32, vp]


class Lamb:
    def __init__(self, flat_params, table, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0, adam=False):
        if adam:
            raise NotImplementedError("adam=True (trust ratio forced to 1) is never used by the reference trainer")
        self.flat = flat_params
        self.table = table
        self.param_groups = [dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
        dev = flat_params.device
        self.exp_avg = torch.zeros_like(flat_params)
        self.exp_avg_sq = torch.zeros_like(flat_params)
        self.steps = [0] * len(table)
        self._scal = torch.zeros(4, device=dev)
        self._norms = torch.zeros(2 * len(table), device=dev)
        self._chunks = {}

    def _chunk_tables(self, active_names):
        key = tuple(sorted(active_names))
        if key not in self._chunks:
            n = len(self.table)
            offs = (i64 * n)(*[t[1] for t in self.table])
            nums = (i64 * n)(*[t[2] for t in self.table])
            act = (i32 * n)(*[1 if t[0] in active_names else 0 for t in self.table])
            cnt = lib.xva_opt_build_chunks(offs, nums, act, n, None, None, None, 0)
            ctid, cstart, clen = (i32 * cnt)(), (i64 * cnt)(), (i32 * cnt)()
            lib.xva_opt_build_chunks(offs, nums, act, n, ctid, cstart, clen, cnt)
            dev = self.flat.device
            self._chunks[key] = (torch.tensor(list(ctid), dtype=torch.int32, device=dev),
                                 torch.tensor(list(cstart), dtype=torch.int64, device=dev),
                                 torch.tensor(list(clen), dtype=torch.int32, device=dev), cnt,
                                 [i for i, t in enumerate(self.table) if t[0] in active_names])
        return self._chunks[key]

    def step(self, flat_grads, active_names, max_grad_norm=1000.0, inv_scale=1.0):
        """clip_grad_norm_(max_grad_norm) + LAMB over the tensors in `active_names` (those that received a gradient)."""
        g = self.param_groups[0]
        ctid, cstart, clen, cnt, idxs = self._chunk_tables(active_names)
        rc = lib.xva_lamb_step(_lib.ptr(self.flat), _lib.ptr(flat_grads), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                               self.flat.numel(), _lib.ptr(ctid), _lib.ptr(cstart), _lib.ptr(clen), cnt, len(self.table),
                               _lib.ptr(self._scal), _lib.ptr(self._norms), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                               g["weight_decay"], max_grad_norm, inv_scale, _lib.stream_ptr())
        _lib.check(rc, "xva_lamb_step")
        for i in idxs:
            self.steps[i] += 1

    @property
    def skipped(self):
        """1.0 if the last step met a non-finite gradient and was skipped on the device (moments and parameters untouched), else 0.0 (device scalar)."""
        return self._scal[3]

    @property
    def grad_norm(self):
        """Pre-clip global gradient norm of the last step (device scalar)."""
        return self._scal[2]

    # ---- checkpoint format of torch.optim.Optimizer.state_dict() over the reference's parameter order ----
    def state_dict(self):
        order = P.reference_param_order(self.table)
        by_name = {t[0]: (k, t) for k, t in enumerate(self.table)}
        m = P.from_flat(self.exp_avg, self.table)
        v = P.from_flat(self.exp_avg_sq, self.table)
        norms = self._norms.detach().cpu()
        state = {}
        for idx, name in enumerate(order):
            k, _ = by_name[name]
            if self.steps[k] == 0:
                continue
            wn = norms[2 * k].sqrt().clamp(0, 10)
            an = norms[2 * k + 1].sqrt()
            state[idx] = {"step": self.steps[k], "exp_avg": m[name], "exp_avg_sq": v[name], "weight_norm": wn, "adam_norm": an,
                          "trust_ratio": 1 if (wn == 0 or an == 0) else wn / an}
        g = dict(self.param_groups[0])
        g["params"] = list(range(len(order)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        order = P.reference_param_order(self.table)
        by_name = {t[0]: (k, t) for k, t in enumerate(self.table)}
        m, v = {}, {}
        for idx, st in sd["state"].items():
            name = order[int(idx)]
            k, _ = by_name[name]
            self.steps[k] = int(st.get("step", 0))
            m[name], v[name] = st["exp_avg"], st["exp_avg_sq"]
        sub = [t for t in self.table if t[0] in m]
        P.to_flat(m, sub, self.exp_avg)
        P.to_flat(v, sub, self.exp_avg_sq)
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]
