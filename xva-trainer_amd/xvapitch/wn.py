"""WN (WaveNet gated stack) and ResidualCouplingBlock of xVAPitch on libxvahip — python/xvapitch/wavenet.py:15-109 and
python/xvapitch/model.py:1476-1535 (mean_only=True, the configuration xVAPitch builds: model.py:1400-1411).

Same constructor arguments, same state_dict keys and layouts as the reference modules (weight-normed convs: `*.weight_g`, `*.weight_v`,
`*.bias`), same (B, C, T) tensors at the interface.  Inside, activations are time-major sequences (B, pad + T + pad, C) with structurally
zero pad rows; every convolution (dilated k-tap in_layers, 1x1 res_skip / cond / pre / post) is an xva_gemm call in implicit-conv form
(forward NT, backward-data NN, backward-weight TN, LeakyReLU-free), the gate / residual-skip split / masks / coupling are the kernels
of csrc/xvapitch_ops.hip, weight norm is xva_hg_weight_norm_fwd / _bwd.  Host code only sequences C calls (no numerics in torch); both
classes are exposed to autograd through one Function each, so they compose with the rest of a model and are checked against the
reference by gradient (tests/test_xvapitch_gpu.py).  dropout_p > 0 is not built (the reference constructs these with dropout 0).
"""
import ctypes as C

import torch

from .. import _lib
from . import ops

lib = _lib.lib
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
lib.xva_wn_gate_fwd.restype = i32
lib.xva_wn_gate_fwd.argtypes = [vp, vp, i64, vp, i32, i32, i32, i32, vp]
lib.xva_wn_gate_bwd.restype = i32
lib.xva_wn_gate_bwd.argtypes = [vp, vp, i64, vp, vp, i32, i32, i32, i32, vp]
lib.xva_wn_res_skip_fwd.restype = i32
lib.xva_wn_res_skip_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
lib.xva_wn_res_skip_bwd.restype = i32
lib.xva_wn_res_skip_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]
lib.xva_seq_mask.restype = i32
lib.xva_seq_mask.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
lib.xva_seq_item_colsum.restype = i32
lib.xva_seq_item_colsum.argtypes = [vp, i32, vp, i32, i32, i32, i64, vp]
lib.xva_coupling_mean_only.restype = i32
lib.xva_coupling_mean_only.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]
lib.xva_coupling_mean_only_bwd.restype = i32
lib.xva_coupling_mean_only_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]
lib.xva_posterior_sample.restype = i32
lib.xva_posterior_sample.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
lib.xva_posterior_sample_bwd.restype = i32
lib.xva_posterior_sample_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
lib.xva_hg_weight_norm_fwd.restype = i32
lib.xva_hg_weight_norm_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]
lib.xva_hg_weight_norm_bwd.restype = i32
lib.xva_hg_weight_norm_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]


class _WnDesc(C.Structure):
    """csrc/hg_wn.h xva_wn_desc: one weight-normed tensor of a batched xva_hg_weight_norm_batch launch"""
    _fields_ = [("v", vp), ("g", vp), ("norm", vp), ("eff", vp), ("effB", vp), ("dW", vp), ("dv", vp), ("dg", vp),
                ("dt", i32), ("kind", i32), ("D0", i32), ("D1", i32), ("k", i32), ("s", i32), ("pconv", i32), ("block0", i32)]


lib.xva_hg_weight_norm_batch.restype = i32
lib.xva_hg_weight_norm_batch.argtypes = [C.POINTER(_WnDesc), i32, i32, vp]
lib.xva_hg_colsum.restype = i32
lib.xva_hg_colsum.argtypes = [vp, i32, vp, i64, i32, f32, vp]

PAD, GUARD = 8, 32


class _WnDims(C.Structure):
    """include/xva_hip.h xva_xvp_wn_dims"""
    _fields_ = [("B", i32), ("T", i32), ("H", i32), ("k", i32), ("rate", i32), ("L", i32), ("dt", i32), ("compute", i32)]


lib.xva_xvp_wn_workspace_bytes.restype = i64
lib.xva_xvp_wn_workspace_bytes.argtypes = [C.POINTER(_WnDims)]
lib.xva_xvp_wn_forward.restype = i32
lib.xva_xvp_wn_forward.argtypes = [C.POINTER(_WnDims), vp, vp, vp, vp, vp, vp, i64, vp]
lib.xva_xvp_wn_backward.restype = i32
lib.xva_xvp_wn_backward.argtypes = [C.POINTER(_WnDims), vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]
_WN_ENGINE = int(__import__("os").environ.get("XVA_XVP_WN_ENGINE", "1"))        # 0: the per-primitive sequencing of forward_seq / backward_seq (same kernels, one stream)


class _SeqArena:
    """One zeroed slab the sequences of a training iteration are carved from.  A Seq is born all-zero (structural pad rows, guard rows, the
    accumulators of the residual / skip paths rely on it); as torch.zeros each costs an allocation and a fill launch, ~500 per xVAPitch iteration.
    seq_arena_begin() — called by XVAPitchStep.generator_pass at the start of an iteration, when every sequence of the previous one is dead
    (its autograd graph has been consumed) — zeroes the part of the slab the last iteration used in ONE memset and starts handing it out
    again.  Without that call (module-level use, the tests) every Seq is its own torch.zeros, as before."""
    slab, cap, used, want, active = None, 0, 0, 0, False


_ARENA = _SeqArena()


def _dev(device):
    d = torch.device(device)
    return torch.device(d.type, torch.cuda.current_device()) if d.type == "cuda" and d.index is None else d


def seq_arena_begin(device):
    a = _ARENA
    need = max(a.want, a.used)
    device = _dev(device)
    if a.slab is None or need > a.cap or a.slab.device != device:
        a.cap = int(need * 1.25) + (64 << 20)
        a.slab = torch.zeros(a.cap, device=device, dtype=torch.uint8)
        Seq._views.clear()                   # the cached views pin the slab they were cut from: a regrown arena must not keep the old one alive
    elif a.used:
        a.slab[:a.used].zero_()
    a.used, a.want, a.active = 0, 0, True


def seq_arena_end():
    """stop carving (sequences created from here on own their storage; the ones handed out stay valid until the next seq_arena_begin)"""
    _ARENA.active = False


def _zeros(rows, Cc, device, dtype):
    a = _ARENA
    n = (rows * Cc * (2 if dtype == torch.bfloat16 else 4) + 255) // 256 * 256
    a.want += n
    if a.active and a.used + n <= a.cap and a.slab.device == _dev(device):
        t = a.slab[a.used:a.used + n].view(dtype)[:rows * Cc].view(rows, Cc)
        a.used += n
        return t
    return torch.zeros(rows, Cc, device=device, dtype=dtype)


class Seq:
    """Time-major sequence (B, PAD + T + PAD, C) with GUARD spare rows before / after (conv taps of the first / last item reach there)."""

    _views = {}          # (slab address, byte offset, B, T, C, dtype) -> (store, view): an iteration carves the SAME sequences from the arena as the one before

    def __init__(self, B, T, Cc, device, dtype):
        self.B, self.T, self.C, self.Tp = B, T, Cc, T + 2 * PAD
        self.dt = 1 if dtype == torch.bfloat16 else 0
        a = _ARENA
        rows = 2 * GUARD + B * self.Tp
        n = (rows * Cc * (2 if self.dt else 4) + 255) // 256 * 256
        if a.active and a.used + n <= a.cap and a.slab.device == _dev(device):
            # the tensor objects of a slab slice are reused across iterations (creating the slice and its two views cost ~8 us of host time, 270 times
            # an iteration); the memory itself was zeroed by seq_arena_begin()
            key = (a.slab.data_ptr(), a.used, B, T, Cc, self.dt)
            hit = Seq._views.get(key)
            if hit is None:
                if len(Seq._views) > 8192:
                    Seq._views.clear()
                store = a.slab[a.used:a.used + n].view(dtype)[:rows * Cc].view(rows, Cc)
                hit = Seq._views[key] = (store, store[GUARD:GUARD + B * self.Tp].view(B, self.Tp, Cc))
            a.want += n
            a.used += n
            self.store, self.view = hit
            return
        self.store = _zeros(rows, Cc, device, dtype)
        self.view = self.store[GUARD:GUARD + B * self.Tp].view(B, self.Tp, Cc)

    @property
    def rows(self):
        return self.B * self.Tp

    def off(self, row=0):
        return (GUARD + row) * self.C


_PREP = {}


def _prepared(key, make):
    """xva_gemm call sites of the sequence convolutions, prepared once per geometry (_lib.PreparedGemm)"""
    pg = _PREP.get(key)
    if pg is None:
        if len(_PREP) > 4096:
            _PREP.clear()
        pg = _PREP[key] = make()
    return pg


def conv_fwd(x, w_eff, bias, y, k, d, compute):
    """y = conv1d(x; k taps, dilation d, 'same' padding) + bias on all rows (pad rows of y are zeroed by the epilogue mask)."""
    Cin, Cout = x.C, y.C
    key = (0, x.B, x.T, Cin, Cout, k, d, compute, x.dt, y.dt, w_eff.dtype)
    def make():
        P = d * (k - 1) // 2
        return _lib.PreparedGemm(x.store, w_eff, y.store, x.rows, Cout, k * Cin, Cin, k * Cin, Cout, layout=_lib.GEMM_NT, compute=compute, bias=bias,
                                 a_offset=x.off(-P), c_offset=y.off(), a_seglen=Cin if k > 1 else 0, a_segadj=d * Cin - Cin if k > 1 else 0,
                                 mask_mode=_lib.MASK_PAD, Tp=x.Tp, mask_pad=PAD, mask_len=x.T)
    _prepared(key, make).run(x.store, w_eff, y.store, bias=bias)


def conv_bwd_data(dy, w_eff, dx, k, d, compute, accumulate):
    Cout, Cin = dy.C, dx.C
    key = (1, dy.B, dy.T, Cin, Cout, k, d, compute, dy.dt, dx.dt, w_eff.dtype, bool(accumulate))
    def make():
        P = d * (k - 1) // 2
        return _lib.PreparedGemm(dy.store, w_eff, dx.store, dy.rows, Cin, k * Cout, Cout, k * Cin, Cin, layout=_lib.GEMM_NN, compute=compute, a_offset=dy.off(P),
                                 c_offset=dx.off(), a_seglen=Cout if k > 1 else 0, a_segadj=-d * Cout - Cout if k > 1 else 0, seglen=Cout if k > 1 else 0,
                                 seg0=0, segstride=Cin if k > 1 else 0, accumulate=accumulate, mask_mode=_lib.MASK_PAD, Tp=dy.Tp, mask_pad=PAD, mask_len=dy.T)
    _prepared(key, make).run(dy.store, w_eff, dx.store)


def conv_bwd_weight(dy, x, dw, db, k, d, compute):
    """dw (Cout, k * Cin) fp32 += dy^T xcat ; db (Cout) += column sums of dy."""
    Cout, Cin = dy.C, x.C
    key = (2, dy.B, dy.T, Cin, Cout, k, d, compute, dy.dt, x.dt, _lib.raw_stream(dw.device))
    def make():
        P = d * (k - 1) // 2
        return _lib.PreparedGemm(dy.store, x.store, dw, Cout, k * Cin, dy.rows, Cout, Cin, k * Cin, layout=_lib.GEMM_TN, compute=compute, accumulate=True, splitk=0,
                                 a_offset=dy.off(), b_offset=x.off(-P), seglen=Cin if k > 1 else 0, seg0=0, segstride=d * Cin - Cin if k > 1 else 0,
                                 sk_ws=_lib.sk_scratch(dw.device))
    _prepared(key, make).run(dy.store, x.store, dw)
    _lib.check(lib.xva_hg_colsum(C.c_void_p(dy.view.data_ptr()), dy.dt, _lib.ptr(db), dy.rows, Cout, 1.0, _lib.stream_ptr()), "xva_hg_colsum")


class _WNConv:
    """One weight-normed Conv1d: parameters in the reference layout, the effective tap-major weight in the activation dtype."""

    def __init__(self, Cin, Cout, k, d, device, dtype, gen):
        self.Cin, self.Cout, self.k, self.d = Cin, Cout, k, d
        v = torch.randn(Cout, Cin, k, generator=gen) * (1.0 / (Cin * k) ** 0.5)
        self.p = {"bias": (torch.randn(Cout, generator=gen) * 0.05).to(device), "weight_g": v.reshape(Cout, -1).norm(dim=1).reshape(Cout, 1, 1).to(device),
                  "weight_v": v.to(device)}
        self.g = {n: torch.zeros_like(t) for n, t in self.p.items()}
        self.eff = torch.zeros(Cout, k * Cin, device=device, dtype=dtype)
        self.norm = torch.zeros(Cout, device=device)
        self.dweff = torch.zeros(Cout, k * Cin, device=device)
        self.dt = 1 if dtype == torch.bfloat16 else 0

    def reparam(self):
        _lib.check(lib.xva_hg_weight_norm_fwd(_lib.ptr(self.p["weight_v"]), _lib.ptr(self.p["weight_g"]), _lib.ptr(self.eff), None, _lib.ptr(self.norm), self.dt,
                                              0, self.Cout, self.Cin, self.k, 1, 0, _lib.stream_ptr()), "xva_hg_weight_norm_fwd")

    def reparam_bwd(self):
        _lib.check(lib.xva_hg_weight_norm_bwd(_lib.ptr(self.dweff), _lib.ptr(self.p["weight_v"]), _lib.ptr(self.p["weight_g"]), _lib.ptr(self.norm),
                                              _lib.ptr(self.g["weight_v"]), _lib.ptr(self.g["weight_g"]), 0, self.Cout, self.Cin, self.k, _lib.stream_ptr()),
                   "xva_hg_weight_norm_bwd")


class WN:
    def __init__(self, in_channels, hidden_channels, kernel_size, dilation_rate, num_layers, c_in_channels=0, dropout_p=0, weight_norm=True,
                 device="cuda", compute="fp32", seed=0):
        assert kernel_size % 2 == 1 and hidden_channels % 2 == 0
        if dropout_p or not weight_norm:
            raise NotImplementedError("WN: dropout_p > 0 / weight_norm=False are not used by xVAPitch and not built")
        if dilation_rate ** (num_layers - 1) * (kernel_size - 1) // 2 > PAD:
            raise NotImplementedError("WN: dilation beyond %d rows of structural padding" % PAD)
        self.H, self.k, self.rate, self.L, self.c_in = hidden_channels, kernel_size, dilation_rate, num_layers, c_in_channels
        self.device = torch.device(device)
        self.compute = 1 if compute == "bf16" else 0
        self.dtype = torch.bfloat16 if self.compute else torch.float32
        gen = torch.Generator().manual_seed(seed)
        H = self.H
        self.in_layers = [_WNConv(H, 2 * H, kernel_size, dilation_rate ** i, self.device, self.dtype, gen) for i in range(num_layers)]
        self.res_skip_layers = [_WNConv(H, 2 * H if i < num_layers - 1 else H, 1, 1, self.device, self.dtype, gen) for i in range(num_layers)]
        self.cond_layer = _WNConv(c_in_channels, 2 * H * num_layers, 1, 1, self.device, torch.float32, gen) if c_in_channels > 0 else None
        # the effective-weight gradients of all the stack's convolutions in ONE buffer (one memset per backward), and the weight-norm
        # reparametrisations (forward and backward) as one batched launch over the stack (csrc/hg_ops.hip xva_hg_weight_norm_batch)
        convs = [c for _, c in self._named()]
        self._dweff = torch.zeros(sum((c.dweff.numel() + 63) // 64 * 64 for c in convs), device=self.device)
        off = 0
        for c in convs:
            c.dweff = self._dweff[off:off + c.dweff.numel()].view(c.dweff.shape)
            off += (c.dweff.numel() + 63) // 64 * 64
        self._descs = None

    def _table(self):
        """the engine's pointer table (csrc/xvp_wn.hip); the bias tensors move once, when FlatGroupAdamW re-homes them into its arenas"""
        b0, g0 = self.in_layers[0].p["bias"], self.res_skip_layers[-1].g["bias"]
        key = (b0.data_ptr(), g0.data_ptr(), _lib.PARAM_EPOCH[0])
        if getattr(self, "_tab_key", None) != key:
            ptrs = []
            for i in range(self.L):
                a, r = self.in_layers[i], self.res_skip_layers[i]
                ptrs += [a.eff, a.p["bias"], r.eff, r.p["bias"], a.dweff, a.g["bias"], r.dweff, r.g["bias"]]
            for t in ptrs:
                if not t.is_contiguous() or not t.is_cuda:
                    raise _lib.XvaError("WN: parameters / gradients must be contiguous device tensors")
            self._tab, self._tab_key = (C.c_void_p * len(ptrs))(*[t.data_ptr() for t in ptrs]), key
        return self._tab

    def _wn_batch(self, backward):
        convs = [c for _, c in self._named()]
        key = tuple(t.data_ptr() for c in (convs[0], convs[-1]) for t in (c.p["weight_v"], c.g["weight_v"]))
        if self._descs is None or self._descs[0] != key:                      # the parameters move once (FlatGroupAdamW's arenas): rebuild then
            arr = (_WnDesc * len(convs))()
            for d, c in zip(arr, convs):
                d.v, d.g, d.norm, d.eff, d.effB = c.p["weight_v"].data_ptr(), c.p["weight_g"].data_ptr(), c.norm.data_ptr(), c.eff.data_ptr(), None
                d.dW, d.dv, d.dg = c.dweff.data_ptr(), c.g["weight_v"].data_ptr(), c.g["weight_g"].data_ptr()
                d.dt, d.kind, d.D0, d.D1, d.k, d.s, d.pconv, d.block0 = c.dt, 0, c.Cout, c.Cin, c.k, 1, 0, 0
            self._descs = (key, arr, len(convs))
        _lib.check(lib.xva_hg_weight_norm_batch(self._descs[1], self._descs[2], int(backward), _lib.stream_ptr()), "xva_hg_weight_norm_batch")

    # ---- reference state_dict (wavenet.py:62-82) ----
    def _named(self):
        out = []
        if self.cond_layer is not None:
            out.append(("cond_layer.", self.cond_layer))
        for i in range(self.L):
            out.append(("in_layers.%d." % i, self.in_layers[i]))
        for i in range(self.L):
            out.append(("res_skip_layers.%d." % i, self.res_skip_layers[i]))
        return out

    def state_dict(self):
        return {pre + n: t.clone() for pre, c in self._named() for n, t in c.p.items()}

    def load_state_dict(self, sd):
        for pre, c in self._named():
            for n in c.p:
                t = sd[pre + n]
                if tuple(t.shape) != tuple(c.p[n].shape):
                    raise ValueError("%s%s: checkpoint shape %s != %s" % (pre, n, tuple(t.shape), tuple(c.p[n].shape)))
                c.p[n].copy_(t.to(c.p[n]))

    def grads(self):
        return {pre + n: t for pre, c in self._named() for n, t in c.g.items()}

    def zero_grad(self):
        for _, c in self._named():
            for t in c.g.values():
                t.zero_()

    # ---- forward / backward over sequences ----
    def forward_seq(self, x, lens, g=None):
        """x: Seq (B, Tp, H) already masked; lens (B) int32; g (B, c_in) fp32 or None.  Returns the output Seq; keeps what backward needs."""
        B, H = x.B, self.H
        self._wn_batch(False)
        gc = None
        if self.cond_layer is not None:
            if g is None:
                raise ValueError("WN was built with c_in_channels=%d: conditioning g is required" % self.c_in)
            gc = torch.empty(B, 2 * H * self.L, device=self.device)
            gq = g.float().contiguous()
            _lib.gemm(gq, self.cond_layer.eff, gc, B, 2 * H * self.L, self.c_in, self.c_in, self.c_in, 2 * H * self.L, layout=_lib.GEMM_NT, compute=0,
                      bias=self.cond_layer.p["bias"])
            self._g_in = gq
        out = Seq(B, x.T, H, self.device, self.dtype)
        self._x, self._a, self._acts, self._gc, self._lens = [x], [], [], gc, lens
        if _WN_ENGINE and H % 8 == 0 and self.L <= 32:
            d = _WnDims(B, x.T, H, self.k, self.rate, self.L, x.dt, self.compute)
            n = int(lib.xva_xvp_wn_workspace_bytes(C.byref(d)))
            if n < 0:
                raise _lib.XvaError("xva_xvp_wn_workspace_bytes: %s" % lib.xva_last_error().decode())
            ws = _zeros((n + 3) // 4, 1, self.device, torch.float32)            # zeroed: from the iteration's sequence arena when there is one
            _lib.check(lib.xva_xvp_wn_forward(C.byref(d), self._table(), C.c_void_p(x.store.data_ptr()), C.c_void_p(out.store.data_ptr()), _lib.ptr(gc), _lib.ptr(lens),
                                              C.c_void_p(ws.data_ptr()), n, _lib.stream_ptr()), "xva_xvp_wn_forward")
            self._eng = (d, ws, n)
            return out
        self._eng = None
        cur = x
        for i in range(self.L):
            a = Seq(B, x.T, 2 * H, self.device, self.dtype)
            conv_fwd(cur, self.in_layers[i].eff, self.in_layers[i].p["bias"], a, self.k, self.rate ** i, self.compute)
            acts = Seq(B, x.T, H, self.device, self.dtype)
            gl = C.c_void_p(gc.data_ptr() + 4 * i * 2 * H) if gc is not None else None
            _lib.check(lib.xva_wn_gate_fwd(C.c_void_p(a.view.data_ptr()), gl, 2 * H * self.L, C.c_void_p(acts.view.data_ptr()), a.dt, B, a.Tp, H,
                                           _lib.stream_ptr()), "xva_wn_gate_fwd")
            # the gate maps the zero pad rows of `a` to tanh(g) * sigmoid(g) != 0 when conditioned: the 1x1 conv below masks its own pad rows,
            # and rows t >= len are masked by the split, so only `acts` itself needs its pads cleared for the weight gradient
            if gc is not None:
                _lib.check(lib.xva_seq_mask(C.c_void_p(acts.view.data_ptr()), acts.dt, B, acts.Tp, PAD, H, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")
            last = i == self.L - 1
            rs = Seq(B, x.T, H if last else 2 * H, self.device, self.dtype)
            conv_fwd(acts, self.res_skip_layers[i].eff, self.res_skip_layers[i].p["bias"], rs, 1, 1, self.compute)
            nxt = None if last else Seq(B, x.T, H, self.device, self.dtype)
            _lib.check(lib.xva_wn_res_skip_fwd(C.c_void_p(rs.view.data_ptr()), C.c_void_p(cur.view.data_ptr()), C.c_void_p(nxt.view.data_ptr()) if nxt else None,
                                               C.c_void_p(out.view.data_ptr()), a.dt, B, a.Tp, PAD, H, int(last), _lib.ptr(lens), _lib.stream_ptr()),
                       "xva_wn_res_skip_fwd")
            self._a.append(a)
            self._acts.append(acts)
            if nxt is not None:
                self._x.append(nxt)
                cur = nxt
        return out

    def backward_seq(self, d_out):
        """d_out: Seq gradient of the output (rows t >= len may hold anything: they are masked).  Returns (d_x Seq, d_g or None); parameter
        gradients accumulate into .grads()."""
        B, H, lens = d_out.B, self.H, self._lens
        dt = d_out.dt
        d_x = Seq(B, d_out.T, H, self.device, self.dtype)          # gradient w.r.t. the running residual stream (zero after the last layer)
        d_gc = torch.zeros(B, 2 * H * self.L, device=self.device) if self._gc is not None else None
        engine = getattr(self, "_eng", None) is not None
        if engine:
            d, ws, n = self._eng
            self._eng = None
            sk = _lib.sk_scratch(self.device)
            _lib.check(lib.xva_xvp_wn_backward(C.byref(d), self._table(), C.c_void_p(self._x[0].store.data_ptr()), C.c_void_p(d_out.store.data_ptr()),
                                               C.c_void_p(d_x.store.data_ptr()), _lib.ptr(self._gc), _lib.ptr(d_gc), _lib.ptr(lens), C.c_void_p(ws.data_ptr()), n,
                                               C.c_void_p(sk.data_ptr()), sk.numel(), _lib.stream_ptr()), "xva_xvp_wn_backward")
        for i in (() if engine else reversed(range(self.L))):
            last = i == self.L - 1
            d_rs = Seq(B, d_out.T, H if last else 2 * H, self.device, self.dtype)
            _lib.check(lib.xva_wn_res_skip_bwd(C.c_void_p(d_x.view.data_ptr()), C.c_void_p(d_out.view.data_ptr()), C.c_void_p(d_rs.view.data_ptr()), dt, B,
                                               d_out.Tp, PAD, H, int(last), _lib.ptr(lens), _lib.stream_ptr()), "xva_wn_res_skip_bwd")
            rsl, inl = self.res_skip_layers[i], self.in_layers[i]
            conv_bwd_weight(d_rs, self._acts[i], rsl.dweff, rsl.g["bias"], 1, 1, self.compute)
            d_acts = Seq(B, d_out.T, H, self.device, self.dtype)
            conv_bwd_data(d_rs, rsl.eff, d_acts, 1, 1, self.compute, accumulate=False)
            d_a = Seq(B, d_out.T, 2 * H, self.device, self.dtype)
            gl = C.c_void_p(self._gc.data_ptr() + 4 * i * 2 * H) if self._gc is not None else None
            _lib.check(lib.xva_wn_gate_bwd(C.c_void_p(self._a[i].view.data_ptr()), gl, 2 * H * self.L, C.c_void_p(d_acts.view.data_ptr()),
                                           C.c_void_p(d_a.view.data_ptr()), dt, B, d_out.Tp, H, _lib.stream_ptr()), "xva_wn_gate_bwd")
            if d_gc is not None:                                   # d(cond)[b] = sum over the item's rows of d_a (pad / dead rows carry zeros)
                _lib.check(lib.xva_seq_item_colsum(C.c_void_p(d_a.view.data_ptr()), dt, C.c_void_p(d_gc.data_ptr() + 4 * i * 2 * H), B, d_a.Tp, 2 * H,
                                                   2 * H * self.L, _lib.stream_ptr()), "xva_seq_item_colsum")
            conv_bwd_weight(d_a, self._x[i], inl.dweff, inl.g["bias"], self.k, self.rate ** i, self.compute)
            # d_x (residual path, already masked by res_skip_bwd's first half when not last) += conv^T(d_a)
            if not last:
                _lib.check(lib.xva_seq_mask(C.c_void_p(d_x.view.data_ptr()), dt, B, d_x.Tp, PAD, H, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")
            conv_bwd_data(d_a, inl.eff, d_x, self.k, self.rate ** i, self.compute, accumulate=True)
        d_g = None
        if d_gc is not None:
            cl = self.cond_layer
            _lib.gemm(d_gc, self._g_in, cl.dweff, 2 * H * self.L, self.c_in, B, 2 * H * self.L, self.c_in, self.c_in, layout=_lib.GEMM_TN, compute=0, accumulate=True)
            _lib.check(lib.xva_hg_colsum(_lib.ptr(d_gc), 0, _lib.ptr(cl.g["bias"]), B, 2 * H * self.L, 1.0, _lib.stream_ptr()), "xva_hg_colsum")
            d_g = torch.empty(B, self.c_in, device=self.device)
            _lib.gemm(d_gc, cl.eff, d_g, B, self.c_in, 2 * H * self.L, 2 * H * self.L, self.c_in, self.c_in, layout=_lib.GEMM_NN, compute=0)
        self._wn_batch(True)
        self._dweff.zero_()
        return d_x, d_g

    # ---- reference interface: x (B, H, T), x_mask (B, 1, T) from lengths, g (B, c_in, 1) ----
    def __call__(self, x, x_mask=None, g=None):
        return _WNFn.apply(x, g, self, _lens_of(x, x_mask), _grad_hook(x.device))


def _grad_hook(device):
    """A requires-grad scalar handed to every block-level autograd.Function: their backward accumulates the PARAMETER gradients into the block's
    own buffers, so it has to run even when no tensor input requires grad (the posterior encoder's inputs are data)."""
    return torch.zeros(1, device=device, requires_grad=True)


def _lens_of(x, x_mask):
    if x_mask is None:
        return torch.full((x.size(0),), x.size(2), device=x.device, dtype=torch.int32)
    return (x_mask.reshape(x.size(0), -1) != 0).sum(1).to(torch.int32).contiguous()


class _WNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, wn, lens, hook=None):
        _lib.require_cuda(x)
        B, H, T = x.shape
        xs = Seq(B, T, H, wn.device, wn.dtype)
        _lib.check(ops.lib.xva_bct_to_seq(_lib.ptr(x.float().contiguous()), C.c_void_p(xs.view.data_ptr()), xs.dt, B, H, T, PAD, None, _lib.stream_ptr()),
                   "xva_bct_to_seq")
        out = wn.forward_seq(xs, lens, g.reshape(B, -1) if g is not None else None)
        ctx.wn, ctx.dims, ctx.has_g = wn, (B, H, T), g is not None
        return ops.seq_to_bct(out.view, T, PAD)

    @staticmethod
    def backward(ctx, d_out):
        wn = ctx.wn
        B, H, T = ctx.dims
        ds = Seq(B, T, H, wn.device, wn.dtype)
        _lib.check(ops.lib.xva_bct_to_seq(_lib.ptr(d_out.float().contiguous()), C.c_void_p(ds.view.data_ptr()), ds.dt, B, H, T, PAD, None, _lib.stream_ptr()),
                   "xva_bct_to_seq")
        d_x, d_g = wn.backward_seq(ds)
        return ops.seq_to_bct(d_x.view, T, PAD), (d_g.reshape(B, -1, 1) if ctx.has_g else None), None, None, None


class _PlainConv1x1:
    """nn.Conv1d(Cin, Cout, 1): weight (Cout, Cin, 1), bias (Cout) — the coupling block's `pre` / `post`."""

    def __init__(self, Cin, Cout, device, dtype, gen, zero=False):
        self.Cin, self.Cout = Cin, Cout
        w = torch.zeros(Cout, Cin, 1) if zero else torch.randn(Cout, Cin, 1, generator=gen) * (1.0 / Cin ** 0.5)
        self.p = {"weight": w.to(device), "bias": (torch.zeros(Cout) if zero else torch.randn(Cout, generator=gen) * 0.05).to(device)}
        self.g = {n: torch.zeros_like(t) for n, t in self.p.items()}
        self.dtype = dtype

    def eff(self):
        return self.p["weight"].reshape(self.Cout, self.Cin).to(self.dtype).contiguous()


class ResidualCouplingBlock:
    """model.py:1476-1535 with mean_only=True: x0, x1 = split(x); h = pre(x0) * mask; h = WN(h, g); m = post(h) * mask;
    forward: x1' = m + x1 * mask (logdet = 0); reverse: x1' = (x1 - m) * mask."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, num_layers, dropout_p=0, cond_channels=0, out_channels_override=None,
                 mean_only=False, device="cuda", compute="fp32", seed=0):
        assert channels % 2 == 0, "channels should be divisible by 2"
        if not mean_only or out_channels_override:
            raise NotImplementedError("ResidualCouplingBlock: only mean_only=True without the expanded-flow projector is built (what xVAPitch trains)")
        self.half, self.hidden = channels // 2, hidden_channels
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(seed)
        self.enc = WN(hidden_channels, hidden_channels, kernel_size, dilation_rate, num_layers, c_in_channels=cond_channels, dropout_p=dropout_p,
                      device=device, compute=compute, seed=seed + 1)
        self.pre = _PlainConv1x1(self.half, hidden_channels, self.device, self.enc.dtype, gen)
        self.post = _PlainConv1x1(hidden_channels, self.half, self.device, self.enc.dtype, gen, zero=False)

    def state_dict(self):
        sd = {"pre." + n: t.clone() for n, t in self.pre.p.items()}
        sd.update({"enc." + k: v for k, v in self.enc.state_dict().items()})
        sd.update({"post." + n: t.clone() for n, t in self.post.p.items()})
        return sd

    def load_state_dict(self, sd):
        for n in self.pre.p:
            self.pre.p[n].copy_(sd["pre." + n])
            self.post.p[n].copy_(sd["post." + n])
        self.enc.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("enc.")})

    def grads(self):
        g = {"pre." + n: t for n, t in self.pre.g.items()}
        g.update({"enc." + k: v for k, v in self.enc.grads().items()})
        g.update({"post." + n: t for n, t in self.post.g.items()})
        return g

    def zero_grad(self):
        self.enc.zero_grad()
        for c in (self.pre, self.post):
            for t in c.g.values():
                t.zero_()

    def __call__(self, x, x_mask, g=None, reverse=False):
        out = _CouplingFn.apply(x, g, self, _lens_of(x, x_mask), bool(reverse), _grad_hook(x.device))
        if reverse:
            return out
        return out, torch.zeros(x.size(0), device=x.device, dtype=x.dtype)          # logdet = sum(log_scale) = 0 for mean_only


class ResidualCouplingBlocks:
    """The flow of xVAPitch (model.py:1358-1422, expanded_flow off): num_flows mean-only coupling blocks, the channel halves flipped after each
    (forward) / before each, last block first (reverse).  state_dict keys `flows.i.*`."""

    def __init__(self, channels, hidden_channels, kernel_size, dilation_rate, num_layers, num_flows=4, cond_channels=0, device="cuda", compute="fp32", seed=0):
        self.flows = [ResidualCouplingBlock(channels, hidden_channels, kernel_size, dilation_rate, num_layers, cond_channels=cond_channels, mean_only=True,
                                            device=device, compute=compute, seed=seed + 7 * i) for i in range(num_flows)]

    def state_dict(self):
        return {"flows.%d.%s" % (i, k): v for i, f in enumerate(self.flows) for k, v in f.state_dict().items()}

    def load_state_dict(self, sd):
        for i, f in enumerate(self.flows):
            pre = "flows.%d." % i
            f.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})

    def grads(self):
        return {"flows.%d.%s" % (i, k): v for i, f in enumerate(self.flows) for k, v in f.grads().items()}

    def zero_grad(self):
        for f in self.flows:
            f.zero_grad()

    def __call__(self, x, x_mask, g=None, reverse=False):
        if not reverse:
            for f in self.flows:
                x, _ = f(x, x_mask, g=g)
                x = torch.flip(x, [1])
        else:
            for f in reversed(self.flows):
                x = f(torch.flip(x, [1]), x_mask, g=g, reverse=True)
        return x


class _CouplingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, blk, lens, reverse, hook=None):
        _lib.require_cuda(x)
        B, Cc, T = x.shape
        half, hid, wn = blk.half, blk.hidden, blk.enc
        x0 = x[:, :half].float().contiguous()
        x1 = x[:, half:].float().contiguous()
        x0s = Seq(B, T, half, wn.device, wn.dtype)
        _lib.check(ops.lib.xva_bct_to_seq(_lib.ptr(x0), C.c_void_p(x0s.view.data_ptr()), x0s.dt, B, half, T, PAD, None, _lib.stream_ptr()), "xva_bct_to_seq")
        h = Seq(B, T, hid, wn.device, wn.dtype)
        conv_fwd(x0s, blk.pre.eff(), blk.pre.p["bias"], h, 1, 1, wn.compute)
        _lib.check(lib.xva_seq_mask(C.c_void_p(h.view.data_ptr()), h.dt, B, h.Tp, PAD, hid, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")
        hw = wn.forward_seq(h, lens, g.reshape(B, -1) if g is not None else None)
        stats = Seq(B, T, half, wn.device, wn.dtype)
        conv_fwd(hw, blk.post.eff(), blk.post.p["bias"], stats, 1, 1, wn.compute)
        _lib.check(lib.xva_seq_mask(C.c_void_p(stats.view.data_ptr()), stats.dt, B, stats.Tp, PAD, half, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")
        out1 = torch.empty(B, half, T, device=x.device, dtype=torch.float32)
        _lib.check(lib.xva_coupling_mean_only(C.c_void_p(stats.view.data_ptr()), _lib.ptr(x1), _lib.ptr(out1), stats.dt, B, half, T, PAD, _lib.ptr(lens),
                                              int(reverse), _lib.stream_ptr()), "xva_coupling_mean_only")
        ctx.blk, ctx.lens, ctx.reverse, ctx.dims, ctx.has_g = blk, lens, reverse, (B, Cc, T), g is not None
        ctx.saved = (x0s, h, hw)
        return torch.cat([x0, out1], 1).to(x.dtype)

    @staticmethod
    def backward(ctx, d_out):
        blk, lens, reverse = ctx.blk, ctx.lens, ctx.reverse
        B, Cc, T = ctx.dims
        half, hid, wn = blk.half, blk.hidden, blk.enc
        x0s, h, hw = ctx.saved
        d1 = d_out[:, half:].float().contiguous()
        d_x1 = torch.empty_like(d1)
        d_stats = Seq(B, T, half, wn.device, wn.dtype)
        _lib.check(lib.xva_coupling_mean_only_bwd(_lib.ptr(d1), _lib.ptr(d_x1), C.c_void_p(d_stats.view.data_ptr()), d_stats.dt, B, half, T, PAD, _lib.ptr(lens),
                                                  int(reverse), _lib.stream_ptr()), "xva_coupling_mean_only_bwd")
        conv_bwd_weight(d_stats, hw, blk.post.g["weight"].view(half, hid), blk.post.g["bias"], 1, 1, wn.compute)      # accumulates into the gradient
        d_hw = Seq(B, T, hid, wn.device, wn.dtype)
        conv_bwd_data(d_stats, blk.post.eff(), d_hw, 1, 1, wn.compute, accumulate=False)
        d_h, d_g = wn.backward_seq(d_hw)
        _lib.check(lib.xva_seq_mask(C.c_void_p(d_h.view.data_ptr()), d_h.dt, B, d_h.Tp, PAD, hid, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")
        conv_bwd_weight(d_h, x0s, blk.pre.g["weight"].view(hid, half), blk.pre.g["bias"], 1, 1, wn.compute)
        d_x0s = Seq(B, T, half, wn.device, wn.dtype)
        conv_bwd_data(d_h, blk.pre.eff(), d_x0s, 1, 1, wn.compute, accumulate=False)
        d_x0 = ops.seq_to_bct(d_x0s.view, T, PAD, into=d_out[:, :half].float().contiguous())
        return torch.cat([d_x0, d_x1], 1), (d_g.reshape(B, -1, 1) if ctx.has_g else None), None, None, None, None


class PosteriorEncoder:
    """model.py:1427-1475: x -> conv1x1 `pre` -> WN (non-causal, conditioned) -> conv1x1 `proj` -> [m | log_scale] -> z = (m + eps * exp(log_scale)) * mask.
    The input is the 513-bin linear spectrogram (xva-trainer_amd/mel.py:TorchSTFTMel.linear).  eps: the N(0, 1) draw (the reference calls
    torch.randn_like inside forward; pass it for reproducibility, else it is drawn here)."""

    def __init__(self, in_channels, out_channels, hidden_channels, kernel_size, dilation_rate, num_layers, cond_channels=0, device="cuda", compute="fp32", seed=0):
        self.Cin, self.Co, self.hidden = in_channels, out_channels, hidden_channels
        # the GEMM operands want rows of a multiple of 4 (fp32) / 8 (bf16) elements: the 513 spectrogram bins are carried as 520 channels, the
        # extra input channels and weight columns zero (they stay zero: their gradient is dy^T times a zero column)
        self.Cp = (in_channels + 7) // 8 * 8
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(seed)
        self.enc = WN(hidden_channels, hidden_channels, kernel_size, dilation_rate, num_layers, c_in_channels=cond_channels, device=device, compute=compute,
                      seed=seed + 1)
        self.pre = _PlainConv1x1(self.Cp, hidden_channels, self.device, self.enc.dtype, gen)
        self.pre.p["weight"][:, in_channels:] = 0
        self.proj = _PlainConv1x1(hidden_channels, 2 * out_channels, self.device, self.enc.dtype, gen)

    def state_dict(self):
        sd = {"pre.weight": self.pre.p["weight"][:, :self.Cin].clone(), "pre.bias": self.pre.p["bias"].clone()}
        sd.update({"enc." + k: v for k, v in self.enc.state_dict().items()})
        sd.update({"proj." + n: t.clone() for n, t in self.proj.p.items()})
        return sd

    def load_state_dict(self, sd):
        self.pre.p["weight"][:, :self.Cin].copy_(sd["pre.weight"])
        self.pre.p["bias"].copy_(sd["pre.bias"])
        for n in self.proj.p:
            self.proj.p[n].copy_(sd["proj." + n])
        self.enc.load_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("enc.")})

    def grads(self):
        g = {"pre.weight": self.pre.g["weight"][:, :self.Cin], "pre.bias": self.pre.g["bias"]}
        g.update({"enc." + k: v for k, v in self.enc.grads().items()})
        g.update({"proj." + n: t for n, t in self.proj.g.items()})
        return g

    def zero_grad(self):
        self.enc.zero_grad()
        for c in (self.pre, self.proj):
            for t in c.g.values():
                t.zero_()

    def __call__(self, x, x_lengths, g=None, eps=None):
        B, _, T = x.shape
        lens = x_lengths.reshape(B).to(device=x.device, dtype=torch.int32).contiguous()
        if eps is None:
            eps = torch.randn(B, self.Co, T, device=x.device)
        if self.Cp != self.Cin:
            x = torch.nn.functional.pad(x, (0, 0, 0, self.Cp - self.Cin))
        z, mean, logs = _PosteriorFn.apply(x, g, eps, self, lens, _grad_hook(x.device))
        x_mask = (torch.arange(T, device=x.device)[None, :] < lens[:, None]).to(x.dtype).unsqueeze(1)     # sequence_mask(x_lengths) as returned by the reference
        return z, mean, logs, x_mask


class _PosteriorFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, g, eps, enc, lens, hook=None):
        _lib.require_cuda(x, eps)
        B, Cin, T = x.shape
        wn, Co, hid = enc.enc, enc.Co, enc.hidden
        xs = Seq(B, T, Cin, wn.device, wn.dtype)
        _lib.check(ops.lib.xva_bct_to_seq(_lib.ptr(x.float().contiguous()), C.c_void_p(xs.view.data_ptr()), xs.dt, B, Cin, T, PAD, None, _lib.stream_ptr()),
                   "xva_bct_to_seq")
        h = Seq(B, T, hid, wn.device, wn.dtype)
        conv_fwd(xs, enc.pre.eff(), enc.pre.p["bias"], h, 1, 1, wn.compute)
        _lib.check(lib.xva_seq_mask(C.c_void_p(h.view.data_ptr()), h.dt, B, h.Tp, PAD, hid, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")
        hw = wn.forward_seq(h, lens, g.reshape(B, -1) if g is not None else None)
        stats = Seq(B, T, 2 * Co, wn.device, wn.dtype)
        conv_fwd(hw, enc.proj.eff(), enc.proj.p["bias"], stats, 1, 1, wn.compute)
        _lib.check(lib.xva_seq_mask(C.c_void_p(stats.view.data_ptr()), stats.dt, B, stats.Tp, PAD, 2 * Co, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")
        eps = eps.float().contiguous()
        z, mean, logs = (torch.empty(B, Co, T, device=x.device) for _ in range(3))
        _lib.check(lib.xva_posterior_sample(C.c_void_p(stats.view.data_ptr()), _lib.ptr(eps), _lib.ptr(z), _lib.ptr(mean), _lib.ptr(logs), stats.dt, B, Co, T, PAD,
                                            _lib.ptr(lens), _lib.stream_ptr()), "xva_posterior_sample")
        ctx.enc, ctx.lens, ctx.dims, ctx.has_g = enc, lens, (B, Cin, T), g is not None
        ctx.saved = (xs, h, hw, stats, eps)
        return z, mean, logs

    @staticmethod
    def backward(ctx, d_z, d_mean, d_logs):
        enc, lens = ctx.enc, ctx.lens
        B, Cin, T = ctx.dims
        wn, Co, hid = enc.enc, enc.Co, enc.hidden
        xs, h, hw, stats, eps = ctx.saved
        d_stats = Seq(B, T, 2 * Co, wn.device, wn.dtype)
        dz, dm, dl = (t.float().contiguous() if t is not None else None for t in (d_z, d_mean, d_logs))
        _lib.check(lib.xva_posterior_sample_bwd(C.c_void_p(stats.view.data_ptr()), _lib.ptr(eps), _lib.ptr(dz), _lib.ptr(dm), _lib.ptr(dl),
                                                C.c_void_p(d_stats.view.data_ptr()), d_stats.dt, B, Co, T, PAD, _lib.ptr(lens), _lib.stream_ptr()),
                   "xva_posterior_sample_bwd")
        conv_bwd_weight(d_stats, hw, enc.proj.g["weight"].view(2 * Co, hid), enc.proj.g["bias"], 1, 1, wn.compute)
        d_hw = Seq(B, T, hid, wn.device, wn.dtype)
        conv_bwd_data(d_stats, enc.proj.eff(), d_hw, 1, 1, wn.compute, accumulate=False)
        d_h, d_g = wn.backward_seq(d_hw)
        _lib.check(lib.xva_seq_mask(C.c_void_p(d_h.view.data_ptr()), d_h.dt, B, d_h.Tp, PAD, hid, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")
        conv_bwd_weight(d_h, xs, enc.pre.g["weight"].view(hid, Cin), enc.pre.g["bias"], 1, 1, wn.compute)
        d_xs = Seq(B, T, Cin, wn.device, wn.dtype)
        conv_bwd_data(d_h, enc.pre.eff(), d_xs, 1, 1, wn.compute, accumulate=False)
        return ops.seq_to_bct(d_xs.view, T, PAD), (d_g.reshape(B, -1, 1) if ctx.has_g else None), None, None, None, None
