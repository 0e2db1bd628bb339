"""RelativePositionTransformer of xVAPitch's text encoder on libxvahip — python/xvapitch/glow_tts.py:59-485 as TextEncoder builds it
(python/xvapitch/model.py:1125-1136: layer_norm_type "2", rel_attn_window_size 4, heads share the relative embeddings, in = hidden = out).

Same constructor arguments, state_dict keys and layouts as the reference module (`attn_layers.i.conv_{q,k,v,o}.{weight,bias}`,
`attn_layers.i.emb_rel_{k,v}`, `norm_layers_{1,2}.i.{gamma,beta}`, `ffn_layers.i.conv_{1,2}.{weight,bias}`), same (B, C, T) tensors and
(B, 1, T) mask at the interface.  Inside, activations are fp32 time-major sequences with structurally zero pad rows (xvapitch/wn.py:Seq);
the q / k / v projections are ONE xva_gemm against the stacked weights, conv_o and the k-tap feed-forward convolutions are xva_gemm calls in
implicit-conv form (ReLU and the residual in the epilogue, the ReLU gate in the backward epilogue), the attention core and LayerNorm are
csrc/xvapitch_ops.hip kernels.  Host code only sequences C calls; the module is one autograd Function, checked against the reference by
output and by every parameter / input gradient (tests/test_xvapitch_gpu.py).  dropout_p > 0: nn.Dropout at the reference's four sites per
layer — the attention weights (glow_tts.py:204, inside the attention kernels), the attention block's output (:473, the conv_o epilogue),
the feed-forward hidden activation (:344, the conv_1 epilogue) and the feed-forward output (:477) — with masks from the keyed hash of
csrc/xva_common.h: site `dropout_site_base + 4 * layer + {0, 1, 2, 3}` under the seed of set_dropout_seed() (a new one per training
iteration), element index = position in the (B, H, T, T) weights / (row of the time-major sequence) * channels + channel; `training =
False` switches it off (eval).  Not built: in_channels != hidden_channels, input_length, layer_norm type "1".
"""
import ctypes as C
import os

import torch

from .. import _lib
from . import ops
from .wn import PAD, Seq, conv_bwd_data, conv_bwd_weight, conv_fwd, _grad_hook, _lens_of, _zeros

lib = _lib.lib
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
lib.xva_relattn_fwd.restype = i32
lib.xva_relattn_fwd.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, i64] + [i32] * 8 + [f32, C.c_uint64, C.c_uint32, vp]
lib.xva_relattn_bwd.restype = i32
lib.xva_relattn_bwd.argtypes = [vp, i64, vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp] + [i32] * 8 + [f32, C.c_uint64, C.c_uint32, vp]
lib.xva_dropout_apply.restype = i32
lib.xva_dropout_apply.argtypes = [vp, vp, i32, i64, f32, C.c_uint64, C.c_uint32, vp]
lib.xva_ln_rows_fwd.restype = i32
lib.xva_ln_rows_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp]
lib.xva_ln_rows_bwd.restype = i32
lib.xva_ln_rows_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]


class _TrDims(C.Structure):
    """include/xva_hip.h xva_xvp_tr_dims"""
    _fields_ = [("B", i32), ("T", i32), ("C", i32), ("F", i32), ("H", i32), ("L", i32), ("k", i32), ("w", i32), ("Co", i32), ("has_proj", i32), ("compute", i32),
                ("p_drop", f32), ("seed", C.c_uint64), ("site0", C.c_uint32)]


PER_LAYER = 18          # XVA_XVP_TR_PER_LAYER: the order of the engine's parameter table
_ORDER = ("attn.conv_q.weight", "attn.conv_q.bias", "attn.conv_k.weight", "attn.conv_k.bias", "attn.conv_v.weight", "attn.conv_v.bias", "attn.conv_o.weight",
          "attn.conv_o.bias", "attn.emb_rel_k", "attn.emb_rel_v", "ffn.conv_1.weight", "ffn.conv_1.bias", "ffn.conv_2.weight", "ffn.conv_2.bias", "norm1.gamma",
          "norm1.beta", "norm2.gamma", "norm2.beta")
lib.xva_xvp_tr_workspace_bytes.restype = i64
lib.xva_xvp_tr_workspace_bytes.argtypes = [C.POINTER(_TrDims)]
lib.xva_xvp_tr_forward.restype = i32
lib.xva_xvp_tr_forward.argtypes = [C.POINTER(_TrDims), vp, vp, vp, vp, vp, i64, vp]
lib.xva_xvp_tr_backward.restype = i32
lib.xva_xvp_tr_backward.argtypes = [C.POINTER(_TrDims), vp, vp, vp, vp, vp, vp, i64, vp, i64, vp]
_ENGINE = os.environ.get("XVA_XVP_TR_ENGINE", "1") != "0"      # 0: the per-primitive Python sequencing below (same kernels; kept as the A / B and for the tests)


def _p(t, elem_off=0):
    return C.c_void_p(t.data_ptr() + 4 * elem_off)


def _mask(s, lens):
    _lib.check(lib.xva_seq_mask(C.c_void_p(s.view.data_ptr()), s.dt, s.B, s.Tp, PAD, s.C, _lib.ptr(lens), _lib.stream_ptr()), "xva_seq_mask")


def _drop(src, dst, p, seed, site):
    """dst = nn.Dropout(src) over the rows of a sequence's view (element index = row * C + channel); the same call on a gradient is the backward"""
    _lib.check(lib.xva_dropout_apply(C.c_void_p(src.view.data_ptr()), C.c_void_p(dst.view.data_ptr()), src.dt, src.rows * src.C, p, seed, site,
                                     _lib.stream_ptr()), "xva_dropout_apply")


def _tapmajor(w):
    """nn.Conv1d weight (Cout, Cin, k) -> (Cout, k * Cin)"""
    return w.permute(0, 2, 1).reshape(w.size(0), -1).contiguous()


class _Layer:
    def __init__(self, Cc, F, H, k, w, device, gen, Co=None):
        """Co: output width of conv_2 / norm2 (the last layer of a stack with out_channels != hidden_channels)"""
        dk = Cc // H
        Co = Cc if Co is None else Co

        def conv(co, ci, kk):
            bound = (1.0 / (ci * kk)) ** 0.5
            return {"weight": ((torch.rand(co, ci, kk, generator=gen) * 2 - 1) * bound).to(device), "bias": ((torch.rand(co, generator=gen) * 2 - 1) * bound).to(device)}
        self.p = {}
        for n in ("q", "k", "v", "o"):
            for a, t in conv(Cc, Cc, 1).items():
                self.p["attn.conv_%s.%s" % (n, a)] = t
        self.p["attn.emb_rel_k"] = (torch.randn(1, 2 * w + 1, dk, generator=gen) * dk ** -0.5).to(device)
        self.p["attn.emb_rel_v"] = (torch.randn(1, 2 * w + 1, dk, generator=gen) * dk ** -0.5).to(device)
        for a, t in conv(F, Cc, k).items():
            self.p["ffn.conv_1." + a] = t
        for a, t in conv(Co, F, k).items():
            self.p["ffn.conv_2." + a] = t
        for n, ch in (("norm1", Cc), ("norm2", Co)):
            self.p[n + ".gamma"] = torch.ones(ch, device=device)
            self.p[n + ".beta"] = torch.zeros(ch, device=device)
        self.g = {n: torch.zeros_like(t) for n, t in self.p.items()}


_KEYMAP = (("attn.", "attn_layers.%d."), ("ffn.", "ffn_layers.%d."), ("norm1.", "norm_layers_1.%d."), ("norm2.", "norm_layers_2.%d."))


class RelativePositionTransformer:
    def __init__(self, in_channels, out_channels, hidden_channels, hidden_channels_ffn, num_heads, num_layers, kernel_size=1, dropout_p=0.0,
                 rel_attn_window_size=None, input_length=None, layer_norm_type="1", device="cuda", seed=0, compute="fp32", dropout_site_base=0):
        if not 0.0 <= dropout_p < 1.0:
            raise ValueError("RelativePositionTransformer: dropout_p must be in [0, 1)")
        self.dropout_p, self.site0, self.training, self.drop_seed = float(dropout_p), int(dropout_site_base), True, int(seed) + 0x5EED
        if in_channels != hidden_channels or not (out_channels == 1 or out_channels % 4 == 0):
            raise NotImplementedError("RelativePositionTransformer: in_channels must equal hidden_channels, out_channels 1 or a multiple of 4")
        if rel_attn_window_size is None or input_length is not None or layer_norm_type != "2":
            raise NotImplementedError("RelativePositionTransformer: built for rel_attn_window_size set, input_length None, layer_norm_type '2'")
        if hidden_channels % num_heads or hidden_channels % 4 or hidden_channels_ffn % 4 or kernel_size % 2 != 1 or kernel_size // 2 > PAD:
            raise NotImplementedError("RelativePositionTransformer: channels must be multiples of 4 and of num_heads, kernel_size odd")
        self.C, self.F, self.H, self.L, self.k, self.w = hidden_channels, hidden_channels_ffn, num_heads, num_layers, kernel_size, rel_attn_window_size
        self.Co = out_channels
        # "fp32": exact-fp32 MFMA products (the parity mode) ; "mixed": the same fp32-stored tensors, operands rounded to bf16 while staged
        # (bf16 MFMA, fp32 accumulation) — the throughput mode of the projections and feed-forward convolutions; attention / LayerNorm stay fp32
        # "split": fp32 storage, every product three bf16 MFMAs on hi + lo split operands (xva_gemm compute 2: ~1e-5 per product at 3 / 16 of the exact pipe's time)
        if compute not in ("fp32", "mixed", "split"):
            raise ValueError("RelativePositionTransformer: compute must be 'fp32', 'mixed' or 'split'")
        self.cmp = {"fp32": 0, "mixed": 1, "split": 2}[compute]
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(seed)
        self.layers = [_Layer(self.C, self.F, self.H, self.k, self.w, self.device, gen, Co=out_channels if i == num_layers - 1 else None)
                       for i in range(num_layers)]
        # hidden != out: `proj` maps the last layer's attention block output to the output width (glow_tts.py:440-441,479-480); with
        # out_channels == 1 (the pitch / energy encoders, model.py:1292-1305) the stack RETURNS proj(x) and the last layer's feed-forward
        # network and second LayerNorm do not reach the output (:482): they are not evaluated here and receive no gradient.
        self.proj = None
        if out_channels != hidden_channels:
            bound = (1.0 / hidden_channels) ** 0.5
            self.proj = {"weight": ((torch.rand(out_channels, hidden_channels, 1, generator=gen) * 2 - 1) * bound).to(self.device),
                         "bias": ((torch.rand(out_channels, generator=gen) * 2 - 1) * bound).to(self.device)}
            self.proj_g = {n: torch.zeros_like(t) for n, t in self.proj.items()}

    # ---- reference state_dict ----
    def _named(self, which):
        for i, l in enumerate(self.layers):
            for n, t in getattr(l, which).items():
                for a, b in _KEYMAP:
                    if n.startswith(a):
                        yield (b % i) + n[len(a):], t
                        break
        if self.proj is not None:
            for n, t in (self.proj if which == "p" else self.proj_g).items():
                yield "proj." + n, t

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self._named("p")}

    def load_state_dict(self, sd):
        mine = dict(self._named("p"))
        if set(mine) != set(sd):
            raise KeyError("RelativePositionTransformer.load_state_dict: key mismatch %s" % sorted(set(mine) ^ set(sd))[:6])
        for k, t in mine.items():
            if tuple(t.shape) != tuple(sd[k].shape):
                raise ValueError("%s: shape %s != %s" % (k, tuple(sd[k].shape), tuple(t.shape)))
            t.copy_(sd[k].to(device=self.device, dtype=torch.float32))

    def grads(self):
        return {k: v for k, v in self._named("g")}

    def zero_grad(self):
        for l in self.layers:
            for t in l.g.values():
                t.zero_()
        if self.proj is not None:
            for t in self.proj_g.values():
                t.zero_()

    def set_dropout_seed(self, seed):
        """the seed of the NEXT forward's dropout masks (its backward reuses it); the trainer passes a new one every iteration"""
        self.drop_seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    def train(self, mode=True):
        self.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def __call__(self, x, x_mask):
        return _TransformerFn.apply(x, self, _lens_of(x, x_mask), _grad_hook(x.device))

    def start(self, x, x_mask):
        """Issue the forward pass NOW without creating the autograd node; attach() creates it later.  autograd runs backward nodes in reverse order of
        creation: a stack whose forward has to be issued first (the text encoder: the alignment waits for it) would otherwise have its backward issued last.
        Returns a handle for attach(), or None when the engine path is off (attach() then runs the forward pass itself)."""
        if not _ENGINE:
            return None
        lens = _lens_of(x, x_mask)
        with torch.no_grad():
            xd = x.detach().float().contiguous()
            B, _, T = xd.shape
            d = self._dims(B, T)
            n = int(lib.xva_xvp_tr_workspace_bytes(C.byref(d)))
            if n < 0:
                raise _lib.XvaError("xva_xvp_tr_workspace_bytes: %s" % lib.xva_last_error().decode())
            ws = _zeros((n + 3) // 4, 1, self.device, torch.float32)
            out = torch.empty(B, self.Co, T, device=xd.device)
            prm, _ = self._tables()
            _lib.check(lib.xva_xvp_tr_forward(C.byref(d), prm, _lib.ptr(xd), _lib.ptr(lens), _lib.ptr(out), C.c_void_p(ws.data_ptr()), n, _lib.stream_ptr()),
                       "xva_xvp_tr_forward")
        return (out, (d, ws, n, lens), lens)

    def attach(self, x, x_mask, handle):
        """the autograd node of a forward pass start() issued (same x): call on the stream start() ran on"""
        if handle is None:
            return self(x, x_mask)
        out, state, lens = handle
        return _TransformerFn.apply(x, self, lens, _grad_hook(x.device), (out, state))

    # ---- the engine calls (csrc/xvp_transformer.hip) ----
    def _tables(self):
        """(parameter, gradient) pointer tables in the engine's order; rebuilt when the tensors have moved (FlatGroupAdamW re-homes them into its arenas once)."""
        first, lastp = self.layers[0].p[_ORDER[0]], self.layers[-1].p[_ORDER[-1]]
        key = (first.data_ptr(), lastp.data_ptr(), self.layers[0].g[_ORDER[0]].data_ptr(), self.layers[-1].g[_ORDER[-1]].data_ptr(), _lib.PARAM_EPOCH[0])
        if getattr(self, "_tab_key", None) != key:
            ps = [l.p[n] for l in self.layers for n in _ORDER] + ([self.proj["weight"], self.proj["bias"]] if self.proj is not None else [])
            gs = [l.g[n] for l in self.layers for n in _ORDER] + ([self.proj_g["weight"], self.proj_g["bias"]] if self.proj is not None else [])
            for t in ps + gs:
                if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                    raise _lib.XvaError("RelativePositionTransformer: parameters / gradients must be contiguous fp32 device tensors")
            arr = C.c_void_p * len(ps)
            self._tab = (arr(*[t.data_ptr() for t in ps]), arr(*[t.data_ptr() for t in gs]))
            self._tab_key = key
        return self._tab

    def _dims(self, B, T):
        pd = self.dropout_p if self.training else 0.0
        return _TrDims(B, T, self.C, self.F, self.H, self.L, self.k, self.w, self.Co, int(self.proj is not None), self.cmp, pd, self.drop_seed, self.site0)

    # ---- sequences ----
    def _proj_padded(self):
        """proj weight (rows padded with zeros to a multiple of 4: out_channels == 1) and bias, as GEMM operands"""
        Co, Cc = self.Co, self.C
        Cp = (Co + 3) // 4 * 4
        w = torch.zeros(Cp, Cc, device=self.device); w[:Co] = self.proj["weight"].reshape(Co, Cc)
        b = torch.zeros(Cp, device=self.device); b[:Co] = self.proj["bias"]
        return w, b

    def _ln(self, x, gamma, beta):
        y = Seq(x.B, x.T, x.C, self.device, torch.float32)
        mean = torch.empty(x.rows, device=self.device); rstd = torch.empty(x.rows, device=self.device)
        _lib.check(lib.xva_ln_rows_fwd(_p(x.view), _lib.ptr(gamma), _lib.ptr(beta), _p(y.view), _lib.ptr(mean), _lib.ptr(rstd), x.rows, x.C, 1e-5,
                                       _lib.stream_ptr()), "xva_ln_rows_fwd")
        return y, mean, rstd

    def forward_seq(self, x, lens):
        B, T, Cc, F, H, k = x.B, x.T, self.C, self.F, self.H, self.k
        dk = Cc // H
        self.saved = []
        mk = lambda ch: Seq(B, T, ch, self.device, torch.float32)
        pd, seed = (self.dropout_p if self.training else 0.0), self.drop_seed
        self.saved_drop = (pd, seed)
        for li, l in enumerate(self.layers):
            p = l.p
            site = self.site0 + 4 * li
            dkw = lambda s_: dict(drop_p=pd, drop_seed=seed, drop_stream=site + s_) if pd > 0 else {}
            last = li == self.L - 1
            Co = self.Co if last else Cc
            xm = mk(Cc); xm.store.copy_(x.store); _mask(xm, lens)                               # x = x * x_mask            (glow_tts.py:471)
            wqkv = torch.cat([p["attn.conv_%s.weight" % n].reshape(Cc, Cc) for n in "qkv"], 0).contiguous()
            bqkv = torch.cat([p["attn.conv_%s.bias" % n] for n in "qkv"]).contiguous()
            qkv = mk(3 * Cc)
            conv_fwd(xm, wqkv, bqkv, qkv, 1, 1, self.cmp)                                              # conv_q / conv_k / conv_v  (:166-168)
            P = torch.empty(B, H, T, T, device=self.device)
            att = mk(Cc)
            base = PAD * 3 * Cc
            _lib.check(lib.xva_relattn_fwd(_p(qkv.view), _p(qkv.view, Cc), _p(qkv.view, 2 * Cc), 3 * Cc, _lib.ptr(p["attn.emb_rel_k"]),
                                           _lib.ptr(p["attn.emb_rel_v"]), _lib.ptr(lens), _lib.ptr(P), _p(att.view), Cc, B, T, H, dk, self.w, 1, x.Tp, PAD,
                                           pd, seed, site, _lib.stream_ptr()), "xva_relattn_fwd")   # attention, dropout(p_attn)  (:173-214)
            s1 = mk(Cc)
            wo = p["attn.conv_o.weight"].reshape(Cc, Cc).contiguous()
            _lib.gemm(att.store, wo, s1.store, att.rows, Cc, Cc, Cc, Cc, Cc, layout=_lib.GEMM_NT, compute=self.cmp, bias=p["attn.conv_o.bias"], a_offset=att.off(),
                      c_offset=s1.off(), R=xm.view, ldr=Cc, mask_mode=_lib.MASK_PAD, Tp=x.Tp, mask_pad=PAD, mask_len=T, **dkw(1))   # x + dropout(conv_o(..))  (:170,473-474)
            x1, m1, r1 = self._ln(s1, p["norm1.gamma"], p["norm1.beta"])                        # norm_layers_1             (:474)
            res = x1                                                                            # the residual branch of the second sub-layer
            if last and self.proj is not None:                                                  # x = proj(x)               (:479-480)
                wp, bp = self._proj_padded()
                res = mk(wp.size(0))
                conv_fwd(x1, wp, bp, res, 1, 1, self.cmp)
            if last and Co == 1:                                                                # the stack returns proj(x) (:482)
                self.saved.append((xm, wqkv, qkv, P, att, wo, s1, m1, r1, x1))
                x = res
                continue
            x1m = mk(Cc); x1m.store.copy_(x1.store); _mask(x1m, lens)                           # FFN: conv_1(pad(x * x_mask)), relu  (:342-343)
            h = mk(F)
            P_ = (k - 1) // 2
            w1 = _tapmajor(p["ffn.conv_1.weight"]); w2 = _tapmajor(p["ffn.conv_2.weight"])
            _lib.gemm(x1m.store, w1, h.store, x1m.rows, F, k * Cc, Cc, k * Cc, F, layout=_lib.GEMM_NT, compute=self.cmp, bias=p["ffn.conv_1.bias"], relu=True,
                      a_offset=x1m.off(-P_), c_offset=h.off(), a_seglen=Cc if k > 1 else 0, a_segadj=0, mask_mode=_lib.MASK_PAD, Tp=x.Tp, mask_pad=PAD,
                      mask_len=T, **dkw(2))                                                     # dropout(relu(.)) = relu(dropout(.)): the scale is >= 0  (:343-344)
            _mask(h, lens)                                                                      # conv_2(pad(x * x_mask)) * x_mask    (:345-346)
            y2 = mk(Co)
            conv_fwd(h, w2, p["ffn.conv_2.bias"], y2, k, 1, self.cmp)
            _mask(y2, lens)
            if pd > 0:
                _drop(y2, y2, pd, seed, site + 3)                                               # y = dropout(ffn(x))      (:477)
            s2 = mk(Co)
            torch.add(res.store, y2.store, out=s2.store)                                        # norm_layers_2(x + y)     (:482)
            x2, m2, r2 = self._ln(s2, p["norm2.gamma"], p["norm2.beta"])
            self.saved.append((xm, wqkv, qkv, P, att, wo, s1, m1, r1, x1, x1m, w1, w2, h, s2, m2, r2))
            x = x2
        out = mk(self.Co); out.store.copy_(x.store[:, :self.Co]); _mask(out, lens)              # x * x_mask               (:483)
        self.lens = lens
        return out

    def backward_seq(self, d_out):
        B, T, Cc, F, H, k = d_out.B, d_out.T, self.C, self.F, self.H, self.k
        dk = Cc // H
        lens = self.lens
        mk = lambda ch: Seq(B, T, ch, self.device, torch.float32)
        pd, seed = self.saved_drop
        dx = mk(self.Co); dx.store.copy_(d_out.store); _mask(dx, lens)

        def proj_bwd(dres, x1):
            """dres = d proj(x1): accumulates the proj gradients, returns d x1"""
            Co = self.Co
            wp, _ = self._proj_padded()
            Cp = wp.size(0)
            if Cp != Co:                                           # one output channel rides in a 4-wide sequence (GEMM leading dimensions)
                wide = mk(Cp); wide.store[:, :Co] = dres.store; dres = wide
            dWp = torch.zeros(Cp, Cc, device=self.device); dbp = torch.zeros(Cp, device=self.device)
            conv_bwd_weight(dres, x1, dWp, dbp, 1, 1, self.cmp)
            self.proj_g["weight"] += dWp[:Co].view(Co, Cc, 1)
            self.proj_g["bias"] += dbp[:Co]
            d = mk(Cc)
            conv_bwd_data(dres, wp, d, 1, 1, self.cmp, False)
            return d
        for li, l, sv in zip(reversed(range(self.L)), reversed(self.layers), reversed(self.saved)):
            p, g = l.p, l.g
            last = li == self.L - 1
            Co = self.Co if last else Cc
            site = self.site0 + 4 * li
            dkw = lambda s_: dict(drop_p=pd, drop_seed=seed, drop_stream=site + s_) if pd > 0 else {}
            if last and Co == 1:
                xm, wqkv, qkv, P, att, wo, s1, m1, r1, x1 = sv
                dx1 = proj_bwd(dx, x1)
            else:
                xm, wqkv, qkv, P, att, wo, s1, m1, r1, x1, x1m, w1, w2, h, s2, m2, r2 = sv
            ds2 = None
            if not (last and Co == 1):
                ds2 = mk(Co)
                _lib.check(lib.xva_ln_rows_bwd(_p(dx.view), _p(s2.view), _lib.ptr(m2), _lib.ptr(r2), _lib.ptr(p["norm2.gamma"]), _p(ds2.view), _lib.ptr(g["norm2.gamma"]),
                                               _lib.ptr(g["norm2.beta"]), dx.rows, Co, _lib.stream_ptr()), "xva_ln_rows_bwd")
                dy2 = mk(Co); dy2.store.copy_(ds2.store); _mask(dy2, lens)                       # y2 = dropout(conv_2(..) * x_mask)
                if pd > 0:
                    _drop(dy2, dy2, pd, seed, site + 3)
                # one zero fill for the layer's tap-major / concatenated weight-gradient scratch (dW2 | dW1 | dWqkv | dbqkv) instead of one per tensor
                scr = torch.zeros(Co * k * F + F * k * Cc + 3 * Cc * Cc + 3 * Cc, device=self.device)
                dW2 = scr[:Co * k * F].view(Co, k * F)
                conv_bwd_weight(dy2, h, dW2, g["ffn.conv_2.bias"], k, 1, self.cmp)
                g["ffn.conv_2.weight"] += dW2.view(Co, k, F).permute(0, 2, 1)
                dh = mk(F)
                P_ = (k - 1) // 2
                # d(conv_1 output) = (dy2 (*) W2) gated by relu (h is stored masked and post-ReLU: h > 0 is both the gate and the mask)
                _lib.gemm(dy2.store, w2, dh.store, dy2.rows, F, k * Co, Co, k * F, F, layout=_lib.GEMM_NN, compute=self.cmp, a_offset=dy2.off(P_), c_offset=dh.off(),
                          a_seglen=Co if k > 1 else 0, a_segadj=-2 * Co if k > 1 else 0, seglen=Co if k > 1 else 0, seg0=0, segstride=F if k > 1 else 0,
                          G=h.view, ldg=F, gate_slope=0.0, mask_mode=_lib.MASK_PAD, Tp=dy2.Tp, mask_pad=PAD, mask_len=T, **dkw(2))
                dW1 = scr[Co * k * F:Co * k * F + F * k * Cc].view(F, k * Cc)
                conv_bwd_weight(dh, x1m, dW1, g["ffn.conv_1.bias"], k, 1, self.cmp)
                g["ffn.conv_1.weight"] += dW1.view(F, k, Cc).permute(0, 2, 1)
                dx1m = mk(Cc)
                conv_bwd_data(dh, w1, dx1m, k, 1, self.cmp, False)
                _mask(dx1m, lens)                                                                # x1m = x1 * x_mask
                dres = proj_bwd(ds2, x1) if (last and self.proj is not None) else ds2            # residual branch: x or proj(x)
                dx1 = mk(Cc)
                torch.add(dres.store, dx1m.store, out=dx1.store)
            ds1 = mk(Cc)
            _lib.check(lib.xva_ln_rows_bwd(_p(dx1.view), _p(s1.view), _lib.ptr(m1), _lib.ptr(r1), _lib.ptr(p["norm1.gamma"]), _p(ds1.view), _lib.ptr(g["norm1.gamma"]),
                                           _lib.ptr(g["norm1.beta"]), dx1.rows, Cc, _lib.stream_ptr()), "xva_ln_rows_bwd")
            dyo = ds1                                                                            # s1 = x + dropout(conv_o(att))
            if pd > 0:
                dyo = mk(Cc); _drop(ds1, dyo, pd, seed, site + 1)
            conv_bwd_weight(dyo, att, g["attn.conv_o.weight"].view(Cc, Cc), g["attn.conv_o.bias"], 1, 1, self.cmp)   # a 1x1 conv: the gradient buffer IS the GEMM's C
            datt = mk(Cc)
            conv_bwd_data(dyo, wo, datt, 1, 1, self.cmp, False)
            dqkv = mk(3 * Cc)
            dS = torch.empty_like(P)
            demb_k = g["attn.emb_rel_k"]; demb_v = g["attn.emb_rel_v"]
            _lib.check(lib.xva_relattn_bwd(_p(datt.view), Cc, _p(qkv.view), _p(qkv.view, Cc), _p(qkv.view, 2 * Cc), 3 * Cc, _lib.ptr(p["attn.emb_rel_k"]),
                                           _lib.ptr(p["attn.emb_rel_v"]), _lib.ptr(lens), _lib.ptr(P), _lib.ptr(dS), _p(dqkv.view), _p(dqkv.view, Cc),
                                           _p(dqkv.view, 2 * Cc), 3 * Cc, _lib.ptr(demb_k), _lib.ptr(demb_v), B, T, H, dk, self.w, 1, dx.Tp, PAD,
                                           pd, seed, site, _lib.stream_ptr()), "xva_relattn_bwd")
            if last and Co == 1:
                scr = torch.zeros(3 * Cc * Cc + 3 * Cc, device=self.device)
            dWqkv = scr[scr.numel() - 3 * Cc * Cc - 3 * Cc:scr.numel() - 3 * Cc].view(3 * Cc, Cc); dbqkv = scr[scr.numel() - 3 * Cc:]
            conv_bwd_weight(dqkv, xm, dWqkv, dbqkv, 1, 1, self.cmp)
            torch._foreach_add_([g["attn.conv_%s.weight" % n] for n in "qkv"] + [g["attn.conv_%s.bias" % n] for n in "qkv"],
                                [dWqkv[j * Cc:(j + 1) * Cc].view(Cc, Cc, 1) for j in range(3)] + [dbqkv[j * Cc:(j + 1) * Cc] for j in range(3)])   # one launch for the six
            dxm = mk(Cc)
            conv_bwd_data(dqkv, wqkv, dxm, 1, 1, self.cmp, False)
            dxm.store += ds1.store                                                               # residual x + y
            _mask(dxm, lens)                                                                     # xm = x * x_mask
            dx = dxm
        return dx


class _TransformerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tr, lens, hook=None, started=None):
        _lib.require_cuda(x)
        B, Cc, T = x.shape
        if started is not None:                      # RelativePositionTransformer.start() has already run the forward pass on this input
            out, state = started
            ctx.tr, ctx.state = tr, state
            return out
        if _ENGINE:
            d = tr._dims(B, T)
            n = int(lib.xva_xvp_tr_workspace_bytes(C.byref(d)))
            if n < 0:
                raise _lib.XvaError("xva_xvp_tr_workspace_bytes: %s" % lib.xva_last_error().decode())
            ws = _zeros((n + 3) // 4, 1, tr.device, torch.float32)          # zeroed: from the iteration's sequence arena when there is one
            out = torch.empty(B, tr.Co, T, device=x.device)
            prm, _ = tr._tables()
            _lib.check(lib.xva_xvp_tr_forward(C.byref(d), prm, _lib.ptr(x.float().contiguous()), _lib.ptr(lens), _lib.ptr(out), C.c_void_p(ws.data_ptr()), n,
                                              _lib.stream_ptr()), "xva_xvp_tr_forward")
            ctx.tr, ctx.state = tr, (d, ws, n, lens)
            return out
        xs = Seq(B, T, Cc, tr.device, torch.float32)
        _lib.check(ops.lib.xva_bct_to_seq(_lib.ptr(x.float().contiguous()), C.c_void_p(xs.view.data_ptr()), 0, B, Cc, T, PAD, None, _lib.stream_ptr()),
                   "xva_bct_to_seq")
        out = tr.forward_seq(xs, lens)
        ctx.tr, ctx.dims = tr, (B, Cc, T)
        return ops.seq_to_bct(out.view, T, PAD)

    @staticmethod
    def backward(ctx, d_out):
        tr = ctx.tr
        if getattr(ctx, "state", None) is not None:
            d, ws, n, lens = ctx.state
            ctx.state = None
            d_x = torch.empty(d.B, tr.C, d.T, device=d_out.device)
            prm, grd = tr._tables()
            sk = _lib.sk_scratch(d_out.device)
            _lib.check(lib.xva_xvp_tr_backward(C.byref(d), prm, grd, _lib.ptr(d_out.float().contiguous()), _lib.ptr(lens), _lib.ptr(d_x), C.c_void_p(ws.data_ptr()),
                                               n, C.c_void_p(sk.data_ptr()), sk.numel(), _lib.stream_ptr()), "xva_xvp_tr_backward")
            return d_x, None, None, None, None
        B, Cc, T = ctx.dims
        ds = Seq(B, T, tr.Co, tr.device, torch.float32)
        _lib.check(ops.lib.xva_bct_to_seq(_lib.ptr(d_out.float().contiguous()), C.c_void_p(ds.view.data_ptr()), 0, B, tr.Co, T, PAD, None, _lib.stream_ptr()),
                   "xva_bct_to_seq")
        d_x = tr.backward_seq(ds)
        return ops.seq_to_bct(d_x.view, T, PAD), None, None, None, None
