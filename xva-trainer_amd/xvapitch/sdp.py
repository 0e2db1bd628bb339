"""Building blocks of xVAPitch's stochastic duration predictor on libxvahip — python/xvapitch/sdp.py:
DilatedDepthSeparableConv (:40-93), ElementwiseAffine (:95-114), ConvFlow with its rational-quadratic spline (:116-176, util.py:203-391),
StochasticDurationPredictor (:179-310, forward / training direction).

Same constructor arguments and state_dict keys / layouts as the reference modules, same (B, C, T) tensors and (B, 1, T) mask at the interface.
Every arithmetic step is a C call wrapped as ONE autograd primitive (depthwise dilated convolution, LayerNorm2, exact GELU, 1x1 convolution =
xva_gemm, mask, add), so the blocks compose with torch autograd like the reference's; parameters are leaf tensors whose `.grad` autograd fills.
Inside, tensors are fp32 time-major (B, T, C); splits / concatenations / flips of the 2-channel flow variable and the per-item sums of
log-determinants are torch view / reduction glue.  StochasticDurationPredictor.forward (the training likelihood, :247-310) is assembled from these primitives; its reverse (sampling) direction (:311-321) is StochasticDurationPredictor.infer.
"""
import ctypes as C

import torch

from .. import _lib
from .wn import _lens_of

lib = _lib.lib
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
lib.xva_dwconv_fwd.restype = i32
lib.xva_dwconv_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
lib.xva_dwconv_bwd.restype = i32
lib.xva_dwconv_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
lib.xva_dropout_apply.restype = i32
lib.xva_dropout_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, C.c_void_p]
lib.xva_gelu_fwd.restype = i32
lib.xva_gelu_fwd.argtypes = [vp, vp, i64, vp]
lib.xva_gelu_bwd.restype = i32
lib.xva_gelu_bwd.argtypes = [vp, vp, vp, i64, vp]
lib.xva_ln_rows_fwd.restype = i32
lib.xva_ln_rows_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, f32, vp]
lib.xva_ln_rows_bwd.restype = i32
lib.xva_ln_rows_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp]
lib.xva_seq_mask.restype = i32
lib.xva_seq_mask.argtypes = [vp, i32, i32, i32, i32, i32, vp, vp]
lib.xva_fp_add_act.restype = i32
lib.xva_fp_add_act.argtypes = [vp, vp, i32, i64, vp]
lib.xva_hg_colsum.restype = i32
lib.xva_hg_colsum.argtypes = [vp, i32, vp, i64, i32, f32, vp]
lib.xva_rq_spline_inv.restype = i32
lib.xva_rq_spline_inv.argtypes = [vp, vp, vp, i64, i32, f32, f32, vp]
lib.xva_rq_spline_fwd.restype = i32
lib.xva_rq_spline_fwd.argtypes = [vp, vp, vp, vp, i64, i32, f32, f32, vp]
lib.xva_rq_spline_bwd.restype = i32
lib.xva_rq_spline_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, vp]
lib.xva_affine_fwd.restype = i32
lib.xva_affine_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
lib.xva_affine_bwd.restype = i32
lib.xva_affine_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
lib.xva_sdp_dequant_fwd.restype = i32
lib.xva_sdp_dequant_fwd.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
lib.xva_sdp_dequant_bwd.restype = i32
lib.xva_sdp_dequant_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, vp]
P = _lib.ptr
ST = _lib.stream_ptr
# 1 (default): a DilatedDepthSeparableConv stack is one autograd node (DDSStack); 0: the per-primitive composition (XVA_SDP_FUSED=0, A/B and tests)
_FUSED_DDS = int(__import__("os").environ.get("XVA_SDP_FUSED", "1"))



def _gbuf(p):
    """Where a backward accumulates parameter p's gradient: (buffer the kernel adds into, what autograd is handed).  A leaf parameter that
    already has a contiguous .grad — always, once train_step.FlatGroupAdamW has moved the gradients into its flat arena (zero_grad zeroes the
    arena, the .grad views stay) — takes the sum directly and autograd gets None: no zero-filled temporary, no add launch per parameter.
    Otherwise (first iteration, computed weights, torch.autograd.grad) a fresh zero tensor that autograd accumulates as usual."""
    g = p.grad if p.is_leaf else None
    if g is not None and g.is_contiguous() and g.dtype == torch.float32:
        return g, None
    z = torch.zeros_like(p, dtype=torch.float32, memory_format=torch.contiguous_format)
    return z, z


class DwConv(torch.autograd.Function):
    """y = depthwise dilated Conv1d(x * x_mask) ('same' zero padding); x (B, T, C), w (C, 1, k), b (C)."""
    @staticmethod
    def forward(ctx, x, w, b, lens, d):
        x = x.contiguous(); B, T, Cc = x.shape; k = w.size(-1)
        y = torch.empty_like(x)
        _lib.check(lib.xva_dwconv_fwd(P(x), P(w.contiguous()), P(b), P(y), P(lens), B, T, Cc, k, d, ST()), "xva_dwconv_fwd")
        ctx.save_for_backward(x, w, lens); ctx.d = d; ctx.params = (w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, lens = ctx.saved_tensors
        B, T, Cc = x.shape; k = w.size(-1)
        dx = torch.empty_like(x)
        (dw, rw), (db, rb) = _gbuf(ctx.params[0]), _gbuf(ctx.params[1])
        _lib.check(lib.xva_dwconv_bwd(P(dy.contiguous()), P(x), P(w.contiguous()), P(dx), P(dw), P(db), P(lens), B, T, Cc, k, ctx.d, ST()), "xva_dwconv_bwd")
        return dx, rw, rb, None, None


class Gelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous(); y = torch.empty_like(x)
        _lib.check(lib.xva_gelu_fwd(P(x), P(y), x.numel(), ST()), "xva_gelu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        _lib.check(lib.xva_gelu_bwd(P(x), P(dy.contiguous()), P(dx), x.numel(), ST()), "xva_gelu_bwd")
        return dx


class LayerNormRows(torch.autograd.Function):
    """LayerNorm2 (sdp.py:13-37): layer_norm over the last (channel) dimension of (B, T, C), eps 1e-5."""
    @staticmethod
    def forward(ctx, x, gamma, beta):
        x = x.contiguous(); rows, Cc = x.numel() // x.size(-1), x.size(-1)
        y = torch.empty_like(x); mean = torch.empty(rows, device=x.device); rstd = torch.empty(rows, device=x.device)
        _lib.check(lib.xva_ln_rows_fwd(P(x), P(gamma), P(beta), P(y), P(mean), P(rstd), rows, Cc, 1e-5, ST()), "xva_ln_rows_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd); ctx.params = (gamma, beta)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        rows, Cc = x.numel() // x.size(-1), x.size(-1)
        dx = torch.empty_like(x)
        (dg, rg), (db, rb) = _gbuf(ctx.params[0]), _gbuf(ctx.params[1])
        _lib.check(lib.xva_ln_rows_bwd(P(dy.contiguous()), P(x), P(mean), P(rstd), P(gamma), P(dx), P(dg), P(db), rows, Cc, ST()), "xva_ln_rows_bwd")
        return dx, rg, rb


_PREP = {}


def _prep(key, make):
    pg = _PREP.get(key)
    if pg is None:
        if len(_PREP) > 1024:
            _PREP.clear()
        pg = _PREP[key] = make()
    return pg


class Conv1x1(torch.autograd.Function):
    """nn.Conv1d(Cin, Cout, 1) on (B, T, Cin): one xva_gemm each for y, dx and dw (Cin, Cout multiples of 4); the three call sites are prepared
    once per (rows, Cin, Cout) (_lib.PreparedGemm)."""
    @staticmethod
    def forward(ctx, x, w, b):
        x = x.contiguous(); Cin = x.size(-1); rows = x.numel() // Cin; Cout = w.size(0)
        w2 = w.reshape(Cout, Cin).contiguous()
        y = torch.empty(*x.shape[:-1], Cout, device=x.device)
        _prep((0, rows, Cin, Cout, b is not None), lambda: _lib.PreparedGemm(x, w2, y, rows, Cout, Cin, Cin, Cin, Cout, layout=_lib.GEMM_NT, compute=0, bias=b)).run(x, w2, y, bias=b)
        ctx.save_for_backward(x, w2); ctx.wshape = tuple(w.shape); ctx.params = (w, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w2 = ctx.saved_tensors
        dy = dy.contiguous(); Cin = x.size(-1); rows = x.numel() // Cin; Cout = w2.size(0)
        dx = torch.empty_like(x)
        _prep((1, rows, Cin, Cout), lambda: _lib.PreparedGemm(dy, w2, dx, rows, Cin, Cout, Cout, Cin, Cin, layout=_lib.GEMM_NN, compute=0)).run(dy, w2, dx)
        (dw, rw), (db, rb) = _gbuf(ctx.params[0]), _gbuf(ctx.params[1])
        _prep((2, rows, Cin, Cout), lambda: _lib.PreparedGemm(dy, x, dw, Cout, Cin, rows, Cout, Cin, Cin, layout=_lib.GEMM_TN, compute=0, accumulate=True,
                                                             splitk=0)).run(dy, x, dw)
        _lib.check(lib.xva_hg_colsum(P(dy), 0, P(db), rows, Cout, 1.0, ST()), "xva_hg_colsum")
        return dx, (rw.view(ctx.wshape) if rw is not None else None), rb


class Mask(torch.autograd.Function):
    """x * x_mask on (B, T, C) (lens = the mask)"""
    @staticmethod
    def forward(ctx, x, lens):
        y = x.contiguous().clone(); B, T, Cc = y.shape
        _lib.check(lib.xva_seq_mask(P(y), 0, B, T, 0, Cc, P(lens), ST()), "xva_seq_mask")
        ctx.save_for_backward(lens)
        return y

    @staticmethod
    def backward(ctx, dy):
        (lens,) = ctx.saved_tensors
        d = dy.contiguous().clone(); B, T, Cc = d.shape
        _lib.check(lib.xva_seq_mask(P(d), 0, B, T, 0, Cc, P(lens), ST()), "xva_seq_mask")
        return d, None


class Dropout(torch.autograd.Function):
    """nn.Dropout on a contiguous tensor (sdp.py:90): y = x * (0 | 1 / (1 - p)) by the keyed hash of (seed, site, flat index)
    (csrc/xva_common.h xva_dropout_scale); the backward applies the same mask to the gradient."""
    @staticmethod
    def forward(ctx, x, p, seed, site):
        x = x.contiguous(); y = torch.empty_like(x)
        _lib.check(lib.xva_dropout_apply(P(x), P(y), 0, x.numel(), p, seed, site, ST()), "xva_dropout_apply")
        ctx.cfg = (p, seed, site)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous(); dx = torch.empty_like(dy)
        _lib.check(lib.xva_dropout_apply(P(dy), P(dx), 0, dy.numel(), *ctx.cfg, ST()), "xva_dropout_apply")
        return dx, None, None, None


class Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        y = a.contiguous().clone()
        _lib.check(lib.xva_fp_add_act(P(y), P(b.contiguous()), 0, y.numel(), ST()), "xva_fp_add_act")
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def _param(t, device):
    return t.to(device=device, dtype=torch.float32).requires_grad_(True)


class _DdsDims(C.Structure):
    """include/xva_hip.h xva_xvp_dds_dims"""
    _fields_ = [("B", C.c_int32), ("T", C.c_int32), ("C", C.c_int32), ("k", C.c_int32), ("L", C.c_int32), ("p_drop", C.c_float), ("seed", C.c_uint64),
                ("site0", C.c_uint32)]


lib.xva_xvp_dds_workspace_bytes.restype = C.c_int64
lib.xva_xvp_dds_workspace_bytes.argtypes = [C.POINTER(_DdsDims)]
lib.xva_xvp_dds_forward.restype = C.c_int32
lib.xva_xvp_dds_forward.argtypes = [C.POINTER(_DdsDims)] + [C.c_void_p] * 6 + [C.c_int64, C.c_void_p]
lib.xva_xvp_dds_backward.restype = C.c_int32
lib.xva_xvp_dds_backward.argtypes = [C.POINTER(_DdsDims)] + [C.c_void_p] * 6 + [C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
_DDS_ENGINE = int(__import__("os").environ.get("XVA_XVP_DDS_ENGINE", "1"))      # 0: the per-primitive sequencing below (same kernels; the A / B and the tests' reference)
_DDS_WS = {}
_CF_FUSED = int(__import__("os").environ.get("XVA_XVP_CF_FUSED", "1"))        # ConvFlow backward: proj gradients / the pre backward + d z as one launch each
lib.xva_small_wgrad.restype = C.c_int32
lib.xva_small_wgrad.argtypes = [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p]
lib.xva_cf_pre_bwd.restype = C.c_int32
lib.xva_cf_pre_bwd.argtypes = [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_void_p]
lib.xva_cf_pre_fwd.restype = C.c_int32
lib.xva_cf_pre_fwd.argtypes = [C.c_void_p] * 6 + [C.c_int64, C.c_int32, C.c_void_p]
lib.xva_cf_mask_slice.restype = lib.xva_cf_pad_mask.restype = C.c_int32
lib.xva_cf_mask_slice.argtypes = lib.xva_cf_pad_mask.argtypes = [C.c_void_p] * 2 + [C.c_int32] * 4 + [C.c_void_p, C.c_void_p]
lib.xva_cf_post_fwd.restype = C.c_int32
lib.xva_cf_post_fwd.argtypes = [C.c_void_p] * 5 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
lib.xva_cf_bwd_head.restype = C.c_int32
lib.xva_cf_bwd_head.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]


def _dds_engine_ok(x, params):
    return _DDS_ENGINE and x.size(-1) % 4 == 0 and 1 <= len(params) // 8 <= 16 and all(p.dtype == torch.float32 and p.is_contiguous() for p in params)


def _dds_fwd_engine(x, g, lens, cfg, params):
    k, L, p_drop, seed, site0 = cfg
    x = x.contiguous(); B, T, Cc = x.shape
    d = _DdsDims(B, T, Cc, k, L, p_drop, seed & 0xFFFFFFFFFFFFFFFF, site0)
    key = (B, T, Cc, k, L, p_drop > 0)
    n = _DDS_WS.get(key)
    if n is None:
        n = _DDS_WS[key] = int(lib.xva_xvp_dds_workspace_bytes(C.byref(d)))
        if n < 0:
            raise _lib.XvaError("xva_xvp_dds_workspace_bytes: %s" % lib.xva_last_error().decode())
    ws = torch.empty(n, dtype=torch.uint8, device=x.device)
    out = torch.empty_like(x)
    prm = (C.c_void_p * (8 * L))(*[p.data_ptr() for p in params])
    gc = g.contiguous() if g is not None else None
    _lib.check(lib.xva_xvp_dds_forward(C.byref(d), prm, P(x), P(gc), P(lens), P(out), P(ws), n, ST()), "xva_xvp_dds_forward")
    return out, ("engine", d, prm, ws, n, lens, params)


def _dds_bwd_engine(state, dy):
    _, d, prm, ws, n, lens, params = state
    bufs = [_gbuf(p) for p in params]
    grd = (C.c_void_p * len(params))(*[b.data_ptr() for b, _ in bufs])
    dx = torch.empty(d.B, d.T, d.C, device=dy.device)
    sk = _lib.sk_scratch(dy.device)
    _lib.check(lib.xva_xvp_dds_backward(C.byref(d), prm, grd, P(dy.contiguous()), P(lens), P(dx), P(ws), n, P(sk), sk.numel(), ST()), "xva_xvp_dds_backward")
    rets = [r for _, r in bufs]
    for i in range(2, len(rets), 8):                       # the 1x1 convolution's weight: handed back in the parameter's own shape
        if rets[i] is not None:
            rets[i] = rets[i].view(params[i].shape)
    return dx, rets


def _dds_fwd(x, g, lens, cfg, params):
    """The kernels of DilatedDepthSeparableConv.forward (sdp.py:70-93) back to back: (x [+ g]) -> L x {DwConv -> LayerNorm2 -> GELU -> Conv1x1 -> LayerNorm2 ->
    GELU -> [Dropout] -> + x} -> mask.  Residual adds and the mask run in place on tensors this call owns.  Returns (out, what _dds_bwd needs)."""
    if _dds_engine_ok(x, params):
        return _dds_fwd_engine(x, g, lens, cfg, params)
    k, L, p_drop, seed, site0 = cfg
    x = x.contiguous(); B, T, Cc = x.shape
    rows, n = B * T, x.numel()
    dev = x.device
    cur = torch.add(x, g) if g is not None else x
    saved = []
    for i in range(L):
        ws, bs, w1, b1, g1, be1, g2, be2 = params[8 * i:8 * i + 8]
        wsc = ws.contiguous()
        t1 = torch.empty_like(cur)
        _lib.check(lib.xva_dwconv_fwd(P(cur), P(wsc), P(bs), P(t1), P(lens), B, T, Cc, k, k ** i, ST()), "xva_dwconv_fwd")
        n1 = torch.empty_like(cur); m1 = torch.empty(rows, device=dev); r1 = torch.empty(rows, device=dev)
        _lib.check(lib.xva_ln_rows_fwd(P(t1), P(g1), P(be1), P(n1), P(m1), P(r1), rows, Cc, 1e-5, ST()), "xva_ln_rows_fwd")
        a1 = torch.empty_like(cur)
        _lib.check(lib.xva_gelu_fwd(P(n1), P(a1), n, ST()), "xva_gelu_fwd")
        w2 = w1.reshape(Cc, Cc).contiguous()
        t2 = torch.empty_like(cur)
        _prep((0, rows, Cc, Cc, True), lambda: _lib.PreparedGemm(a1, w2, t2, rows, Cc, Cc, Cc, Cc, Cc, layout=_lib.GEMM_NT, compute=0, bias=b1)).run(a1, w2, t2, bias=b1)
        n2 = torch.empty_like(cur); m2 = torch.empty(rows, device=dev); r2 = torch.empty(rows, device=dev)
        _lib.check(lib.xva_ln_rows_fwd(P(t2), P(g2), P(be2), P(n2), P(m2), P(r2), rows, Cc, 1e-5, ST()), "xva_ln_rows_fwd")
        a2 = torch.empty_like(cur)
        _lib.check(lib.xva_gelu_fwd(P(n2), P(a2), n, ST()), "xva_gelu_fwd")
        if p_drop > 0:
            nxt = torch.empty_like(cur)
            _lib.check(lib.xva_dropout_apply(P(a2), P(nxt), 0, n, p_drop, seed, site0 + i, ST()), "xva_dropout_apply")
        else:
            nxt = a2
        _lib.check(lib.xva_fp_add_act(P(nxt), P(cur), 0, n, ST()), "xva_fp_add_act")          # x = x + y, in place on the branch's own tensor
        saved.append((cur, wsc, t1, m1, r1, g1, n1, a1, w2, t2, m2, r2, g2, n2))
        cur = nxt
    if L == 0 and g is None:
        cur = cur.clone()
    _lib.check(lib.xva_seq_mask(P(cur), 0, B, T, 0, Cc, P(lens), ST()), "xva_seq_mask")
    return cur, (saved, lens, cfg, params, (B, T, Cc))


def _dds_bwd(state, dy):
    """-> (d x [= d g], [8 L parameter-gradient returns: None where the sum went straight into the parameter's .grad])"""
    if state[0] == "engine":
        return _dds_bwd_engine(state, dy)
    saved, lens, cfg, params, (B, T, Cc) = state
    k, L, p_drop, seed, site0 = cfg
    rows, n = B * T, B * T * Cc
    d = dy.contiguous().clone()
    _lib.check(lib.xva_seq_mask(P(d), 0, B, T, 0, Cc, P(lens), ST()), "xva_seq_mask")
    rets = [None] * (8 * L)
    for i in reversed(range(L)):
        cur, wsc, t1, m1, r1, g1, n1, a1, w2, t2, m2, r2, g2, n2 = saved[i]
        ws, bs, w1, b1, _, be1, _, be2 = params[8 * i:8 * i + 8]
        if p_drop > 0:
            da2 = torch.empty_like(d)
            _lib.check(lib.xva_dropout_apply(P(d), P(da2), 0, n, p_drop, seed, site0 + i, ST()), "xva_dropout_apply")
        else:
            da2 = d
        dn2 = torch.empty_like(d)
        _lib.check(lib.xva_gelu_bwd(P(n2), P(da2), P(dn2), n, ST()), "xva_gelu_bwd")
        dt2 = torch.empty_like(d)
        (dg2, rg2), (db2, rb2) = _gbuf(g2), _gbuf(be2)
        _lib.check(lib.xva_ln_rows_bwd(P(dn2), P(t2), P(m2), P(r2), P(g2), P(dt2), P(dg2), P(db2), rows, Cc, ST()), "xva_ln_rows_bwd")
        da1 = torch.empty_like(d)
        _prep((1, rows, Cc, Cc), lambda: _lib.PreparedGemm(dt2, w2, da1, rows, Cc, Cc, Cc, Cc, Cc, layout=_lib.GEMM_NN, compute=0)).run(dt2, w2, da1)
        (dw1, rw1), (dbb1, rb1) = _gbuf(w1), _gbuf(b1)
        _prep((2, rows, Cc, Cc), lambda: _lib.PreparedGemm(dt2, a1, dw1, Cc, Cc, rows, Cc, Cc, Cc, layout=_lib.GEMM_TN, compute=0, accumulate=True,
                                                            splitk=0)).run(dt2, a1, dw1)
        _lib.check(lib.xva_hg_colsum(P(dt2), 0, P(dbb1), rows, Cc, 1.0, ST()), "xva_hg_colsum")
        dn1 = torch.empty_like(d)
        _lib.check(lib.xva_gelu_bwd(P(n1), P(da1), P(dn1), n, ST()), "xva_gelu_bwd")
        dt1 = torch.empty_like(d)
        (dg1, rg1), (db1, rbe1) = _gbuf(g1), _gbuf(be1)
        _lib.check(lib.xva_ln_rows_bwd(P(dn1), P(t1), P(m1), P(r1), P(g1), P(dt1), P(dg1), P(db1), rows, Cc, ST()), "xva_ln_rows_bwd")
        dxb = torch.empty_like(d)
        (dws, rws), (dbs, rbs) = _gbuf(ws), _gbuf(bs)
        _lib.check(lib.xva_dwconv_bwd(P(dt1), P(cur), P(wsc), P(dxb), P(dws), P(dbs), P(lens), B, T, Cc, k, k ** i, ST()), "xva_dwconv_bwd")
        _lib.check(lib.xva_fp_add_act(P(dxb), P(d), 0, n, ST()), "xva_fp_add_act")             # d(x) = d(residual) + d(branch)
        d = dxb
        rets[8 * i:8 * i + 8] = [rws, rbs, (rw1.view(w1.shape) if rw1 is not None else None), rb1, rg1, rbe1, rg2, rb2]
    return d, rets


class DDSStack(torch.autograd.Function):
    """DilatedDepthSeparableConv.forward (sdp.py:70-93) as ONE autograd node: the same kernels in the same order as the per-primitive composition, issued
    back to back from one forward / one backward instead of through 8 L + 2 autograd nodes — the duration predictor runs ten of these stacks per
    iteration on (16 x 100 x 192) tensors and was bound by the host's per-node cost, not by the kernels.  Inputs: x (B, T, C), g (B, T,
    C) or None, lens, cfg = (k, L, p_drop, seed, site0), then 8 parameters per layer (convs_sep w, b; convs_1x1 w, b; norms_1 gamma, beta; norms_2
    gamma, beta)."""

    @staticmethod
    def forward(ctx, x, g, lens, cfg, *params):
        out, ctx.state = _dds_fwd(x, g, lens, cfg, params)
        ctx.has_g = g is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        d, rets = _dds_bwd(ctx.state, dy)
        return (d, d if ctx.has_g else None, None, None) + tuple(rets)


class ConvFlowFn(torch.autograd.Function):
    """ConvFlow.forward (sdp.py:147-176, the training direction) as ONE autograd node: x0 -> pre (Conv1d(1, H, 1): an outer product) -> the DDS stack conditioned
    on g -> proj (H -> 3K - 1, zero-padded to a multiple of 4 columns for the GEMM) -> mask -> rational-quadratic spline of x1 -> (stack, mask, per-item
    log-determinant).  Inputs: z (B, T, 2), g (B, T, H), lens, cfg = (dds cfg, K, bound, module), pre w, b, proj w, b, then the stack's 8 L parameters."""

    @staticmethod
    def forward(ctx, z, g, lens, cfg, pre_w, pre_b, proj_w, proj_b, *params):
        dds_cfg, K, bound, mod = cfg
        B, T, _ = z.shape
        H, NP = pre_b.numel(), 3 * K - 1
        NPp = (NP + 3) // 4 * 4
        rows = B * T
        fused = _CF_FUSED and H <= 256
        if fused:
            zc = z.contiguous()
            xs = torch.empty(2, B, T, device=z.device); x0, x1 = xs[0], xs[1]
            h = torch.empty(B, T, H, device=z.device)
            _lib.check(lib.xva_cf_pre_fwd(P(zc), P(pre_w.contiguous()), P(pre_b), P(x0), P(x1), P(h), rows, H, ST()), "xva_cf_pre_fwd")
        else:
            zt = z.permute(2, 0, 1).contiguous()                                   # (2, B, T): x0 = zt[0], x1 = zt[1], both contiguous
            x0, x1 = zt[0], zt[1]
            h = torch.addcmul(pre_b.view(1, 1, H), x0.unsqueeze(-1), pre_w.view(1, 1, H))
        h2, dds_state = _dds_fwd(h, g, lens, dds_cfg, params)
        wp, bp = mod._proj_pad(proj_w, proj_b, NP, NPp, H)                          # persistent zero-padded copies (two small copies, no fills / cats)
        hp = torch.empty(B, T, NPp, device=z.device)
        _prep((0, rows, H, NPp, True), lambda: _lib.PreparedGemm(h2, wp, hp, rows, NPp, H, H, H, NPp, layout=_lib.GEMM_NT, compute=0, bias=bp)).run(h2, wp, hp, bias=bp)
        if fused:
            hs = torch.empty(B, T, NP, device=z.device)
            _lib.check(lib.xva_cf_mask_slice(P(hp), P(hs), B, T, NPp, NP, P(lens), ST()), "xva_cf_mask_slice")
        else:
            _lib.check(lib.xva_seq_mask(P(hp), 0, B, T, 0, NPp, P(lens), ST()), "xva_seq_mask")
            hs = hp[..., :NP].contiguous()
        y1 = torch.empty_like(x1); ld = torch.empty_like(x1)
        _lib.check(lib.xva_rq_spline_fwd(P(x1), P(hs), P(y1), P(ld), x1.numel(), K, 1.0 / H ** 0.5, bound, ST()), "xva_rq_spline_fwd")
        ctx.state = (dds_state, x0, x1, h2, wp, hs, lens, (B, T, H, K, NP, NPp, bound), (pre_w, pre_b, proj_w, proj_b))
        if fused:
            out = torch.empty(B, T, 2, device=z.device); ldsum = torch.empty(B, device=z.device)
            _lib.check(lib.xva_cf_post_fwd(P(x0), P(y1), P(ld), P(out), P(ldsum), B, T, P(lens), ST()), "xva_cf_post_fwd")
            return out, ldsum
        out = torch.stack([x0, y1], -1)
        _lib.check(lib.xva_seq_mask(P(out), 0, B, T, 0, 2, P(lens), ST()), "xva_seq_mask")
        _lib.check(lib.xva_seq_mask(P(ld), 0, B, T, 0, 1, P(lens), ST()), "xva_seq_mask")
        return out, ld.sum(1)

    @staticmethod
    def backward(ctx, d_out, d_logdet):
        dds_state, x0, x1, h2, wp, hs, lens, (B, T, H, K, NP, NPp, bound), (pre_w, pre_b, proj_w, proj_b) = ctx.state
        rows = B * T
        dev = x0.device
        fused = _CF_FUSED and H <= 256
        if fused:
            dm = torch.empty(2, B, T, device=dev); d_ld = torch.empty(B, T, device=dev)
            _lib.check(lib.xva_cf_bwd_head(P(d_out.contiguous()), P(d_logdet.contiguous().float()), P(dm), P(d_ld), B, T, P(lens), ST()), "xva_cf_bwd_head")
        else:
            dm = d_out.permute(2, 0, 1).contiguous()                                # (2, B, T)
            _lib.check(lib.xva_seq_mask(P(dm), 0, 2 * B, T, 0, 1, P(torch.cat([lens, lens])), ST()), "xva_seq_mask")
            d_ld = d_logdet.reshape(B, 1).expand(B, T).contiguous()
            _lib.check(lib.xva_seq_mask(P(d_ld), 0, B, T, 0, 1, P(lens), ST()), "xva_seq_mask")
        dx1 = torch.empty_like(x1); dhs = torch.empty_like(hs)
        _lib.check(lib.xva_rq_spline_bwd(P(x1), P(hs), P(dm[1]), P(d_ld), P(dx1), P(dhs), x1.numel(), K, 1.0 / H ** 0.5, bound, ST()), "xva_rq_spline_bwd")
        if fused:
            dhp = torch.empty(B, T, NPp, device=dev)
            _lib.check(lib.xva_cf_pad_mask(P(dhs), P(dhp), B, T, NPp, NP, P(lens), ST()), "xva_cf_pad_mask")
        else:
            dhp = torch.nn.functional.pad(dhs, (0, NPp - NP))
            _lib.check(lib.xva_seq_mask(P(dhp), 0, B, T, 0, NPp, P(lens), ST()), "xva_seq_mask")
        dh2 = torch.empty(B, T, H, device=dev)
        _prep((1, rows, H, NPp), lambda: _lib.PreparedGemm(dhp, wp, dh2, rows, H, NPp, NPp, H, H, layout=_lib.GEMM_NN, compute=0)).run(dhp, wp, dh2)
        if fused:
            wb = torch.zeros(NPp * H + NPp, device=dev)                             # d(proj weight | bias), one fill
            dwp, dbp = wb[:NPp * H].view(NPp, H), wb[NPp * H:]
            _lib.check(lib.xva_small_wgrad(P(dhp), P(h2), P(dwp), P(dbp), rows, NPp, H, ST()), "xva_small_wgrad")
            dh, rets = _dds_bwd(dds_state, dh2)                                     # d(pre output) = d(conditioning g)
            dh = dh.contiguous()
            (gw, rw), (gb, rb) = _gbuf(pre_w), _gbuf(pre_b)
            dz = torch.empty(B, T, 2, device=dev)
            _lib.check(lib.xva_cf_pre_bwd(P(dh), P(pre_w.contiguous()), P(x0), P(dm), P(dx1), P(dz), P(gw), P(gb), rows, H, ST()), "xva_cf_pre_bwd")   # dm = [d x0 | d x1'] : its first half
            return (dz, dh, None, None, (rw.view(pre_w.shape) if rw is not None else None), rb, dwp[:NP].reshape(proj_w.shape), dbp[:NP]) + tuple(rets)
        dwp = torch.zeros(NPp, H, device=dev)
        _prep((2, rows, H, NPp), lambda: _lib.PreparedGemm(dhp, h2, dwp, NPp, H, rows, NPp, H, H, layout=_lib.GEMM_TN, compute=0, accumulate=True,
                                                           splitk=0)).run(dhp, h2, dwp)
        dbp = torch.zeros(NPp, device=dev)
        _lib.check(lib.xva_hg_colsum(P(dhp), 0, P(dbp), rows, NPp, 1.0, ST()), "xva_hg_colsum")
        dh, rets = _dds_bwd(dds_state, dh2)                                         # d(pre output) = d(conditioning g)
        d_x0 = (dh * pre_w.view(1, 1, H)).sum(-1) + dm[0]
        d_pre_w = (dh * x0.unsqueeze(-1)).sum((0, 1)).view(pre_w.shape)
        d_pre_b = dh.sum((0, 1))
        dz = torch.stack([d_x0, dx1], -1)
        return (dz, dh, None, None, d_pre_w, d_pre_b, dwp[:NP].reshape(proj_w.shape), dbp[:NP]) + tuple(rets)


class DilatedDepthSeparableConv:
    """sdp.py:40-93: per layer  y = GELU(LN(dwconv_{d = k^i}(x * x_mask))); y = GELU(LN(conv1x1(y))); x = x + y;  output x * x_mask."""

    def __init__(self, channels, kernel_size, num_layers, dropout_p=0.0, device="cuda", seed=0, dropout_site_base=0):
        """dropout_p > 0: nn.Dropout after the second GELU of every layer (sdp.py:90), site dropout_site_base + layer, flat index of the (B, T, C)
        activation, under the seed of `drop_seed` (set per iteration by the owner); `training = False` switches it off."""
        if not 0.0 <= dropout_p < 1.0:
            raise ValueError("DilatedDepthSeparableConv: dropout_p must be in [0, 1)")
        self.dropout_p, self.site0, self.training, self.drop_seed = float(dropout_p), int(dropout_site_base), True, int(seed) + 0x5EED
        if channels % 4 or kernel_size % 2 != 1 or kernel_size > 7:
            raise NotImplementedError("DilatedDepthSeparableConv: channels must be a multiple of 4, kernel_size odd <= 7")
        self.C, self.k, self.L = channels, kernel_size, num_layers
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(seed)
        u = lambda shape, fan: (torch.rand(*shape, generator=gen) * 2 - 1) * (1.0 / fan) ** 0.5
        self.p = {}
        for i in range(num_layers):
            self.p["convs_sep.%d.weight" % i] = _param(u((channels, 1, kernel_size), kernel_size), self.device)
            self.p["convs_sep.%d.bias" % i] = _param(u((channels,), kernel_size), self.device)
            self.p["convs_1x1.%d.weight" % i] = _param(u((channels, channels, 1), channels), self.device)
            self.p["convs_1x1.%d.bias" % i] = _param(u((channels,), channels), self.device)
            for n in ("norms_1", "norms_2"):
                self.p["%s.%d.gamma" % (n, i)] = _param(torch.ones(channels), self.device)
                self.p["%s.%d.beta" % (n, i)] = _param(torch.zeros(channels), self.device)

    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.p.items()}

    def load_state_dict(self, sd):
        if set(sd) != set(self.p):
            raise KeyError("DilatedDepthSeparableConv.load_state_dict: key mismatch %s" % sorted(set(sd) ^ set(self.p))[:6])
        for k, t in self.p.items():
            if tuple(t.shape) != tuple(sd[k].shape):
                raise ValueError("%s: shape %s != %s" % (k, tuple(sd[k].shape), tuple(t.shape)))
            with torch.no_grad():
                t.copy_(sd[k])

    def parameters(self):
        return list(self.p.values())

    def forward_btc(self, x, lens, g=None):
        """x, g: (B, T, C)"""
        p = self.p
        if _FUSED_DDS:
            ps = []
            for i in range(self.L):
                ps += [p["convs_sep.%d.weight" % i], p["convs_sep.%d.bias" % i], p["convs_1x1.%d.weight" % i], p["convs_1x1.%d.bias" % i],
                       p["norms_1.%d.gamma" % i], p["norms_1.%d.beta" % i], p["norms_2.%d.gamma" % i], p["norms_2.%d.beta" % i]]
            drop = self.dropout_p if self.training else 0.0
            return DDSStack.apply(x, g, lens, (self.k, self.L, float(drop), int(self.drop_seed), int(self.site0)), *ps)
        if g is not None:
            x = Add.apply(x, g)
        for i in range(self.L):
            y = DwConv.apply(x, p["convs_sep.%d.weight" % i], p["convs_sep.%d.bias" % i], lens, self.k ** i)
            y = Gelu.apply(LayerNormRows.apply(y, p["norms_1.%d.gamma" % i], p["norms_1.%d.beta" % i]))
            y = Conv1x1.apply(y, p["convs_1x1.%d.weight" % i], p["convs_1x1.%d.bias" % i])
            y = Gelu.apply(LayerNormRows.apply(y, p["norms_2.%d.gamma" % i], p["norms_2.%d.beta" % i]))
            if self.dropout_p > 0 and self.training:
                y = Dropout.apply(y, self.dropout_p, self.drop_seed, self.site0 + i)
            x = Add.apply(x, y)
        return Mask.apply(x, lens)

    def __call__(self, x, x_mask, g=None):
        """(B, C, T) in and out, like the reference"""
        _lib.require_cuda(x)
        lens = _lens_of(x, x_mask)
        y = self.forward_btc(x.float().transpose(1, 2).contiguous(), lens, g.float().transpose(1, 2).contiguous() if g is not None else None)
        return y.transpose(1, 2)


class RqSpline(torch.autograd.Function):
    """x (...), h (..., 3K - 1) -> y, log|det| : the forward rational-quadratic transform with linear tails (util.py:203-391)."""
    @staticmethod
    def forward(ctx, x, h, K, wh_scale, bound):
        x = x.contiguous(); h = h.contiguous()
        y = torch.empty_like(x); ld = torch.empty_like(x)
        _lib.check(lib.xva_rq_spline_fwd(P(x), P(h), P(y), P(ld), x.numel(), K, wh_scale, bound, ST()), "xva_rq_spline_fwd")
        ctx.save_for_backward(x, h); ctx.cfg = (K, wh_scale, bound)
        return y, ld

    @staticmethod
    def backward(ctx, dy, dld):
        x, h = ctx.saved_tensors
        K, wh_scale, bound = ctx.cfg
        dx = torch.empty_like(x); dh = torch.empty_like(h)
        _lib.check(lib.xva_rq_spline_bwd(P(x), P(h), P(dy.contiguous()), P(dld.contiguous()), P(dx), P(dh), x.numel(), K, wh_scale, bound, ST()), "xva_rq_spline_bwd")
        return dx, dh, None, None, None


class _Module:
    def state_dict(self):
        return {k: v.detach().clone() for k, v in self.p.items()}

    def load_state_dict(self, sd):
        if set(sd) != set(self.p):
            raise KeyError("%s.load_state_dict: key mismatch %s" % (type(self).__name__, sorted(set(sd) ^ set(self.p))[:6]))
        for k, t in self.p.items():
            if tuple(t.shape) != tuple(sd[k].shape):
                raise ValueError("%s: shape %s != %s" % (k, tuple(sd[k].shape), tuple(t.shape)))
            with torch.no_grad():
                t.copy_(sd[k])

    def parameters(self):
        return list(self.p.values())


class ConvFlow(_Module):
    """sdp.py:116-176 (in_channels = 2, the predictor's case; forward direction): x0 conditions a spline over x1.
    h = proj(DDSConv(pre(x0), g)) * x_mask -> [10 widths | 10 heights | 9 derivatives] per token; y1, log|det| = spline(x1, h)."""

    def __init__(self, in_channels, hidden_channels, kernel_size, num_layers, num_bins=10, tail_bound=5.0, device="cuda", seed=0):
        if in_channels != 2:
            raise NotImplementedError("ConvFlow: built for the 2-channel flow variable of the duration predictor")
        self.H, self.K, self.bound = hidden_channels, num_bins, float(tail_bound)
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(seed)
        self.convs = DilatedDepthSeparableConv(hidden_channels, kernel_size, num_layers, device=device, seed=seed + 1)
        NP = 3 * num_bins - 1
        u = lambda shape, fan: (torch.rand(*shape, generator=gen) * 2 - 1) * (1.0 / fan) ** 0.5
        self.p = {"pre.weight": _param(u((hidden_channels, 1, 1), 1), self.device), "pre.bias": _param(u((hidden_channels,), 1), self.device),
                  "proj.weight": _param(torch.zeros(NP, hidden_channels, 1), self.device), "proj.bias": _param(torch.zeros(NP), self.device)}
        for k, v in self.convs.p.items():
            self.p["convs." + k] = v

    def _proj_pad(self, proj_w, proj_b, NP, NPp, H):
        """proj's weight / bias zero-padded to NPp output columns in buffers that live with the module (the pad rows are written once)"""
        buf = getattr(self, "_projp", None)
        if buf is None or buf[0].device != proj_w.device:
            buf = self._projp = (torch.zeros(NPp, H, device=proj_w.device), torch.zeros(NPp, device=proj_w.device))
        with torch.no_grad():
            buf[0][:NP].copy_(proj_w.detach().reshape(NP, H))
            buf[1][:NP].copy_(proj_b.detach())
        return buf

    def forward_btc(self, z, lens, g=None, reverse=False):
        """z (B, T, 2), g (B, T, H) -> (B, T, 2), logdet (B) ; reverse=True: the inverse map, (B, T, 2) only"""
        p, H, NP = self.p, self.H, 3 * self.K - 1
        if _FUSED_DDS and not reverse and g is not None:
            c = self.convs
            ps = []
            for i in range(c.L):
                ps += [c.p["convs_sep.%d.weight" % i], c.p["convs_sep.%d.bias" % i], c.p["convs_1x1.%d.weight" % i], c.p["convs_1x1.%d.bias" % i],
                       c.p["norms_1.%d.gamma" % i], c.p["norms_1.%d.beta" % i], c.p["norms_2.%d.gamma" % i], c.p["norms_2.%d.beta" % i]]
            drop = c.dropout_p if c.training else 0.0
            cfg = ((c.k, c.L, float(drop), int(c.drop_seed), int(c.site0)), self.K, self.bound, self)
            return ConvFlowFn.apply(z, g, lens, cfg, p["pre.weight"], p["pre.bias"], p["proj.weight"], p["proj.bias"], *ps)
        x0, x1 = z[..., 0:1], z[..., 1]
        # pre: Conv1d(1, H, 1) and proj: Conv1d(H, 3K - 1, 1) ride in GEMMs whose narrow dimension is zero-padded to a multiple of 4
        x0p = torch.cat([x0, torch.zeros(*x0.shape[:-1], 3, device=z.device)], -1)
        wpre = torch.cat([p["pre.weight"].reshape(H, 1), torch.zeros(H, 3, device=z.device)], 1).reshape(H, 4, 1)
        h = Conv1x1.apply(x0p, wpre, p["pre.bias"])
        h = self.convs.forward_btc(h, lens, g)
        NPp = (NP + 3) // 4 * 4
        wproj = torch.cat([p["proj.weight"].reshape(NP, H), torch.zeros(NPp - NP, H, device=z.device)], 0).reshape(NPp, H, 1)
        bproj = torch.cat([p["proj.bias"], torch.zeros(NPp - NP, device=z.device)])
        h = Mask.apply(Conv1x1.apply(h, wproj, bproj), lens)[..., :NP]
        if reverse:                                                  # sdp.py:158-176 with inverse=True: x1 = spline^-1(y1); no log|det| (sampling path)
            hc = h.contiguous(); y1 = x1.contiguous(); x1n = torch.empty_like(y1)
            _lib.check(lib.xva_rq_spline_inv(P(y1), P(hc), P(x1n), y1.numel(), self.K, 1.0 / H ** 0.5, self.bound, ST()), "xva_rq_spline_inv")
            return Mask.apply(torch.stack([x0[..., 0], x1n], -1), lens)
        y1, ld = RqSpline.apply(x1, h, self.K, 1.0 / H ** 0.5, self.bound)
        out = Mask.apply(torch.stack([x0[..., 0], y1], -1), lens)
        ldm = Mask.apply(ld.unsqueeze(-1), lens)
        return out, ldm.sum((1, 2))

    def __call__(self, x, x_mask, g=None):
        _lib.require_cuda(x)
        lens = _lens_of(x, x_mask)
        y, logdet = self.forward_btc(x.float().transpose(1, 2).contiguous(), lens, g.float().transpose(1, 2).contiguous() if g is not None else None)
        return y.transpose(1, 2), logdet


class Affine(torch.autograd.Function):
    """ElementwiseAffine.forward (sdp.py:107-111) on (B, T, C): y, logdet (B)"""
    @staticmethod
    def forward(ctx, x, log_scale, translation, lens):
        x = x.contiguous(); B, T, Cc = x.shape
        ls, tr = log_scale.reshape(-1).contiguous(), translation.reshape(-1).contiguous()
        y = torch.empty_like(x); ld = torch.empty(B, device=x.device)
        _lib.check(lib.xva_affine_fwd(P(x), P(ls), P(tr), P(y), P(ld), P(lens), B, T, Cc, ST()), "xva_affine_fwd")
        ctx.save_for_backward(x, ls, lens); ctx.shapes = (tuple(log_scale.shape), tuple(translation.shape))
        return y, ld

    @staticmethod
    def backward(ctx, dy, dld):
        x, ls, lens = ctx.saved_tensors
        B, T, Cc = x.shape
        dx = torch.empty_like(x); dls = torch.zeros_like(ls); dtr = torch.zeros_like(ls)
        _lib.check(lib.xva_affine_bwd(P(x), P(ls), P(dy.contiguous()), P(dld.contiguous()), P(dx), P(dls), P(dtr), P(lens), B, T, Cc, ST()), "xva_affine_bwd")
        return dx, dls.view(ctx.shapes[0]), dtr.view(ctx.shapes[1]), None


class Dequant(torch.autograd.Function):
    """z_u, dr (B, T) -> log(max(dr - sigmoid(z_u), 1e-5)) * mask, (logsigmoid(z_u) + logsigmoid(-z_u)) * mask   (sdp.py:283-296)"""
    @staticmethod
    def forward(ctx, zu, dr, lens):
        zu = zu.contiguous(); dr = dr.contiguous(); B, T = zu.shape
        z0 = torch.empty_like(zu); ls = torch.empty_like(zu)
        _lib.check(lib.xva_sdp_dequant_fwd(P(zu), P(dr), P(z0), P(ls), P(lens), B, T, ST()), "xva_sdp_dequant_fwd")
        ctx.save_for_backward(zu, dr, lens)
        return z0, ls

    @staticmethod
    def backward(ctx, dz0, dls):
        zu, dr, lens = ctx.saved_tensors
        B, T = zu.shape
        d = torch.empty_like(zu)
        _lib.check(lib.xva_sdp_dequant_bwd(P(zu), P(dr), P(dz0.contiguous()), P(dls.contiguous()), P(d), P(lens), B, T, ST()), "xva_sdp_dequant_bwd")
        return d, None, None


class ElementwiseAffine(_Module):
    def __init__(self, channels, device="cuda"):
        self.p = {"translation": _param(torch.zeros(channels, 1), device), "log_scale": _param(torch.zeros(channels, 1), device)}

    def forward_btc(self, x, lens, g=None, reverse=False):
        if reverse:                                                  # sdp.py:112-113: (x - translation) * exp(-log_scale) * mask
            return Mask.apply((x - self.p["translation"].reshape(1, 1, -1)) * torch.exp(-self.p["log_scale"].reshape(1, 1, -1)), lens)
        return Affine.apply(x, self.p["log_scale"], self.p["translation"], lens)


class StochasticDurationPredictor(_Module):
    """sdp.py:179-310, training direction: the negative log-likelihood (B,) of the durations dr under the flow, with variational dequantisation
    (posterior flows conditioned on text + duration encodings).  `noise` (B, 2, T): the N(0, 1) draw of :281 (drawn with torch when None)."""

    def __init__(self, in_channels, hidden_channels, kernel_size, dropout_p, num_flows=4, cond_channels=0, language_emb_dim=0, device="cuda", seed=0,
                 dropout_site_base=0):
        """dropout_p: nn.Dropout inside `convs` and `post_convs` (sdp.py:227,237; the flows' own DilatedDepthSeparableConv have none, :144) — sites
        dropout_site_base + {0, 1, 2} and + {3, 4, 5}."""
        if language_emb_dim:
            in_channels += language_emb_dim
        if in_channels % 4 or hidden_channels % 4 or (cond_channels or 0) % 4 or (language_emb_dim or 0) % 4:
            raise NotImplementedError("StochasticDurationPredictor: channel counts must be multiples of 4")
        self.H = hidden_channels
        self.device = torch.device(device)
        gen = torch.Generator().manual_seed(seed)
        u = lambda shape, fan: (torch.rand(*shape, generator=gen) * 2 - 1) * (1.0 / fan) ** 0.5
        H = hidden_channels
        self.p = {}

        def conv(name, co, ci):
            self.p[name + ".weight"] = _param(u((co, ci, 1), ci), self.device)
            self.p[name + ".bias"] = _param(u((co,), ci), self.device)

        def sub(name, m):
            for k, v in m.p.items():
                self.p[name + "." + k] = v
            return m
        conv("pre", H, in_channels)
        self.convs = sub("convs", DilatedDepthSeparableConv(H, kernel_size, 3, dropout_p=dropout_p, device=device, seed=seed + 1, dropout_site_base=dropout_site_base))
        conv("proj", H, H)
        self.flows = [sub("flows.0", ElementwiseAffine(2, device))] + [sub("flows.%d" % (i + 1), ConvFlow(2, H, kernel_size, 3, device=device, seed=seed + 10 + i))
                                                                       for i in range(num_flows)]
        conv("post_pre", H, 1)
        self.post_convs = sub("post_convs", DilatedDepthSeparableConv(H, kernel_size, 3, dropout_p=dropout_p, device=device, seed=seed + 2,
                                                                       dropout_site_base=dropout_site_base + 3))
        conv("post_proj", H, H)
        self.post_flows = [sub("post_flows.0", ElementwiseAffine(2, device))] + [sub("post_flows.%d" % (i + 1), ConvFlow(2, H, kernel_size, 3, device=device,
                                                                                                                      seed=seed + 20 + i)) for i in range(num_flows)]
        self.has_cond = bool(cond_channels)
        if self.has_cond:
            conv("cond", H, cond_channels)
        self.has_lang = bool(language_emb_dim)
        if self.has_lang:
            conv("cond_lang", H, language_emb_dim)

    def set_dropout_seed(self, seed):
        self.convs.drop_seed = self.post_convs.drop_seed = int(seed) & 0xFFFFFFFFFFFFFFFF

    def train(self, mode=True):
        self.convs.training = self.post_convs.training = bool(mode)
        return self

    def eval(self):
        return self.train(False)

    def _text_condition(self, x, lens, g, lang_emb):
        """sdp.py:260-275: proj(convs(pre(x) + cond(g) + cond_lang(lang_emb))) * mask, time-major (B, T, H)"""
        p, H = self.p, self.H
        B, _, T = x.shape
        tm = lambda t: t.float().transpose(1, 2).contiguous()
        xs = Conv1x1.apply(tm(x), p["pre.weight"], p["pre.bias"])
        if g is not None:
            xs = Add.apply(xs, Conv1x1.apply(tm(g), p["cond.weight"], p["cond.bias"]).expand(B, T, H).contiguous())
        if lang_emb is not None:
            xs = Add.apply(xs, Conv1x1.apply(tm(lang_emb), p["cond_lang.weight"], p["cond_lang.bias"]).expand(B, T, H).contiguous())
        xs = self.convs.forward_btc(xs, lens)
        return Mask.apply(Conv1x1.apply(xs, p["proj.weight"], p["proj.bias"]), lens)

    @torch.no_grad()
    def infer(self, x, x_mask, g=None, lang_emb=None, noise_scale=1.0, noise=None):
        """The sampling direction, `forward(..., reverse=True)` (sdp.py:311-321): z = noise * noise_scale runs BACKWARDS through the flows — the
        list reversed, the second ConvFlow from the end of that list dropped ("a useless vflow"), a channel flip before every flow — and the
        first channel is log w.  x (B, C, T), x_mask (B, 1, T), noise (B, 2, T) N(0, 1) (drawn with torch when None) -> logw (B, 1, T).
        The caller runs it in eval mode (no dropout in `convs`)."""
        _lib.require_cuda(x)
        lens = _lens_of(x, x_mask)
        B, _, T = x.shape
        xs = self._text_condition(x, lens, g, lang_emb)
        if noise is None:
            noise = torch.randn(B, 2, T, device=x.device)
        z = (noise.float() * noise_scale).transpose(1, 2).contiguous()
        flows = list(reversed(self.flows))
        flows = flows[:-2] + [flows[-1]]
        for flow in flows:
            z = torch.flip(z, [2]).contiguous()
            z = flow.forward_btc(z, lens, xs, reverse=True)
        return z[..., 0:1].transpose(1, 2).contiguous()

    def __call__(self, x, x_mask, dr, g=None, lang_emb=None, noise=None):
        """x (B, C, T), x_mask (B, 1, T), dr (B, 1, T), g (B, Cg, 1), lang_emb (B, Cl, 1 or T), noise (B, 2, T) -> nll (B,)"""
        _lib.require_cuda(x, dr)
        import math
        p, H = self.p, self.H
        lens = _lens_of(x, x_mask)
        B, _, T = x.shape
        tm = lambda t: t.float().transpose(1, 2).contiguous()
        xs = self._text_condition(x, lens, g, lang_emb)
        # condition encoder of the durations: Conv1d(1, H, 1) as a GEMM over a 4-wide zero-padded input
        drs = tm(dr)
        wpp = torch.cat([p["post_pre.weight"].reshape(H, 1), torch.zeros(H, 3, device=x.device)], 1).reshape(H, 4, 1)
        h = Conv1x1.apply(torch.cat([drs, torch.zeros(B, T, 3, device=x.device)], -1), wpp, p["post_pre.bias"])
        h = self.post_convs.forward_btc(h, lens)
        h = Mask.apply(Conv1x1.apply(h, p["post_proj.weight"], p["post_proj.bias"]), lens)
        if noise is None:
            noise = torch.randn(B, 2, T, device=x.device)
        nz = Mask.apply(tm(noise), lens)
        cond_q = Add.apply(xs, h)
        z_q, ld_q = nz, 0.0
        for idx, flow in enumerate(self.post_flows):
            z_q, ld = flow.forward_btc(z_q, lens, cond_q)
            ld_q = ld_q + ld
            if idx > 0:
                z_q = torch.flip(z_q, [2])
        z_u, z_v = z_q[..., 0], z_q[..., 1]
        z0, lsig = Dequant.apply(z_u, drs[..., 0], lens)
        ld_q = ld_q + lsig.sum(1)
        nll_post = (-0.5 * (math.log(2 * math.pi) * 2 * lens.float() + (nz ** 2).sum((1, 2)))) - ld_q
        ld_tot = -z0.sum(1)
        z = torch.stack([z0, z_v], -1)
        for idx, flow in enumerate(self.flows):
            z, ld = flow.forward_btc(z, lens, xs)
            ld_tot = ld_tot + ld
            if idx > 0:
                z = torch.flip(z, [2])
        nll_flow = 0.5 * (math.log(2 * math.pi) * 2 * lens.float() + (z ** 2).sum((1, 2))) - ld_tot
        return nll_flow + nll_post
