"""GPU parity of the fused attention kernels (csrc/attention.hip) against an fp64 torch restatement of MultiHeadAttn
(python/fastpitch1_1/fastpitch/transformer.py:109-130, n_head = 1, d_head = 64) on the same bf16-rounded inputs, without and
with dropout (masks = the oracle's HashDropout, i.e. the same stateless function the kernels use)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup():
    from xva_trainer_amd import _lib
    lib = _lib.lib
    vp, i32, f32 = C.c_void_p, C.c_int32, C.c_float
    lib.xva_fp_attention_fwd.restype = i32
    lib.xva_fp_attention_fwd.argtypes = [vp, vp, vp, vp, i32, i32, f32, f32, C.c_uint64, C.c_uint32, vp]
    lib.xva_fp_attention_bwd.restype = i32
    lib.xva_fp_attention_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, C.c_uint64, C.c_uint32, vp]
    i64 = C.c_int64
    lib.xva_fp_attention_fwd_pairs.restype = i32
    lib.xva_fp_attention_fwd_pairs.argtypes = [vp, i64, vp, vp, i64, vp, i32, i32, f32, f32, C.c_uint64, C.c_uint32, vp]
    lib.xva_fp_attention_bwd_pairs.restype = i32
    lib.xva_fp_attention_bwd_pairs.argtypes = [vp, i64, vp, i64, vp, i64, vp, vp, vp, vp, i64, i32, i32, f32, f32, C.c_uint64, C.c_uint32, vp]
    return _lib, lib


def _rel(a, b):
    return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("B,T,lens", [(3, 150, [150, 37, 90]), (2, 700, [700, 333]), (4, 62, [62, 1, 17, 40])])
def test_fused_attention_fwd_bwd(B, T, lens, p_drop):
    from oracle import fastpitch as ofp
    L, lib = _setup()
    torch.manual_seed(B * 1000 + T)
    Tp = T + 2
    seed, stream = 4242, 7
    qkv = torch.zeros(B, Tp, 192, device="cuda", dtype=torch.bfloat16)
    qkv[:, 1:T + 1] = (torch.randn(B, T, 192, device="cuda") * 1.5).bfloat16()
    lens_t = torch.tensor(lens, device="cuda", dtype=torch.int32)
    d_av = torch.zeros(B, Tp, 64, device="cuda", dtype=torch.bfloat16)
    live = (torch.arange(T, device="cuda")[None, :] < lens_t[:, None])
    d_av[:, 1:T + 1] = (torch.randn(B, T, 64, device="cuda") * live[..., None]).bfloat16()   # dead rows carry no gradient (LN mask)
    av = torch.full((B, Tp, 64), 9.0, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B, Tp, device="cuda")
    d_qkv = torch.full((B, Tp, 192), 9.0, device="cuda", dtype=torch.bfloat16)
    dscr = torch.zeros(B, Tp, device="cuda")
    st = L.stream_ptr()
    L.check(lib.xva_fp_attention_fwd(L.ptr(qkv), L.ptr(lens_t), L.ptr(av), L.ptr(lse), B, Tp, 0.125, p_drop, seed, stream, st))
    L.check(lib.xva_fp_attention_bwd(L.ptr(qkv), L.ptr(av), L.ptr(d_av), L.ptr(lse), L.ptr(dscr), L.ptr(lens_t), L.ptr(d_qkv), B, Tp, 0.125,
                                     p_drop, seed, stream, st))
    torch.cuda.synchronize()
    # reference (unpadded coordinates), fp64, same masks
    x = qkv[:, 1:T + 1].double().cpu().requires_grad_(True)
    q, k, v = x[..., :64], x[..., 64:128], x[..., 128:]
    score = torch.bmm(q, k.transpose(1, 2)) * 0.125
    kmask = ~(torch.arange(T)[None, :] < torch.tensor(lens)[:, None])
    score = score.masked_fill(kmask[:, None, :], -float("inf"))
    prob = torch.softmax(score, dim=2)
    if p_drop > 0:
        prob = ofp.HashDropout(p_drop, seed).prob(stream, prob)
    out = torch.bmm(prob, v)
    out.backward(d_av[:, 1:T + 1].double().cpu())
    lv = live.cpu()
    assert _rel(av[:, 1:T + 1].cpu()[lv], out.detach()[lv]) < 1.5e-2
    ref_lse = torch.logsumexp(score.detach(), dim=2)
    assert (lse[:, 1:T + 1].cpu().double() - ref_lse)[lv].abs().max().item() < 2e-3
    g = d_qkv[:, 1:T + 1].cpu()
    for name, sl in (("dQ", slice(0, 64)), ("dK", slice(64, 128)), ("dV", slice(128, 192))):
        assert _rel(g[..., sl], x.grad[..., sl]) < 2.5e-2, name
    # structural rows and dead keys get exactly zero gradient
    assert d_qkv[:, 0].abs().max().item() == 0.0 and d_qkv[:, Tp - 1].abs().max().item() == 0.0
    dead = ~lv
    assert g[dead].abs().max().item() == 0.0 if dead.any() else True


def _pair(x):
    """fp32 tensor -> (2, ...) bf16: hi = bf16(x), lo = bf16(x - hi) (include/xva_gemm.h `planes`)"""
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return torch.stack([hi, lo]).contiguous()


def _val(p):
    return p[0].double() + p[1].double()


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("B,T,lens", [(3, 150, [150, 37, 90]), (2, 700, [700, 333]), (4, 62, [62, 1, 17, 40])])
def test_fused_attention_on_split_pairs(B, T, lens, p_drop):
    """The flash-style kernels of the split-products mode: qkv / d(av) given as split-bf16 pairs, every product hi.hi + hi.lo + lo.hi, outputs as pairs —
    against the fp64 restatement on the pairs' values: ~16 mantissa bits per operand, i.e. 1e-5, not the bf16 kernels' 1e-2."""
    from oracle import fastpitch as ofp
    L, lib = _setup()
    torch.manual_seed(B * 1000 + T + 1)
    Tp = T + 2
    seed, stream = 4243, 11
    x32 = torch.zeros(B, Tp, 192, device="cuda")
    x32[:, 1:T + 1] = torch.randn(B, T, 192, device="cuda") * 1.5
    qkv = _pair(x32)
    lens_t = torch.tensor(lens, device="cuda", dtype=torch.int32)
    live = (torch.arange(T, device="cuda")[None, :] < lens_t[:, None])
    g32 = torch.zeros(B, Tp, 64, device="cuda")
    g32[:, 1:T + 1] = torch.randn(B, T, 64, device="cuda") * live[..., None]
    d_av = _pair(g32)
    av = torch.full((2, B, Tp, 64), 9.0, device="cuda", dtype=torch.bfloat16)
    lse = torch.zeros(B, Tp, device="cuda")
    d_qkv = torch.full((2, B, Tp, 192), 9.0, device="cuda", dtype=torch.bfloat16)
    dscr = torch.zeros(B, Tp, device="cuda")
    st = L.stream_ptr()
    L.check(lib.xva_fp_attention_fwd_pairs(L.ptr(qkv), qkv[0].numel(), L.ptr(lens_t), L.ptr(av), av[0].numel(), L.ptr(lse), B, Tp, 0.125, p_drop, seed, stream, st))
    L.check(lib.xva_fp_attention_bwd_pairs(L.ptr(qkv), qkv[0].numel(), L.ptr(av), av[0].numel(), L.ptr(d_av), d_av[0].numel(), L.ptr(lse), L.ptr(dscr),
                                           L.ptr(lens_t), L.ptr(d_qkv), d_qkv[0].numel(), B, Tp, 0.125, p_drop, seed, stream, st))
    torch.cuda.synchronize()
    x = _val(qkv)[:, 1:T + 1].cpu().requires_grad_(True)
    q, k, v = x[..., :64], x[..., 64:128], x[..., 128:]
    score = torch.bmm(q, k.transpose(1, 2)) * 0.125
    kmask = ~(torch.arange(T)[None, :] < torch.tensor(lens)[:, None])
    score = score.masked_fill(kmask[:, None, :], -float("inf"))
    prob = torch.softmax(score, dim=2)
    if p_drop > 0:
        prob = ofp.HashDropout(p_drop, seed).prob(stream, prob)
    out = torch.bmm(prob, v)
    out.backward(_val(d_av)[:, 1:T + 1].cpu())
    lv = live.cpu()
    assert _rel(_val(av)[:, 1:T + 1].cpu()[lv], out.detach()[lv]) < 3e-5
    ref_lse = torch.logsumexp(score.detach(), dim=2)
    assert (lse[:, 1:T + 1].cpu().double() - ref_lse)[lv].abs().max().item() < 2e-5
    g = _val(d_qkv)[:, 1:T + 1].cpu()
    for name, sl in (("dQ", slice(0, 64)), ("dK", slice(64, 128)), ("dV", slice(128, 192))):
        assert _rel(g[..., sl], x.grad[..., sl]) < 6e-5, name
    assert d_qkv[:, :, 0].abs().max().item() == 0.0 and d_qkv[:, :, Tp - 1].abs().max().item() == 0.0
    dead = ~lv
    assert g[dead].abs().max().item() == 0.0 if dead.any() else True
