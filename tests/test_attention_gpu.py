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
