"""GPU parity of the resident-operand convolution weight-gradient kernel (csrc/wgrad_res.h) against an fp64 autograd reference built
from the same bf16-rounded operands: HiFi-GAN's resblock shapes (python/hifigan/models.py:17-48: C = 32 / 64 / 128, k = 3 / 7 / 11,
dilation 1 / 3 / 5) and the grouped / strided scale-discriminator shapes (:203-228: k = 41, stride 1 / 2 / 4, 4 or 16 groups), per-item
K blocks with ragged lengths and the merged (one block) form.  fp32 C: the only difference to the reference is summation order."""
import csv
import os
import tempfile

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from xva_trainer_amd import _lib
    _lib.lib.xva_gemm_set_wgrad.restype = int
    return _lib


def _rel(out, ref):
    return ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


def _case(L, items, T_in, Cin, Cout, G, k, s, d, merged=False, seed=0, alpha=1.0, wgrad=1, bias_grad=None):
    torch.manual_seed(seed)
    P = (k * d - d) // 2
    T_out = (T_in + 2 * P - d * (k - 1) - 1) // s + 1
    Cig, Cog = Cin // G, Cout // G
    PAD = 32
    Hp = PAD + T_in + PAD
    xbuf = torch.zeros(items * Hp + 2 * PAD, Cin, device="cuda", dtype=torch.bfloat16)     # zero guard rows either side (the merged form reads P rows before row 0)
    x = xbuf[PAD:PAD + items * Hp].view(items, Hp, Cin)
    x[:, PAD:PAD + T_in] = torch.randn(items, T_in, Cin, device="cuda").bfloat16()
    if merged:      # output rows laid out like the input (stride 1 only): pad rows of dY are zero, one K block over all rows
        assert s == 1 and T_out == T_in
        dy = torch.zeros(items, Hp, Cout, device="cuda", dtype=torch.bfloat16)
        dy[:, PAD:PAD + T_out] = torch.randn(items, T_out, Cout, device="cuda").bfloat16()
        dyv = dy[:, PAD:PAD + T_out]
    else:
        dy = torch.randn(items, T_out, Cout, device="cuda").bfloat16()
        dyv = dy
    dW = torch.full((G, Cog, k * Cig), 0.5, device="cuda")
    ws = torch.zeros(48 << 18, device="cuda")     # 48 MiB of slabs
    old = L.lib.xva_gemm_set_wgrad(wgrad)
    L.lib.xva_prof_enable(1)
    try:
        kw = dict(layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws, alpha=alpha, seglen=Cig, seg0=0, segstride=d * Cin - Cig,
                  batch2=G, sA2=Cog, sB2=Cig, sC2=Cog * k * Cig, a_rowpitch=Cin)
        if bias_grad is not None:          # the layer's bias gradient from the same launch (xva_gemm colsum_out)
            kw["colsum_out"] = bias_grad.data_ptr()
        if merged:
            L.gemm(dy, xbuf, dW, Cog, k * Cig, items * Hp, Cout, s * Cin, k * Cig, b_offset=(PAD - P) * Cin, **kw)
        else:
            L.gemm(dy, xbuf, dW, Cog, k * Cig, items * T_out, Cout, s * Cin, k * Cig, b_offset=(PAD + PAD - P) * Cin,
                   kb_len=T_out, kb_sA=T_out * Cout, kb_sB=Hp * Cin, **kw)
    finally:
        L.lib.xva_gemm_set_wgrad(old)
        L.lib.xva_prof_enable(0)
    path = os.path.join(tempfile.mkdtemp(), "p.csv")
    L.lib.xva_prof_dump(path.encode())
    rows = list(csv.DictReader(open(path)))
    # the launch must really have run on the resident kernel (a silent fallback to the general tiles would pass the numerics too)
    assert len(rows) == 1 and (int(rows[0]["bn"]) // 100000 == 8) == bool(wgrad), rows
    xr = x[:, PAD:PAD + T_in].double().transpose(1, 2)
    w = torch.zeros(Cout, Cig, k, dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv1d(xr, w, stride=s, padding=P, dilation=d, groups=G)
    assert y.shape[2] == T_out
    (y * dyv.double().transpose(1, 2)).sum().backward()
    ref = 0.5 + alpha * w.grad.permute(0, 2, 1).reshape(G, Cog, k * Cig)
    if bias_grad is not None:
        return dW, ref, alpha * dyv.double().sum(dim=(0, 1))
    return dW, ref


# generator resblock convolutions (stride 1, dense): per-item K blocks and the merged form
@pytest.mark.parametrize("C,k,d,T,merged", [(32, 11, 5, 1024, False), (32, 3, 1, 1000, False), (32, 7, 3, 512, True), (64, 11, 5, 640, False),
                                             (64, 7, 1, 350, False), (64, 3, 3, 512, True), (128, 11, 5, 512, False), (128, 7, 3, 300, True),
                                             (128, 3, 1, 400, False), (16, 5, 1, 700, False)])
def test_resblock_weight_gradients(C, k, d, T, merged):
    L = _lib()
    dW, ref = _case(L, 6, T, C, C, 1, k, 1, d, merged=merged, seed=C + k + d)
    assert _rel(dW, ref) < 3e-6


# grouped / strided convolutions of the scale discriminators (k = 41, 20 zero rows either side)
@pytest.mark.parametrize("Cin,Cout,G,s,T", [(128, 128, 4, 2, 1024), (128, 256, 16, 2, 1100), (256, 512, 16, 4, 2048), (512, 1024, 16, 4, 2052),
                                             (1024, 1024, 16, 1, 544), (128, 128, 4, 2, 1111)])
def test_grouped_strided_weight_gradients(Cin, Cout, G, s, T):
    L = _lib()
    dW, ref = _case(L, 4, T, Cin, Cout, G, 41, s, 1, seed=Cin + s)
    assert _rel(dW, ref) < 3e-6


def test_scale_and_agreement_with_the_general_kernel():
    L = _lib()
    a, ref = _case(L, 5, 900, 64, 64, 1, 11, 1, 3, seed=3, alpha=1.0 / 3, wgrad=1)
    b, _ = _case(L, 5, 900, 64, 64, 1, 11, 1, 3, seed=3, alpha=1.0 / 3, wgrad=0)
    assert _rel(a, ref) < 3e-6 and _rel(b, ref) < 3e-6
    a2, _ = _case(L, 5, 900, 64, 64, 1, 11, 1, 3, seed=3, alpha=1.0 / 3, wgrad=1)
    assert torch.equal(a, a2)       # slabs + ordered reduction: bit-reproducible


# the bias gradient of the same layer out of the weight-gradient launch (one MFMA against a fragment of ones per k-step, one wave of the first column group)
@pytest.mark.parametrize("C,G,k,s,d,T,merged,alpha", [(32, 1, 11, 1, 5, 1024, False, 1.0), (64, 1, 3, 1, 3, 512, True, 1.0 / 3), (128, 1, 7, 1, 1, 300, True, 1.0),
                                                       (128, 1, 3, 1, 1, 400, False, 0.5), (256, 16, 41, 1, 1, 600, False, 1.0), (128, 4, 41, 2, 1, 1024, False, 1.0)])
def test_bias_gradient_from_the_weight_gradient_launch(C, G, k, s, d, T, merged, alpha):
    L = _lib()
    db = torch.full((C,), 0.25, device="cuda")
    dW, ref, ref_b = _case(L, 6, T, C, C, G, k, s, d, merged=merged, seed=C + k + d + 1, alpha=alpha, bias_grad=db)
    assert _rel(dW, ref) < 3e-6
    assert _rel(db - 0.25, ref_b) < 2e-6


def test_bias_gradient_field_is_refused_off_the_resident_kernel():
    """colsum_out on a product another kernel would take is an error, not a silently missing gradient"""
    L = _lib()
    db = torch.zeros(32, device="cuda")
    with pytest.raises(Exception, match="colsum_out"):
        _case(L, 6, 512, 32, 32, 1, 3, 1, 1, wgrad=0, bias_grad=db)
