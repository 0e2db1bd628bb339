"""Round 5: the four-rows-per-wave bf16 LayerNorm forward (csrc/fp_ops.hip layernorm_fwd4_kernel; transformer.py:75,146) and the backward fed by its
statistics, against the one-row-per-wave forward it replaces (xva_fp_set_ln4(0)) and against an fp64 restatement, through the C ABI, incl. ragged lengths, row
counts that are no multiple of four, the dropout-masked second output and the dgamma / dbeta sums."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
CH = 384


def _call_fwd(L, _lib, x, gamma, beta, lens, Tp, mask):
    rows = x.shape[0]
    y = torch.full_like(x, 7.0); mean = torch.zeros(rows, device="cuda"); rstd = torch.zeros(rows, device="cuda")
    rc = L.xva_fp_layernorm_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), 1, _lib.ptr(mean), _lib.ptr(rstd), C.c_int64(rows), CH, mask,
                                _lib.ptr(lens), Tp, C.c_float(0.0), C.c_uint64(0), 0, _lib.stream_ptr())
    assert rc == 0, L.xva_last_error()
    return y, mean, rstd


def _call_bwd(L, _lib, dy, x, mean, rstd, gamma, lens, Tp, mask, drop):
    rows = x.shape[0]
    dx = torch.full_like(x, 7.0); dxm = torch.full_like(x, 7.0); dg = torch.zeros(CH, device="cuda"); db = torch.zeros(CH, device="cuda")
    rc = L.xva_fp_layernorm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dx), _lib.ptr(dxm) if drop else None, 1,
                                _lib.ptr(dg), _lib.ptr(db), C.c_int64(rows), CH, mask, _lib.ptr(lens), Tp, 0, C.c_float(0.0), C.c_uint64(0), 0,
                                C.c_float(0.1 if drop else 0.0), C.c_uint64(12345), 7, None, None, _lib.stream_ptr())
    assert rc == 0, L.xva_last_error()
    return dx, (dxm if drop else None), dg, db


@pytest.mark.parametrize("B,T,mask", [(3, 37, 2), (5, 101, 1), (2, 860, 2), (1, 5, 0)])
def test_ln4_equals_the_one_row_kernels_and_fp64(B, T, mask):
    from xva_trainer_amd import _lib
    L = _lib.lib
    Tp = T + 2
    rows = B * Tp                                              # 117, 515, 1724, 7: multiples of four and not
    g = torch.Generator().manual_seed(B * 1000 + T)
    x = (torch.randn(rows, CH, generator=g) * 1.5 + 0.3).cuda().bfloat16()
    dy = torch.randn(rows, CH, generator=g).cuda().bfloat16()
    gamma = (1 + 0.2 * torch.randn(CH, generator=g)).cuda(); beta = (0.1 * torch.randn(CH, generator=g)).cuda()
    lens = torch.randint(max(1, T // 2), T + 1, (B,), generator=g).int().cuda()
    res = {}
    for mode in (1, 0):
        old = L.xva_fp_set_ln4(mode)
        try:
            y, mean, rstd = _call_fwd(L, _lib, x, gamma, beta, lens, Tp, mask)
            out = [y, mean, rstd]
            for drop in (0, 1):
                out += list(_call_bwd(L, _lib, dy, x, mean, rstd, gamma, lens, Tp, mask, drop))
            torch.cuda.synchronize()
            res[mode] = out
        finally:
            L.xva_fp_set_ln4(old)
    names = ["y", "mean", "rstd", "dx", None, "dgamma", "dbeta", "dx(drop)", "dxm", "dgamma(drop)", "dbeta(drop)"]
    for n, a, b in zip(names, res[1], res[0]):
        if n is None:
            continue
        a, b = a.double(), b.double()
        tol = 8e-3 if a.dtype == torch.bfloat16 or n in ("y", "dx", "dx(drop)", "dxm") else 1e-5      # bf16 outputs: a different summation order may flip a last bit
        assert ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() < tol, n
        if n in ("y", "dx", "dxm"):
            assert (a != b).double().mean().item() < 2e-2, n                                          # ... on a few elements only
    # fp64 restatement of the rows' arithmetic (live rows; dead rows are zero)
    t = torch.arange(rows) % Tp
    b_of = torch.arange(rows) // Tp
    live = torch.ones(rows, dtype=torch.bool) if mask == 0 else ((t > 0) & (t < Tp - 1) & ((mask == 1) | (t <= lens.cpu()[b_of])))
    xd = x.double().cpu(); mu = xd.mean(1, keepdim=True); var = ((xd - mu) ** 2).mean(1, keepdim=True); rs = (var + 1e-5).rsqrt()
    yref = ((xd - mu) * rs * gamma.double().cpu() + beta.double().cpu()) * live[:, None]
    y = res[1][0].double().cpu()
    assert ((y - yref).abs().max() / yref.abs().max()).item() < 6e-3
    assert torch.allclose(res[1][1].double().cpu(), mu[:, 0], atol=1e-5) and torch.allclose(res[1][2].double().cpu(), rs[:, 0], rtol=1e-4)
    gd = dy.double().cpu() * live[:, None]
    xh = (xd - mu) * rs; dh = gd * gamma.double().cpu()
    dxref = rs * (dh - dh.mean(1, keepdim=True) - xh * (dh * xh).mean(1, keepdim=True)) * live[:, None]
    dx = res[1][3].double().cpu()
    assert ((dx - dxref).abs().max() / dxref.abs().max()).item() < 6e-3
    assert torch.allclose(res[1][5].double().cpu(), (gd * xh).sum(0), rtol=1e-4, atol=1e-3) and torch.allclose(res[1][6].double().cpu(), gd.sum(0), rtol=1e-4, atol=1e-3)
    # the masked copy: zero or dx / 0.9, the same mask in both kernels
    dxm1, dxm0, dx1 = res[1][8].float(), res[0][8].float(), res[1][7].float()
    assert torch.equal(dxm1 == 0, dxm0 == 0) or ((dxm1 == 0) != (dxm0 == 0)).float().mean().item() < 1e-3
    kept = dxm1 != 0
    assert 0.85 < kept.float().mean().item() / max(1e-9, (dx1 != 0).float().mean().item()) < 0.95
    assert ((dxm1[kept] - dx1[kept] / 0.9).abs().max() / dx1.abs().max()).item() < 8e-3
