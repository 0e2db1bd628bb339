"""Golden vectors of the stochastic duration predictor's building blocks: run the REFERENCE modules of python/xvapitch/sdp.py (imported in
the build container behind oracle/ref_import.py's stubs) on seeded inputs, assert that oracle/xvapitch.py reproduces them, and record
state_dict, inputs, outputs and every parameter / input gradient in tests/golden/xvapitch_sdp.npz.

    python oracle/gen_golden_xvapitch_sdp.py
"""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402
from oracle import xvapitch as oxv  # noqa: E402


def main():
    ref_import._install_stubs()
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    sdp = importlib.import_module("python.xvapitch.sdp")
    torch.manual_seed(31)
    out = {}
    B, Cc, T, K, L = 3, 48, 41, 3, 3
    lens = torch.tensor([41, 23, 5])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    # ---- DilatedDepthSeparableConv with the conditioning input g (the flows call it with g = the text encoding, sdp.py:160)
    m = sdp.DilatedDepthSeparableConv(Cc, K, L, dropout_p=0.0)
    for n, p in m.named_parameters():
        if "gamma" in n or "beta" in n:
            p.data += 0.1 * torch.randn_like(p)
    x = torch.randn(B, Cc, T, requires_grad=True)
    g = torch.randn(B, Cc, T, requires_grad=True)
    r = torch.randn(B, Cc, T)
    y = m(x * 1.0, x_mask, g=g * 1.0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    xo, go = x.detach().clone().requires_grad_(True), g.detach().clone().requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = oxv.dds_conv(leaves, xo, x_mask, go, K, L)
    assert torch.allclose(y, yo, rtol=1e-5, atol=1e-6), float((y - yo).abs().max())
    (y * r).sum().backward()
    (yo * r).sum().backward()
    for n, p in m.named_parameters():
        assert torch.allclose(leaves[n].grad, p.grad, rtol=1e-4, atol=1e-5), n
        out["dds_grad/" + n] = p.grad.numpy()
    assert torch.allclose(xo.grad, x.grad, rtol=1e-4, atol=1e-5) and torch.allclose(go.grad, g.grad, rtol=1e-4, atol=1e-5)
    out.update({"dds_cfg": np.array([B, Cc, T, K, L]), "lens": lens.numpy(), "dds_x": x.detach().numpy(), "dds_g": g.detach().numpy(), "dds_r": r.numpy(),
                "dds_y": y.detach().numpy(), "dds_dx": x.grad.numpy(), "dds_dg": g.grad.numpy()})
    for k, v in sd.items():
        out["dds_sd/" + k] = v.numpy()
    # ---- ConvFlow (2-channel flow variable; the reference zero-initialises `proj`: give it values so that the spline is exercised), inputs
    # reaching into the linear tails (|x| > 5) as well
    Hh = 32
    cf = sdp.ConvFlow(2, Hh, K, num_layers=3)
    cf.proj.weight.data = 0.8 * torch.randn_like(cf.proj.weight)
    cf.proj.bias.data = 0.3 * torch.randn_like(cf.proj.bias)
    z = (torch.randn(B, 2, T) * 3.0).requires_grad_(True)
    gc = torch.randn(B, Hh, T, requires_grad=True)
    rz = torch.randn(B, 2, T); rl = torch.randn(B)
    zy, zld = cf(z * 1.0, x_mask, g=gc * 1.0)
    sdc = {k: v.detach().clone() for k, v in cf.state_dict().items()}
    zo, gco = z.detach().clone().requires_grad_(True), gc.detach().clone().requires_grad_(True)
    lv = {k: v.clone().requires_grad_(True) for k, v in sdc.items()}
    zyo, zldo = oxv.conv_flow(lv, zo, x_mask, gco, Hh, K, 3)
    assert torch.allclose(zy, zyo, rtol=1e-5, atol=1e-5) and torch.allclose(zld, zldo, rtol=1e-5, atol=1e-4), (float((zy - zyo).abs().max()), float((zld - zldo).abs().max()))
    ((zy * rz).sum() + (zld * rl).sum()).backward()
    ((zyo * rz).sum() + (zldo * rl).sum()).backward()
    for n, p in cf.named_parameters():
        assert torch.allclose(lv[n].grad, p.grad, rtol=1e-4, atol=1e-4), (n, float((lv[n].grad - p.grad).abs().max()))
        out["cf_grad/" + n] = p.grad.numpy()
    assert torch.allclose(zo.grad, z.grad, rtol=1e-4, atol=1e-5) and torch.allclose(gco.grad, gc.grad, rtol=1e-4, atol=1e-5)
    assert float((z.detach().abs() > 5).float().mean()) > 0.02          # some inputs are in the tails
    out.update({"cf_cfg": np.array([B, Hh, T, K, 3, 10]), "cf_z": z.detach().numpy(), "cf_g": gc.detach().numpy(), "cf_rz": rz.numpy(), "cf_rl": rl.numpy(),
                "cf_y": zy.detach().numpy(), "cf_logdet": zld.detach().numpy(), "cf_dz": z.grad.numpy(), "cf_dg": gc.grad.numpy()})
    for k, v in sdc.items():
        out["cf_sd/" + k] = v.numpy()
    # ---- StochasticDurationPredictor, training direction, with speaker (cond) and language conditioning; dropout off; the N(0, 1) draw of
    # sdp.py:281 is the first use of torch's generator inside forward(), so the same seed reproduces it outside
    Cin, Hs, Cg, Cl = 24, 32, 8, 4
    pr = sdp.StochasticDurationPredictor(Cin, Hs, K, 0.0, 4, cond_channels=Cg, language_emb_dim=Cl)
    pr.eval()
    for n, p in pr.named_parameters():
        if n.endswith("proj.weight") and "flows" in n:
            p.data = 0.5 * torch.randn_like(p)
        elif n.endswith("proj.bias") and "flows" in n:
            p.data = 0.2 * torch.randn_like(p)
        elif "log_scale" in n or "translation" in n:
            p.data = 0.3 * torch.randn_like(p)
    xs = torch.randn(B, Cin + Cl, T, requires_grad=True)
    dr = (torch.randint(1, 9, (B, 1, T)).float() * x_mask)
    gs = torch.randn(B, Cg, 1, requires_grad=True)
    le = torch.randn(B, Cl, 1, requires_grad=True)
    rn = torch.randn(B)
    torch.manual_seed(99)
    nll = pr(xs * 1.0, x_mask, dr=dr, g=gs * 1.0, lang_emb=le * 1.0)
    torch.manual_seed(99)
    noise = torch.randn(B, 2, T)
    sds = {k: v.detach().clone() for k, v in pr.state_dict().items()}
    xso, gso, leo = (t.detach().clone().requires_grad_(True) for t in (xs, gs, le))
    lvs = {k: v.clone().requires_grad_(True) for k, v in sds.items()}
    nllo = oxv.sdp_forward(lvs, xso, x_mask, dr, noise, Hs, K, 4, g=gso, lang_emb=leo)
    assert torch.allclose(nll, nllo, rtol=1e-5, atol=1e-3), (nll, nllo)
    (nll * rn).sum().backward()
    (nllo * rn).sum().backward()
    for n, p in pr.named_parameters():
        # ten flows deep in fp32: compared by relative L2 norm (element-wise the two fp32 evaluation orders differ by ~1e-3 of the tensor's scale)
        rel = float((lvs[n].grad - p.grad).norm() / p.grad.norm().clamp_min(1e-12))
        assert rel < 2e-4, (n, rel)
        out["sdp_grad/" + n] = p.grad.numpy()
    assert torch.allclose(xso.grad, xs.grad, rtol=1e-3, atol=1e-5) and torch.allclose(gso.grad, gs.grad, rtol=1e-3, atol=1e-4)
    out.update({"sdp_cfg": np.array([B, Cin, Hs, Cg, Cl, T, K]), "sdp_x": xs.detach().numpy(), "sdp_dr": dr.numpy(), "sdp_g": gs.detach().numpy(), "sdp_lang": le.detach().numpy(),
                "sdp_noise": noise.numpy(), "sdp_r": rn.numpy(), "sdp_nll": nll.detach().numpy(), "sdp_dx": xs.grad.numpy(), "sdp_dg": gs.grad.numpy(),
                "sdp_dlang": le.grad.numpy()})
    for k, v in sds.items():
        out["sdp_sd/" + k] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_sdp.npz")
    np.savez_compressed(path, **out)
    print("xvapitch_sdp.npz", len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
