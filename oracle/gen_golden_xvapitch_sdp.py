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
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_sdp.npz")
    np.savez_compressed(path, **out)
    print("xvapitch_sdp.npz", len(out), "arrays", os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
