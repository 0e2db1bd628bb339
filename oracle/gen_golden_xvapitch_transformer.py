"""Golden vectors of the xVAPitch text encoder's RelativePositionTransformer: run the REFERENCE module (python/xvapitch/glow_tts.py, imports
with torch alone) in the build container on seeded inputs, assert that oracle/xvapitch.py:rel_transformer reproduces it, and record the
state_dict, input, output and every parameter / input gradient in tests/golden/xvapitch_transformer.npz.

    python oracle/gen_golden_xvapitch_transformer.py
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
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    glow = importlib.import_module("python.xvapitch.glow_tts")
    torch.manual_seed(21)
    # TextEncoder's configuration (model.py:1125-1136) at a reduced width / depth (hidden 100: a head width, 50, that is no multiple of 8 — like the 98 of hidden 196 = 192 + 4), 2 heads
    B, Cc, Fh, H, L, K, W, T = 3, 100, 64, 2, 2, 3, 4, 29
    lens = torch.tensor([29, 17, 6])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    m = glow.RelativePositionTransformer(in_channels=Cc, out_channels=Cc, hidden_channels=Cc, hidden_channels_ffn=Fh, num_heads=H, num_layers=L, kernel_size=K,
                                         dropout_p=0.0, layer_norm_type="2", rel_attn_window_size=W)
    m.eval()
    for n, p in m.named_parameters():
        if "gamma" in n or "beta" in n:
            p.data += 0.1 * torch.randn_like(p)
    x = torch.randn(B, Cc, T, requires_grad=True)
    r = torch.randn(B, Cc, T)
    y = m(x * 1.0, x_mask)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    xo = x.detach().clone().requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = oxv.rel_transformer(leaves, xo, x_mask, H, L, K, W)
    assert torch.allclose(y, yo, rtol=1e-5, atol=1e-5), float((y - yo).abs().max())
    (y * r).sum().backward()
    (yo * r).sum().backward()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    for n, g in grads.items():
        assert torch.allclose(leaves[n].grad, g, rtol=1e-4, atol=1e-5), (n, float((leaves[n].grad - g).abs().max()))
    assert torch.allclose(xo.grad, x.grad, rtol=1e-4, atol=1e-5)
    out = {"cfg": np.array([B, Cc, Fh, H, L, K, W, T]), "lens": lens.numpy(), "x": x.detach().numpy(), "r": r.numpy(), "y": y.detach().numpy(), "dx": x.grad.numpy()}
    for k, v in sd.items():
        out["sd/" + k] = v.numpy()
    for k, v in grads.items():
        out["grad/" + k] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_transformer.npz")
    np.savez_compressed(path, **out)
    print("xvapitch_transformer.npz", len(out), "arrays", os.path.getsize(path), "bytes; |y|", float(y.abs().mean()))


if __name__ == "__main__":
    main()
