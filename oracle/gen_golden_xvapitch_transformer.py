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
    # ---- the pitch / energy encoders' form (model.py:1292-1305): out_channels = 1, so `proj` exists and the stack returns proj(x)
    C2, F2 = 36, 32
    m2 = glow.RelativePositionTransformer(in_channels=C2, out_channels=1, hidden_channels=C2, hidden_channels_ffn=F2, num_heads=H, num_layers=2, kernel_size=K,
                                          dropout_p=0.0, layer_norm_type="2", rel_attn_window_size=W)
    m2.eval()
    x2 = torch.randn(B, C2, T, requires_grad=True)
    r2 = torch.randn(B, 1, T)
    y2 = m2(x2 * 1.0, x_mask)
    sd2 = {k: v.detach().clone() for k, v in m2.state_dict().items()}
    x2o = x2.detach().clone().requires_grad_(True)
    leaves2 = {k: v.clone().requires_grad_(True) for k, v in sd2.items()}
    y2o = oxv.rel_transformer(leaves2, x2o, x_mask, H, 2, K, W)
    assert y2.shape == (B, 1, T) and torch.allclose(y2, y2o, rtol=1e-5, atol=1e-5)
    (y2 * r2).sum().backward()
    (y2o * r2).sum().backward()
    out.update({"p_cfg": np.array([B, C2, F2, H, 2, K, W, T]), "p_x": x2.detach().numpy(), "p_r": r2.numpy(), "p_y": y2.detach().numpy(), "p_dx": x2.grad.numpy()})
    for k, v in sd2.items():
        out["p_sd/" + k] = v.numpy()
    for n, p in m2.named_parameters():
        if p.grad is not None:                                  # the last layer's feed-forward network and second norm do not reach the output
            assert torch.allclose(leaves2[n].grad, p.grad, rtol=1e-4, atol=1e-5), n
            out["p_grad/" + n] = p.grad.numpy()
        else:
            assert leaves2[n].grad is None or float(leaves2[n].grad.abs().max()) == 0.0, n
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_transformer.npz")
    np.savez_compressed(path, **out)
    print("xvapitch_transformer.npz", len(out), "arrays", os.path.getsize(path), "bytes; |y|", float(y.abs().mean()))


if __name__ == "__main__":
    main()
