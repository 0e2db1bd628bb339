"""Golden vectors that pin the DROPOUT SITES of oracle/xvapitch.py to the reference: run the reference's RelativePositionTransformer (both forms
TextEncoder / the pitch predictor build, python/xvapitch/glow_tts.py) and StochasticDurationPredictor (python/xvapitch/sdp.py) in TRAIN mode in
the build container, with nn.Dropout.forward replaced by a deterministic stand-in — the k-th call with p > 0 multiplies by the keyed-hash mask
of site k's id over the tensor's own flat order (oracle/xvapitch.py HashDrop, layout "flat") — assert that the oracle with the same hook at its
sites reproduces outputs and every gradient, and record inputs / state_dicts / outputs / gradients in tests/golden/xvapitch_dropout.npz.

    python oracle/gen_golden_xvapitch_dropout.py
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

SEED = 20260929


class PatchedDropout:
    """nn.Dropout.forward -> the site hook, in call order (calls with p == 0 — the flows' own DilatedDepthSeparableConv — are not sites)"""

    def __init__(self, sites, hook):
        self.sites, self.hook, self.k = list(sites), hook, 0

    def __enter__(self):
        self.orig = torch.nn.Dropout.forward
        me = self

        def fwd(mod, x):
            if mod.p <= 0 or not mod.training:
                return x
            site = me.sites[me.k]
            me.k += 1
            return me.hook(site, x)
        torch.nn.Dropout.forward = fwd
        return self

    def __exit__(self, *a):
        torch.nn.Dropout.forward = self.orig
        assert self.k == len(self.sites), "the reference called nn.Dropout %d times, %d sites expected" % (self.k, len(self.sites))


def main():
    ref_import._install_stubs()
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    glow = importlib.import_module("python.xvapitch.glow_tts")
    sdp = importlib.import_module("python.xvapitch.sdp")
    torch.manual_seed(77)
    out = {"seed": np.array([SEED], dtype=np.int64)}
    B, T = 3, 29
    lens = torch.tensor([29, 17, 6])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    out["lens"] = lens.numpy()

    def transformer(tag, Cc, Co, Fh, H, L, K, W, p):
        m = glow.RelativePositionTransformer(in_channels=Cc, out_channels=Co, hidden_channels=Cc, hidden_channels_ffn=Fh, num_heads=H, num_layers=L, kernel_size=K,
                                             dropout_p=p, layer_norm_type="2", rel_attn_window_size=W)
        m.train()
        for n, q in m.named_parameters():
            if "gamma" in n or "beta" in n:
                q.data += 0.1 * torch.randn_like(q)
        x = torch.randn(B, Cc, T, requires_grad=True)
        r = torch.randn(B, Co, T)
        hook = oxv.HashDrop(p, SEED, layout="flat")
        with PatchedDropout([4 * i + s for i in range(L) for s in range(4)], hook):
            y = m(x * 1.0, x_mask)
        sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
        xo = x.detach().clone().requires_grad_(True)
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        yo = oxv.rel_transformer(leaves, xo, x_mask, H, L, K, W, drop=hook)
        assert torch.allclose(y, yo, rtol=1e-5, atol=1e-5), float((y - yo).abs().max())
        y_eval = oxv.rel_transformer(sd, x.detach(), x_mask, H, L, K, W)
        assert float((y_eval - y.detach()).abs().max()) > 1e-2, "the dropout stand-in did not change the output"
        (y * r).sum().backward()
        (yo * r).sum().backward()
        for n, q in m.named_parameters():
            if q.grad is not None:
                assert torch.allclose(leaves[n].grad, q.grad, rtol=1e-4, atol=1e-5), (tag, n, float((leaves[n].grad - q.grad).abs().max()))
                out["%s_grad/%s" % (tag, n)] = q.grad.numpy()
            else:
                assert leaves[n].grad is None or float(leaves[n].grad.abs().max()) == 0.0, n
        assert torch.allclose(xo.grad, x.grad, rtol=1e-4, atol=1e-5)
        out.update({tag + "_cfg": np.array([B, Cc, Co, Fh, H, L, K, W, T]), tag + "_p": np.array([p], dtype=np.float32), tag + "_x": x.detach().numpy(),
                    tag + "_r": r.numpy(), tag + "_y": y.detach().numpy(), tag + "_dx": x.grad.numpy()})
        for k, v in sd.items():
            out["%s_sd/%s" % (tag, k)] = v.numpy()
        return float(y.abs().mean())

    a1 = transformer("te", 100, 100, 64, 2, 2, 3, 4, 0.1)          # TextEncoder's form (model.py:1125-1136), reduced
    a2 = transformer("pp", 36, 1, 32, 2, 2, 3, 4, 0.1)             # the pitch predictor's form (out_channels 1, model.py:1292-1305)

    # ---- StochasticDurationPredictor, training direction, dropout 0.5 in convs / post_convs (model.py:124-132)
    Cin, Hh, Cg, Cl, Ts = 16, 32, 8, 4, 23
    lens2 = torch.tensor([23, 12, 4])
    m_ = (torch.arange(Ts)[None, :] < lens2[:, None]).float().unsqueeze(1)
    dp = sdp.StochasticDurationPredictor(Cin, Hh, 3, 0.5, 4, cond_channels=Cg, language_emb_dim=Cl)
    dp.train()
    for n, q in dp.named_parameters():
        if n.endswith("proj.weight") and "flows" in n:                # the reference zero-initialises the splines' projections
            q.data = 0.1 * torch.randn_like(q)
        if n.endswith("proj.bias") and "flows" in n:
            q.data = 0.05 * torch.randn_like(q)
    x = torch.randn(B, Cin + Cl, Ts, requires_grad=True)
    dr = (torch.randint(1, 6, (B, 1, Ts)).float() * m_)
    g = torch.randn(B, Cg, 1)
    le = torch.randn(B, Cl, 1)
    noise = torch.randn(B, 2, Ts)
    hook = oxv.HashDrop(0.5, SEED + 1, layout="flat")
    orig_randn = torch.randn

    def fixed_randn(*a, **k):                                           # sdp.py:281 draws the dequantisation noise inside forward
        return noise.clone() if tuple(a[0] if isinstance(a[0], (tuple, list, torch.Size)) else a) == tuple(noise.shape) else orig_randn(*a, **k)
    torch.randn = fixed_randn
    try:
        with PatchedDropout(list(range(6)), hook):
            nll = dp(x * 1.0, m_, dr=dr, g=g, lang_emb=le)
    finally:
        torch.randn = orig_randn
    sd = {k: v.detach().clone() for k, v in dp.state_dict().items()}
    xo = x.detach().clone().requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    nllo = oxv.sdp_forward(leaves, xo, m_, dr, noise, Hh, 3, 4, g=g, lang_emb=le, drop=hook)
    assert torch.allclose(nll, nllo, rtol=1e-4, atol=1e-3), (nll, nllo)
    nll_eval = oxv.sdp_forward(sd, x.detach(), m_, dr, noise, Hh, 3, 4, g=g, lang_emb=le)
    assert float((nll_eval - nll.detach()).abs().max()) > 1e-2
    nll.sum().backward()
    nllo.sum().backward()
    for n, q in dp.named_parameters():
        assert torch.allclose(leaves[n].grad, q.grad, rtol=1e-3, atol=1e-4), (n, float((leaves[n].grad - q.grad).abs().max()))
        out["sdp_grad/" + n] = q.grad.numpy()
    assert torch.allclose(xo.grad, x.grad, rtol=1e-3, atol=1e-4)
    out.update({"sdp_cfg": np.array([B, Cin, Hh, Cg, Cl, Ts]), "sdp_lens": lens2.numpy(), "sdp_x": x.detach().numpy(), "sdp_dr": dr.numpy(), "sdp_g": g.numpy(),
                "sdp_le": le.numpy(), "sdp_noise": noise.numpy(), "sdp_nll": nll.detach().numpy(), "sdp_dx": x.grad.numpy()})
    for k, v in sd.items():
        out["sdp_sd/" + k] = v.numpy()
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_dropout.npz")
    np.savez_compressed(path, **out)
    print("xvapitch_dropout.npz", len(out), "arrays", os.path.getsize(path), "bytes; |y|", a1, a2, "nll", nll.detach().tolist())


if __name__ == "__main__":
    main()
