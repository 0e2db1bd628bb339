"""Golden vectors of xVAPitch's waveform decoder: run the REFERENCE HifiganGenerator (python/xvapitch/hifigan.py, constructed with the arguments of
python/xvapitch/model.py:134-149) in the build container on a seeded state_dict (oracle.hifigan.init_vits_decoder_sd — regenerated from the seed
by the tests, the fixture stores its checksum) and seeded latent / speaker inputs; assert oracle/hifigan.py:vits_decoder equal to it; record the
waveform, d(latent) and the gradient of every parameter (norm + evenly spaced samples of all, a few tensors in full) for a fixed linear loss.

    python oracle/gen_golden_vits_decoder.py          -> tests/golden/vits_decoder.npz

Gradient bound: the waveform agrees with the restatement to 1e-5, the gradients only to ~2e-3 — and the SAME restatement evaluated in fp64
instead of fp32 moves its own gradients by as much (d z 1.8e-3, worst parameter 4.4e-3) while its output moves by 9e-7: ~50 M LeakyReLU
gates, a handful of pre-activations within rounding of zero, and a gate that flips changes the derivative discontinuously (slope 1 <-> 0.1).
Gradient comparisons against this fixture therefore use 1e-2."""
import importlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import golden_util as gu, hifigan as ohg, ref_import  # noqa: E402

FULL = ["conv_pre.bias", "cond_layer.bias", "conv_post.weight", "ups.0.weight_g", "ups.3.weight_v",
        "resblocks.0.convs1.0.weight_g", "resblocks.11.convs2.2.weight_v", "resblocks.5.convs1.1.bias"]
SEED, B, CIN, CCOND, T = 777, 2, 192, 512, 32


def main():
    ref_import._install_stubs()
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    hg = importlib.import_module("python.xvapitch.hifigan")
    m = hg.HifiganGenerator(CIN, 1, "1", [[1, 3, 5], [1, 3, 5], [1, 3, 5]], [3, 7, 11], [16, 16, 4, 4], 512, [8, 8, 2, 2], inference_padding=0,
                            cond_channels=CCOND, conv_pre_weight_norm=False, conv_post_weight_norm=False, conv_post_bias=False)
    sd = ohg.init_vits_decoder_sd(SEED, CIN, CCOND)
    assert set(sd) == set(m.state_dict()), sorted(set(sd) ^ set(m.state_dict()))[:8]
    m.load_state_dict(sd)
    m.train()
    gen = torch.Generator().manual_seed(SEED + 1)
    z = torch.randn(B, CIN, T, generator=gen).requires_grad_(True)
    g = torch.nn.functional.normalize(torch.randn(B, CCOND, generator=gen)).unsqueeze(-1)
    r = torch.randn(B, 1, T * 256, generator=gen)
    y = m(z, g=g)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    zo = z.detach().clone().requires_grad_(True)
    yo = ohg.vits_decoder(leaves, zo, g)
    assert torch.allclose(y, yo, rtol=1e-5, atol=1e-6), float((y - yo).abs().max())
    (y * r).sum().backward()
    (yo * r).sum().backward()
    grads = {n: p.grad.detach().clone() for n, p in m.named_parameters()}
    errs = []
    for n, gr in grads.items():
        err = float((leaves[n].grad - gr).norm() / gr.norm()); worst = max(globals().get("_w", 0.0), err); globals()["_w"] = worst
        errs.append((err, n))
    errs.sort(reverse=True)
    assert errs[0][0] < 1e-2, errs[:4]
    print("oracle vs reference: worst parameter-gradient rel error %.2e, d z %.2e" % (globals().get("_w", 0.0), float((zo.grad - z.grad).norm() / z.grad.norm())))
    assert float((zo.grad - z.grad).norm() / z.grad.norm()) < 1e-2
    keys = sorted(grads)
    flat, off = gu.pack_samples(grads, keys, 512)
    out = {"cfg": np.array([SEED, B, CIN, CCOND, T]), "sd_checksum": np.float64(sum(float(v.double().sum()) for v in sd.values())),
           "z": z.detach().numpy(), "g": g.squeeze(-1).numpy(), "r": r.numpy(), "y": y.detach().numpy(), "dz": z.grad.numpy(),
           "grad_keys": np.array(keys), "grad_samples": flat, "grad_offsets": off,
           "grad_norms": np.array([float(grads[k].norm()) for k in keys], dtype=np.float32)}
    for k in FULL:
        out["grad/" + k] = grads[k].numpy()
    path = os.path.join(ROOT, "tests", "golden", "vits_decoder.npz")
    np.savez_compressed(path, **out)
    print("vits_decoder.npz: %.2f MB; |y| max %.3f; %d gradient tensors" % (os.path.getsize(path) / 1e6, float(y.abs().max()), len(keys)))


if __name__ == "__main__":
    main()
