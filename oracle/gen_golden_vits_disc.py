"""Golden vectors of xVAPitch's VitsDiscriminator passes: run the REFERENCE classes (DiscriminatorS / VitsDiscriminator compiled in memory from
python/xvapitch/model.py:1548-1640, DiscriminatorP imported from python/xvapitch/hifigan.py) and the reference loss functions
(python/xvapitch/losses.py:64-84 feature / generator loss, :331-343 discriminator loss, compiled from their source lines) on a seeded state_dict
(oracle.hifigan.init_vits_disc_sd, regenerated from the seed by the tests) and two synthetic waveforms:

  D pass:  loss_disc = discriminator_loss(D(y), D(y_hat.detach()))            -> d loss / d every discriminator parameter
  G pass:  loss_gen + loss_feat = generator_loss(D(y_hat)) + feature_loss     -> d / d y_hat

    python oracle/gen_golden_vits_disc.py            -> tests/golden/vits_disc.npz
Asserts oracle/hifigan.py:vits_disc equal to the reference first."""
import importlib
import os
import sys
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import golden_util as gu, hifigan as ohg, mel as omel, ref_import  # noqa: E402

SEED, B, SEG = 515, 2, 8192
FULL = ["nets.0.convs.0.weight_v", "nets.0.convs.1.weight_v", "nets.0.convs.4.weight_g", "nets.0.conv_post.weight_v", "nets.0.convs.3.bias",
        "nets.1.convs.0.weight_v", "nets.5.conv_post.weight_v", "nets.3.convs.2.bias"]


def main():
    ref_import._install_stubs()
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    hg = importlib.import_module("python.xvapitch.hifigan")
    src = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "model.py")).read()
    ns = {"torch": torch, "nn": torch.nn, "Conv1d": torch.nn.Conv1d, "DiscriminatorP": hg.DiscriminatorP}
    exec(compile(src[src.index("class DiscriminatorS(torch.nn.Module):"):src.index("def mask_from_lens(lens, max_len= None):")], "model.py:VitsDiscriminator", "exec"), ns)
    lsrc = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "losses.py")).read()
    a = lsrc.index("    def feature_loss(feats_real, feats_generated):")
    exec(compile(textwrap.dedent(lsrc[a:lsrc.index("    @staticmethod", a)]), "losses.py:feature_loss", "exec"), ns)
    a = lsrc.index("    def generator_loss(scores_fake):")
    exec(compile(textwrap.dedent(lsrc[a:lsrc.index("    @staticmethod", a)]), "losses.py:generator_loss", "exec"), ns)
    a = lsrc.index("    def discriminator_loss(scores_real, scores_fake):")
    exec(compile(textwrap.dedent(lsrc[a:lsrc.index("    def forward(self, scores_disc_real, scores_disc_fake):", a)]), "losses.py:discriminator_loss", "exec"), ns)
    D = ns["VitsDiscriminator"](use_spectral_norm=False)
    sd = ohg.init_vits_disc_sd(SEED)
    assert list(sd) == [k for k in D.state_dict()], "key order differs from VitsDiscriminator.state_dict()"
    D.load_state_dict(sd)
    D.train()
    y = torch.from_numpy(np.stack([omel.peak_normalize(omel.synth_wave(SEG, SEED + i)) * 0.95 for i in range(B)]).astype(np.float32)).unsqueeze(1)
    gen = torch.Generator().manual_seed(SEED + 9)
    y_hat = (0.7 * torch.roll(y, 37, dims=2) + 0.1 * torch.randn(B, 1, SEG, generator=gen)).clamp(-1, 1).requires_grad_(True)
    # ---- D pass (xvapitch/model.py:366-384 + VitsDiscriminatorLoss)
    sr, _, sf, _ = D(y, y_hat.detach())
    loss_disc, _, _ = ns["discriminator_loss"](sr, sf)
    D.zero_grad()
    loss_disc.backward()
    grads = {n: p.grad.detach().clone() for n, p in D.named_parameters()}
    # ---- G pass (:313-315 + losses.py:195-196)
    sr, fr, sf, ff = D(y, y_hat)
    loss_gen = ns["generator_loss"](sf)[0]
    loss_feat = ns["feature_loss"](fr, ff)
    (loss_gen + loss_feat).backward()
    d_wav = y_hat.grad.detach().clone()
    # ---- restatement
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yh = y_hat.detach().clone().requires_grad_(True)
    rs, fr2, gs, fg2 = ohg.vits_disc(leaves, y, yh.detach())
    ld = ohg.discriminator_loss(rs, gs)
    assert abs(float(ld.detach()) - float(loss_disc)) < 1e-5 * float(loss_disc)
    ld.backward()
    worst = max((float((leaves[n].grad - g).norm() / g.norm()), n) for n, g in grads.items())
    rs, fr2, gs, fg2 = ohg.vits_disc({k: v.detach() for k, v in leaves.items()}, y, yh)
    lg, lf = ohg.generator_loss(gs), ohg.feature_loss([[t.detach() for t in f] for f in fr2], fg2)
    assert abs(float(lg.detach()) - float(loss_gen)) < 1e-5 * float(loss_gen) and abs(float(lf.detach()) - float(loss_feat)) < 1e-5 * float(loss_feat)
    (lg + lf).backward()
    edw = float((yh.grad - d_wav).norm() / d_wav.norm())
    print("oracle vs reference: worst parameter gradient %.2e (%s), d_wav %.2e" % (worst[0], worst[1], edw))
    assert worst[0] < 1e-2 and edw < 1e-2
    keys = sorted(grads)
    flat, off = gu.pack_samples(grads, keys, 512)
    out = {"cfg": np.array([SEED, B, SEG]), "sd_checksum": np.float64(sum(float(v.double().sum()) for v in sd.values())),
           "y": y.squeeze(1).numpy(), "y_hat": y_hat.detach().squeeze(1).numpy(), "loss_disc": np.float32(loss_disc.item()),
           "loss_gen": np.float32(loss_gen.item()), "loss_feat": np.float32(loss_feat.item()), "d_wav": d_wav.squeeze(1).numpy(),
           "grad_keys": np.array(keys), "grad_samples": flat, "grad_offsets": off, "grad_norms": np.array([float(grads[k].norm()) for k in keys], dtype=np.float32)}
    for k in FULL:
        out["grad/" + k] = grads[k].numpy()
    path = os.path.join(ROOT, "tests", "golden", "vits_disc.npz")
    np.savez_compressed(path, **out)
    print("vits_disc.npz: %.2f MB; loss_disc %.5f loss_gen %.5f loss_feat %.5f; %d gradient tensors" % (os.path.getsize(path) / 1e6, loss_disc.item(), loss_gen.item(),
                                                                                                        loss_feat.item(), len(keys)))


if __name__ == "__main__":
    main()
