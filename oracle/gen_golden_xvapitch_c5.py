"""Golden vectors of ONE xVAPitch training iteration's two passes (BASELINE config C5, without the optimiser updates), from the reference's own code:

  generator pass      xVAPitch.forward(batch, optimizer_idx=0), python/xvapitch/model.py:272-364: train_step (compiled from its source lines, with the
                      reference HifiganGenerator as decoder), VitsDiscriminator on (generated, real) segments (:313-315), and the total of
                      VitsGeneratorLoss.forward (python/xvapitch/losses.py:187-300): mel x 45 + KL + duration + pitch + generator loss + feature loss —
                      the feature loss called as the reference calls it, feature_loss(feats_disc_fake, feats_disc_real) (:196), i.e. with its
                      .detach() on the generated features
  discriminator pass  optimizer_idx=1 (:366-384): VitsDiscriminator on the cached (generated.detach(), real) segments, discriminator_loss

Same builder, batch and random draws as gen_golden_xvapitch_genpass.py; the discriminator state_dict is regenerated from a seed
(oracle.hifigan.init_vits_disc_sd).  Records the six generator-side losses, loss_disc, d(total)/d(every generator parameter) and
d(loss_disc)/d(every discriminator parameter) as norms + 256 evenly spaced samples each.

    python oracle/gen_golden_xvapitch_c5.py            -> tests/golden/xvapitch_c5.npz"""
import importlib
import os
import sys
import textwrap

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_xvapitch_genpass as gp, golden_util as gu, hifigan as ohg, ref_import  # noqa: E402

DISC_SEED = 616


def main():
    bd = gp.build()
    ns, xa, m, c = bd["ns"], bd["xa"], bd["m"], bd["c"]
    B, Tt, Ty, x_lens, y_lens, tokens, y, dvec, lids, pitch, wav, zeros_t = (bd[k] for k in ("B", "Tt", "Ty", "x_lens", "y_lens", "tokens", "y", "dvec", "lids",
                                                                                             "pitch", "wav", "zeros_t"))
    hg = importlib.import_module("python.xvapitch.hifigan")
    src = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "model.py")).read()
    dns = {"torch": torch, "nn": torch.nn, "Conv1d": torch.nn.Conv1d, "DiscriminatorP": hg.DiscriminatorP}
    exec(compile(src[src.index("class DiscriminatorS(torch.nn.Module):"):src.index("def mask_from_lens(lens, max_len= None):")], "model.py:VitsDiscriminator", "exec"), dns)
    lsrc = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "losses.py")).read()
    for head, tail in (("    def feature_loss(feats_real, feats_generated):", "    @staticmethod"), ("    def generator_loss(scores_fake):", "    @staticmethod"),
                       ("    def discriminator_loss(scores_real, scores_fake):", "    def forward(self, scores_disc_real, scores_disc_fake):")):
        a = lsrc.index(head)
        exec(compile(textwrap.dedent(lsrc[a:lsrc.index(tail, a)]), "losses.py", "exec"), dns)
    D = dns["VitsDiscriminator"](use_spectral_norm=False)
    dsd = ohg.init_vits_disc_sd(DISC_SEED)
    D.load_state_dict(dsd)
    D.train()
    SEED = 123
    torch.manual_seed(SEED)
    out = m.train_step(tokens, x_lens, y, y_lens, pitch, zeros_t, wav, aux_input={"d_vectors": dvec, "language_ids": lids})
    # ---- generator pass: model.py:313-315 + losses.py:187-300
    scores_fake, feats_fake, _, feats_real = D(out["model_outputs"], out["waveform_seg"])
    mel, mel_hat = gp.reference_mel(xa, out["waveform_seg"], out["model_outputs"])
    y_mask = (torch.arange(Ty)[None, :] < y_lens[:, None]).float()
    loss_mel = F.l1_loss(mel, mel_hat, reduction="none").mean() * 45
    loss_gen = dns["generator_loss"](scores_fake)[0]
    loss_feat = dns["feature_loss"](feats_fake, feats_real)                    # the reference's argument order (losses.py:196)
    loss_kl, _ = ns["kl_loss"](out["z_p"], out["logs_q"], out["m_p"], out["logs_p"], y_mask.unsqueeze(1))
    loss_dur = torch.sum(out["loss_duration"].float())
    lp = F.mse_loss(out["pitch_tgt"], out["pitch_pred"], reduction="none") * out["mask"].unsqueeze(1)
    loss_pitch = lp.sum() / out["mask"].sum() / out["pitch_pred"].shape[0] * 0.1
    loss = loss_kl + loss_feat + loss_mel + loss_gen + loss_dur + loss_pitch  # losses.py:300 (lang_pred_loss = 0)
    m.zero_grad(); D.zero_grad()
    loss.backward()
    ggrads = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}
    # ---- discriminator pass: model.py:366-384
    y_cache, wav_cache = out["model_outputs"].detach(), out["waveform_seg"].detach()
    sf, _, sr, _ = D(y_cache, wav_cache)
    loss_disc, _, _ = dns["discriminator_loss"](sr, sf)
    D.zero_grad()
    loss_disc.backward()
    dgrads = {n: p.grad.detach().clone() for n, p in D.named_parameters()}
    # ---- the restated adversarial terms agree with the reference's
    rs, fr, gs, fg = ohg.vits_disc(dsd, wav_cache, y_cache)
    assert abs(float(ohg.generator_loss(gs)) - float(loss_gen)) < 1e-4 * float(loss_gen) and abs(float(ohg.feature_loss(fr, fg)) - float(loss_feat)) < 1e-4 * float(loss_feat)
    assert abs(float(ohg.discriminator_loss(rs, gs)) - float(loss_disc)) < 1e-4 * float(loss_disc)
    gkeys, dkeys = sorted(ggrads), sorted(dgrads)
    gflat, goff = gu.pack_samples(ggrads, gkeys, 256)
    dflat, doff = gu.pack_samples(dgrads, dkeys, 256)
    res = {"disc_seed": np.int64(DISC_SEED), "disc_checksum": np.float64(sum(float(v.double().sum()) for v in dsd.values())),
           "model_outputs": out["model_outputs"].detach().numpy(),
           "loss_mel": np.float32(loss_mel.item()), "loss_kl": np.float32(loss_kl.item()), "loss_duration": np.float32(loss_dur.item()),
           "loss_pitch": np.float32(loss_pitch.item()), "loss_gen": np.float32(loss_gen.item()), "loss_feat": np.float32(loss_feat.item()),
           "loss": np.float32(loss.item()), "loss_disc": np.float32(loss_disc.item()),
           "g_keys": np.array(gkeys), "g_samples": gflat, "g_offsets": goff, "g_norms": np.array([float(ggrads[k].norm()) for k in gkeys], dtype=np.float32),
           "d_keys": np.array(dkeys), "d_samples": dflat, "d_offsets": doff, "d_norms": np.array([float(dgrads[k].norm()) for k in dkeys], dtype=np.float32)}
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_c5.npz")
    np.savez_compressed(path, **res)
    print("xvapitch_c5.npz: %.2f MB; mel %.4f kl %.4f dur %.4f pitch %.4f gen %.4f feat %.4f | total %.4f | disc %.4f"
          % (os.path.getsize(path) / 1e6, loss_mel.item(), loss_kl.item(), loss_dur.item(), loss_pitch.item(), loss_gen.item(), loss_feat.item(), loss.item(),
             loss_disc.item()))


if __name__ == "__main__":
    main()
