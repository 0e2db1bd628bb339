"""HiFi-GAN golden vectors: run the REFERENCE Generator / MultiPeriodDiscriminator / MultiScaleDiscriminator, its loss
functions and torch.optim.AdamW through one training iteration exactly as python/hifigan/xva_train.py:479-515 does, on
seeded state_dicts + synthetic audio, and record losses, the generated waveform, gradient and post-step summaries.
State_dicts are regenerated from the seed (oracle.hifigan.init_*_sd); the fixture stores checksums."""
import itertools
import json
import os

import numpy as np
import torch
import torch.nn.functional as F

from oracle import golden_util as gu
from oracle import hifigan as ohg

# gradient tensors stored in full (checkpoint layout; "g." = generator, "mpd." / "msd." = discriminators), next to the samples of all 404
FULL_GRADS = [
    "g.ups.3.weight_v", "g.ups.2.weight_v", "g.resblocks.3.convs1.1.weight_v", "g.resblocks.8.convs2.2.weight_v",
    "g.resblocks.11.convs1.0.weight_v", "g.resblocks.6.convs1.0.weight_v", "g.conv_post.weight_v", "g.conv_pre.bias",
    "g.resblocks.0.convs1.0.weight_g",
    "mpd.discriminators.0.convs.1.weight_v", "mpd.discriminators.3.convs.0.weight_v", "mpd.discriminators.4.conv_post.weight_v",
    "mpd.discriminators.2.convs.4.bias", "mpd.discriminators.1.convs.3.weight_g",
    "msd.discriminators.0.convs.0.weight_orig", "msd.discriminators.0.convs.2.weight_orig", "msd.discriminators.0.conv_post.weight_orig",
    "msd.discriminators.1.convs.2.weight_v", "msd.discriminators.2.conv_post.weight_v", "msd.discriminators.2.convs.5.weight_g",
]

CASES = [("hg_step_b2", 2, 4321)]


def run_reference(ns, g_sd, mpd_sd, msd_sd, x, y_wav, y_mel):
    hm = ns.hifigan_models
    h = hm.AttrDict(json.load(open(os.path.join(os.path.dirname(hm.__file__), "config_v1.json"))))
    h.USE_EMB_CONDITIONING = False
    G, MPD, MSD = hm.Generator(h), hm.MultiPeriodDiscriminator(), hm.MultiScaleDiscriminator()
    G.load_state_dict(g_sd); MPD.load_state_dict(mpd_sd); MSD.load_state_dict(msd_sd)
    G.train(); MPD.train(); MSD.train()
    optim_g = torch.optim.AdamW(G.parameters(), h.learning_rate, betas=[h.adam_b1, h.adam_b2])
    optim_d = torch.optim.AdamW(itertools.chain(MSD.parameters(), MPD.parameters()), h.learning_rate, betas=[h.adam_b1, h.adam_b2])
    mel_fn = ns.hifigan_meldataset.mel_spectrogram
    y = y_wav.unsqueeze(1)
    y_g_hat = G(x)
    y_g_hat_mel = mel_fn(y_g_hat.squeeze(1), h.n_fft, h.num_mels, h.sampling_rate, h.hop_size, h.win_size, h.fmin, h.fmax_for_loss)
    optim_d.zero_grad()
    y_df_hat_r, y_df_hat_g, _, _ = MPD(y, y_g_hat.detach())
    loss_disc_f, _, _ = hm.discriminator_loss(y_df_hat_r, y_df_hat_g)
    y_ds_hat_r, y_ds_hat_g, _, _ = MSD(y, y_g_hat.detach())
    loss_disc_s, _, _ = hm.discriminator_loss(y_ds_hat_r, y_ds_hat_g)
    loss_disc_all = loss_disc_s + loss_disc_f
    loss_disc_all.backward()
    d_grads = {"mpd." + n: p.grad.detach().clone() for n, p in MPD.named_parameters()}
    d_grads.update({"msd." + n: p.grad.detach().clone() for n, p in MSD.named_parameters()})
    optim_d.step()
    optim_g.zero_grad()
    loss_mel = F.l1_loss(y_mel, y_g_hat_mel) * 45
    y_df_hat_r, y_df_hat_g, fmap_f_r, fmap_f_g = MPD(y, y_g_hat)
    y_ds_hat_r, y_ds_hat_g, fmap_s_r, fmap_s_g = MSD(y, y_g_hat)
    loss_fm_f = hm.feature_loss(fmap_f_r, fmap_f_g)
    loss_fm_s = hm.feature_loss(fmap_s_r, fmap_s_g)
    loss_gen_f, _ = hm.generator_loss(y_df_hat_g)
    loss_gen_s, _ = hm.generator_loss(y_ds_hat_g)
    loss_gen_all = loss_gen_s + loss_gen_f + loss_fm_s + loss_fm_f + loss_mel
    loss_gen_all.backward()
    g_grads = {n: p.grad.detach().clone() for n, p in G.named_parameters()}
    optim_g.step()
    out = {"loss_disc_all": float(loss_disc_all), "loss_disc_f": float(loss_disc_f), "loss_disc_s": float(loss_disc_s),
           "loss_gen_all": float(loss_gen_all), "loss_mel": float(loss_mel), "loss_fm_f": float(loss_fm_f), "loss_fm_s": float(loss_fm_s),
           "loss_gen_f": float(loss_gen_f), "loss_gen_s": float(loss_gen_s)}
    return out, g_grads, d_grads, y_g_hat.detach(), G.state_dict(), MPD.state_dict(), MSD.state_dict()


def generate(ns, out_dir):
    for name, B, seed in CASES:
        g_sd, mpd_sd, msd_sd = ohg.init_generator_sd(seed), ohg.init_mpd_sd(seed + 1), ohg.init_msd_sd(seed + 2)
        x, y_wav, y_mel = ohg.synth_batch(B, seed + 3)
        c = lambda sd: {k: v.clone() for k, v in sd.items()}
        ref_out, ref_gg, ref_dg, ref_y, G2, P2, S2 = run_reference(ns, c(g_sd), c(mpd_sd), c(msd_sd), x, y_wav, y_mel)
        g2, p2, s2 = c(g_sd), c(mpd_sd), c(msd_sd)
        o_out, o_gg, o_dg, o_y = ohg.train_step(g2, p2, s2, x, y_wav, y_mel, {}, {})
        # ---- pin the oracle to the live reference before writing the fixture
        for k in ref_out:
            assert abs(o_out[k] - ref_out[k]) <= 2e-5 * max(1.0, abs(ref_out[k])), (k, o_out[k], ref_out[k])
        assert torch.allclose(o_y, ref_y, rtol=1e-4, atol=2e-5), float((o_y - ref_y).abs().max())
        def nrel(a, b):
            return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
        worst = max(((nrel(o_gg[k], v), "G " + k) for k, v in ref_gg.items()))
        worst_d = max(((nrel(o_dg[k], v), "D " + k) for k, v in ref_dg.items()))
        print("worst grad rel-norm errors oracle vs reference:", worst, worst_d)
        assert worst[0] < 1e-3 and worst_d[0] < 1e-3
        # The first AdamW step is ~lr * sign(g), ill-conditioned where |g| ~ fp32 noise, so the optimizer restatement is pinned
        # on IDENTICAL gradients: oracle.adamw_step(reference grads) must reproduce the reference's post-step parameters.
        chk = c(g_sd)
        ohg.adamw_step({k: chk[k] for k in ref_gg}, ref_gg, {})
        for k in ref_gg:
            assert torch.allclose(chk[k], G2[k], rtol=1e-5, atol=1e-7), ("AdamW G", k)
        chk_p, chk_s = c(mpd_sd), c(msd_sd)
        params = {k: (chk_p if k.startswith("mpd.") else chk_s)[k[4:]] for k in ref_dg}
        ohg.adamw_step(params, ref_dg, {})
        for k in ref_dg:
            assert torch.allclose(params[k], (P2 if k.startswith("mpd.") else S2)[k[4:]], rtol=1e-5, atol=1e-7), ("AdamW D", k)
        for k in ("discriminators.0.convs.0.weight_u", "discriminators.0.convs.3.weight_v", "discriminators.0.conv_post.weight_u"):
            assert torch.allclose(s2[k], S2[k], rtol=1e-4, atol=1e-6), ("spectral-norm buffer", k)
        gk, dk = sorted(ref_gg), sorted(ref_dg)
        rec = {
            "seed": np.int64(seed), "B": np.int64(B),
            "loss_names": np.array(sorted(ref_out)), "losses": np.array([ref_out[k] for k in sorted(ref_out)]),
            "x_mel": x.numpy(), "y_wav": y_wav.numpy(), "y_mel": y_mel.numpy(), "y_g_hat": ref_y.numpy(),
            "g_grad_keys": np.array(gk), "g_grad_l2": np.array([float(ref_gg[k].double().norm()) for k in gk]),
            "d_grad_keys": np.array(dk), "d_grad_l2": np.array([float(ref_dg[k].double().norm()) for k in dk]),
            "g_delta_l2": np.array([float((G2[k].double() - g_sd[k].double()).norm()) for k in gk]),
            "d_delta_l2": np.array([float(((P2 if k.startswith("mpd.") else S2)[k[4:]].double()
                                            - (mpd_sd if k.startswith("mpd.") else msd_sd)[k[4:]].double()).norm()) for k in dk]),
            "msd_u0_after": S2["discriminators.0.convs.0.weight_u"].numpy(),
            "g_conv_post_v_grad": ref_gg["conv_post.weight_v"].numpy(),
            "g_ups3_v_grad": ref_gg["ups.3.weight_v"].numpy(),
            "sd_checksum": np.array([float(sd[k].double().sum()) for sd in (g_sd, mpd_sd, msd_sd) for k in sorted(sd)]),
        }
        rec["g_grad_samples"], rec["g_grad_sample_off"] = gu.pack_samples(ref_gg, gk, 1024)
        rec["d_grad_samples"], rec["d_grad_sample_off"] = gu.pack_samples(ref_dg, dk, 1024)
        both = dict(ref_dg)
        both.update({"g." + k: v for k, v in ref_gg.items()})
        full = [k for k in FULL_GRADS if k in both]
        assert len(full) == len(FULL_GRADS), set(FULL_GRADS) - set(full)
        rec["grad_full_keys"] = np.array(full)
        for i, k in enumerate(full):
            rec["grad_full_%d" % i] = both[k].numpy()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, ref_out)
