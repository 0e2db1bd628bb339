"""Golden vectors of xVAPitch's generator pass without the adversarial terms: the reference's OWN `xVAPitch.train_step`
(python/xvapitch/model.py:681-870, compiled from its source lines in memory as in gen_golden_xvapitch_acoustic.py) with the reference
HifiganGenerator as `waveform_decoder` (built as model.py:134-149 at the fixture's latent / speaker widths; its 13.7 M-parameter state_dict is
regenerated from a seed by oracle.hifigan.init_vits_decoder_sd), --pitch 1, followed by the loss terms of VitsGeneratorLoss.forward that do not
involve the discriminator (losses.py:187-193 mel L1 x 45 through the reference TorchSTFT, :213-220 KL / duration, :224-241 pitch).

    python oracle/gen_golden_xvapitch_genpass.py       -> tests/golden/xvapitch_genpass.npz

Records the batch, the three random draws (posterior eps, duration-predictor noise, segment starts), the outputs and losses, d(loss)/d(every
parameter) of the acoustic modules and the decoder as norms + evenly spaced samples (a few in full).  Asserts the CPU restatement
(oracle.xvapitch.acoustic_losses + oracle.hifigan.vits_decoder + oracle.mel.mel_m3) equal to the reference run first."""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_xvapitch_acoustic as ga, golden_util as gu, hifigan as ohg, mel as omel, ref_import, xvapitch as oxv  # noqa: E402

DEC_SEED, SEG = 909, 8          # the engine wants segments of >= 2048 samples (8 frames); the model trains on 32


def build():
    """The reference objects and the seeded batch shared with gen_golden_xvapitch_c5.py."""
    ns = ga.load_reference()
    hg = importlib.import_module("python.xvapitch.hifigan")
    xa = importlib.import_module("python.xvapitch.audio")
    c = ga.CFG
    torch.manual_seed(41)
    m = ga.Holder(ns)
    m.train_step = types.MethodType(ns["train_step"], m)
    m._set_cond_input = types.MethodType(ns["_set_cond_input"], m)
    m.args.pitch = 1
    m.spec_segment_size = SEG
    dec = hg.HifiganGenerator(c["latent"], 1, "1", [[1, 3, 5], [1, 3, 5], [1, 3, 5]], [3, 7, 11], [16, 16, 4, 4], 512, [8, 8, 2, 2], inference_padding=0,
                              cond_channels=c["dvec"], conv_pre_weight_norm=False, conv_post_weight_norm=False, conv_post_bias=False)
    dec_sd = ohg.init_vits_decoder_sd(DEC_SEED, c["latent"], c["dvec"])
    dec.load_state_dict(dec_sd)
    m.waveform_decoder = dec
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.startswith("waveform_decoder."):
                continue
            if "gamma" in n or "beta" in n or n.endswith("log_scale") or n.endswith("translation"):
                p += 0.1 * torch.randn_like(p)
            if n.endswith("post.weight") or n.endswith("post.bias") or (".proj." in n and "duration_predictor.flows" in n) or (".proj." in n and "post_flows" in n):
                p += 0.05 * torch.randn_like(p)
    B, Tt, Ty = 3, 19, 50
    x_lens, y_lens = torch.tensor([19, 11, 7]), torch.tensor([50, 37, 21])
    tokens = torch.randint(1, c["vocab"], (B, Tt)) * (torch.arange(Tt)[None, :] < x_lens[:, None])
    y = torch.rand(B, c["spec_bins"], Ty) * (torch.arange(Ty)[None, None, :] < y_lens[:, None, None])
    dvec = torch.randn(B, c["dvec"])
    lids = torch.tensor([0, 3, 1])
    pitch = (torch.rand(B, 1, Ty) * 3 - 1.2).clamp_min(0) * (torch.arange(Ty)[None, None, :] < y_lens[:, None, None])
    wav = torch.from_numpy(np.stack([omel.synth_wave(Ty * 256, 70 + i) for i in range(B)]).astype(np.float32)).unsqueeze(1) * 0.9
    zeros_t = torch.zeros(B, 1, Tt)
    return {"ns": ns, "xa": xa, "m": m, "c": c, "dec_sd": dec_sd, "B": B, "Tt": Tt, "Ty": Ty, "x_lens": x_lens, "y_lens": y_lens, "tokens": tokens, "y": y,
            "dvec": dvec, "lids": lids, "pitch": pitch, "wav": wav, "zeros_t": zeros_t}


def reference_mel(xa, a, b):
    """TorchSTFT mels of two waveforms (losses.py:29-46,187-188) through the reference class."""
    real_stft = torch.stft

    def stft_compat(*args, **k):                                         # torch >= 2 refuses return_complex=False on real input
        k["return_complex"] = True
        return torch.view_as_real(real_stft(*args, **k))
    xa.torch.stft = stft_compat
    try:
        stft = xa.TorchSTFT(1024, 256, 1024, sample_rate=22050, mel_fmin=0, mel_fmax=8000, n_mels=80, use_mel=True, do_amp_to_db=True)
        return stft(a.float()), stft(b.float())
    finally:
        xa.torch.stft = real_stft


def main():
    bd = build()
    ns, xa, m, c, dec_sd = bd["ns"], bd["xa"], bd["m"], bd["c"], bd["dec_sd"]
    B, Tt, Ty, x_lens, y_lens, tokens, y, dvec, lids, pitch, wav, zeros_t = (bd[k] for k in ("B", "Tt", "Ty", "x_lens", "y_lens", "tokens", "y", "dvec", "lids",
                                                                                             "pitch", "wav", "zeros_t"))
    SEED = 123
    torch.manual_seed(SEED)
    out = m.train_step(tokens, x_lens, y, y_lens, pitch, zeros_t, wav, aux_input={"d_vectors": dvec, "language_ids": lids})
    torch.manual_seed(SEED)
    eps = torch.randn(B, c["latent"], Ty)                                # model.py:1472
    noise = torch.randn(B, 2, Tt)                                        # sdp.py:281
    slice_ids = (torch.rand([B]) * (y_lens - SEG + 1)).long()            # util.py:160-162, the step's third draw
    assert torch.equal(ns["segment"](wav, slice_ids * 256, SEG * 256), out["waveform_seg"])
    mel, mel_hat = reference_mel(xa, out["waveform_seg"], out["model_outputs"])
    y_mask = (torch.arange(Ty)[None, :] < y_lens[:, None]).float()
    loss_mel = F.l1_loss(mel, mel_hat, reduction="none").mean() * 45                                             # losses.py:189-193
    loss_kl, _ = ns["kl_loss"](out["z_p"], out["logs_q"], out["m_p"], out["logs_p"], y_mask.unsqueeze(1))
    loss_dur = torch.sum(out["loss_duration"].float())
    lp = F.mse_loss(out["pitch_tgt"], out["pitch_pred"], reduction="none") * out["mask"].unsqueeze(1)
    loss_pitch = lp.sum() / out["mask"].sum() / out["pitch_pred"].shape[0] * 0.1
    loss = loss_mel + loss_kl + loss_dur + loss_pitch
    m.zero_grad()
    loss.backward()
    grads = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("waveform_decoder.")}
    # ---- restatement
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    dl = {k: v.clone().requires_grad_(True) for k, v in dec_sd.items()}
    o = oxv.acoustic_losses(leaves, tokens, x_lens, y, y_lens, dvec, lids, eps, noise, c, pitch_padded=pitch, pe_scaling=m.args.pe_scaling)
    g = F.normalize(dvec).unsqueeze(-1)
    wav_hat = ohg.vits_decoder(dl, oxv.segment(o["z"], slice_ids, SEG), g)
    assert torch.allclose(wav_hat, out["model_outputs"], rtol=1e-4, atol=1e-5), float((wav_hat - out["model_outputs"]).abs().max())
    o_mel = F.l1_loss(omel.mel_m3(out["waveform_seg"].squeeze(1)), omel.mel_m3(wav_hat.squeeze(1)), reduction="none").mean() * 45
    assert abs(float(o_mel.detach()) - float(loss_mel)) < 1e-4 * float(loss_mel), (float(o_mel.detach()), float(loss_mel))
    (o["loss"] + o_mel).backward()
    errs = []
    for n, gr in grads.items():
        lv = dl[n[len("waveform_decoder."):]] if n.startswith("waveform_decoder.") else leaves[n]
        go = lv.grad if lv.grad is not None else torch.zeros_like(gr)
        if float(gr.norm()) < 1e-5 * gr.numel() ** 0.5:
            continue
        errs.append((float((go - gr).norm() / gr.norm()), n))
    errs.sort(reverse=True)
    print("oracle vs reference, worst gradient errors:", errs[:4], "of", len(errs))
    assert errs[0][0] < 1e-2, errs[:4]                                    # LeakyReLU-gate bound of the decoder (gen_golden_vits_decoder.py)
    keys = sorted(grads)
    flat, off = gu.pack_samples(grads, keys, 256)
    res = {"cfg_keys": np.array(sorted(c)), "cfg_vals": np.array([c[k] for k in sorted(c)]), "dec_seed": np.int64(DEC_SEED), "seg": np.int64(SEG),
           "dec_checksum": np.float64(sum(float(v.double().sum()) for v in dec_sd.values())),
           "tokens": tokens.numpy(), "x_lens": x_lens.numpy(), "y": y.numpy(), "y_lens": y_lens.numpy(), "dvec": dvec.numpy(), "lids": lids.numpy(),
           "pitch": pitch.numpy(), "wav": wav.numpy(), "eps": eps.numpy(), "noise": noise.numpy(), "slice_ids": slice_ids.numpy(),
           "model_outputs": out["model_outputs"].detach().numpy(), "z": out["z"].detach().numpy(), "z_p": out["z_p"].detach().numpy(),
           "loss_mel": np.float32(loss_mel.item()), "loss_kl": np.float32(loss_kl.item()), "loss_duration": np.float32(loss_dur.item()),
           "loss_pitch": np.float32(loss_pitch.item()),
           "grad_keys": np.array(keys), "grad_samples": flat, "grad_offsets": off, "grad_norms": np.array([float(grads[k].norm()) for k in keys], dtype=np.float32)}
    for k, v in sd.items():
        res["sd/" + k] = v.numpy()
    for k in ("posterior_encoder.pre.bias", "posterior_encoder.proj.bias", "emb_l.weight", "waveform_decoder.conv_pre.bias", "waveform_decoder.cond_layer.bias",
              "waveform_decoder.conv_post.weight", "flow.flows.0.pre.bias"):
        res["grad/" + k] = grads[k].numpy()
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_genpass.npz")
    np.savez_compressed(path, **res)
    print("xvapitch_genpass.npz: %.2f MB; loss_mel %.4f kl %.4f dur %.4f pitch %.4f; slice_ids %s" % (os.path.getsize(path) / 1e6, loss_mel.item(), loss_kl.item(),
                                                                                                 loss_dur.item(), loss_pitch.item(), slice_ids.tolist()))


if __name__ == "__main__":
    main()
