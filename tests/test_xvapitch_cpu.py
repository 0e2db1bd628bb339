"""CPU: oracle/xvapitch.py against the vectors recorded from the reference's xVAPitch modules (tests/golden/xvapitch_blocks.npz,
written by oracle/gen_golden_xvapitch.py): WN with conditioning, the mean-only ResidualCouplingBlock (forward, reverse), maximum_path,
segment, kl_loss — values and autograd gradients."""
import os

import numpy as np
import pytest
import torch


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "xvapitch_blocks.npz"))


def _sd(g, pre):
    return {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


def _mask(lens, T):
    return (torch.arange(T)[None, :] < torch.as_tensor(lens)[:, None]).float().unsqueeze(1)


def test_wn_oracle_matches_reference(golden_dir):
    from oracle import xvapitch as oxv
    g = _g(golden_dir)
    B, H, T, CIN, L, K = [int(v) for v in g["wn_cfg"]]
    sd = {k: v.clone().requires_grad_(True) for k, v in _sd(g, "wn_sd/").items()}
    x = torch.from_numpy(g["wn_x"]).requires_grad_(True)
    cond = torch.from_numpy(g["wn_g"]).requires_grad_(True)
    y = oxv.wn(sd, x, _mask(g["wn_lens"], T), cond, hidden=H, kernel_size=K, dilation_rate=1, num_layers=L)
    assert torch.allclose(y, torch.from_numpy(g["wn_y"]), rtol=1e-5, atol=1e-6)
    (y * torch.from_numpy(g["wn_r"])).sum().backward()
    assert torch.allclose(x.grad, torch.from_numpy(g["wn_dx"]), rtol=1e-4, atol=1e-5)
    assert torch.allclose(cond.grad, torch.from_numpy(g["wn_dg"]), rtol=1e-4, atol=1e-5)
    for k, ref in _sd(g, "wn_grad/").items():
        assert torch.allclose(sd[k].grad, ref, rtol=1e-3, atol=1e-5), k


def test_coupling_oracle_matches_reference(golden_dir):
    from oracle import xvapitch as oxv
    g = _g(golden_dir)
    B, CH, H, T, L, K = [int(v) for v in g["cp_cfg"]]
    sd = _sd(g, "cp_sd/")
    x = torch.from_numpy(g["cp_x"])
    m = _mask(g["wn_lens"], T)
    kw = dict(hidden=H, kernel_size=K, dilation_rate=1, num_layers=L)
    assert torch.allclose(oxv.coupling(sd, x, m, **kw), torch.from_numpy(g["cp_y"]), rtol=1e-5, atol=1e-6)
    rev = oxv.coupling(sd, x, m, reverse=True, **kw)
    assert torch.allclose(rev, torch.from_numpy(g["cp_yrev"]), rtol=1e-5, atol=1e-6)
    # the flow is invertible on the live positions: reverse(forward(x)) == x there (x1 is masked by both directions)
    back = oxv.coupling(sd, oxv.coupling(sd, x, m, **kw), m, reverse=True, **kw)
    half = CH // 2
    assert torch.allclose(back[:, half:], x[:, half:] * m, rtol=1e-4, atol=1e-5) and torch.equal(back[:, :half], x[:, :half])


def test_maximum_path_segment_kl_oracles(golden_dir):
    from oracle import xvapitch as oxv
    g = _g(golden_dir)
    path = oxv.maximum_path(g["mp_value"], g["mp_mask"])
    assert np.array_equal(path, g["mp_path"])
    xl, yl = g["mp_mask"][:, :, 0].sum(1), g["mp_mask"][:, 0, :].sum(1)
    for b in range(path.shape[0]):                       # a monotonic path: one text position per live frame, never moving back, ending on the last symbol
        if xl[b] > yl[b]:
            continue                                     # more symbols than frames (item 2: 12 x 12 is fine, item 3: 1 symbol): the DP cannot cover them all
        cols = path[b, :, :int(yl[b])]
        assert np.all(cols.sum(0) == 1)
        pos = cols.argmax(0)
        assert np.all(np.diff(pos) >= 0) and np.all(np.diff(pos) <= 1) and pos[-1] == xl[b] - 1
    assert torch.equal(oxv.segment(torch.from_numpy(g["sg_x"]), g["sg_idx"], 4), torch.from_numpy(g["sg_out"]))
    t4 = [torch.from_numpy(a).requires_grad_(True) for a in g["kl_in"]]
    l, sw = oxv.kl_loss(*t4, torch.from_numpy(g["kl_mask"]))
    assert abs(l.item() - float(g["kl_loss"])) < 1e-5 and torch.allclose(sw, torch.from_numpy(g["kl_sw"]), rtol=1e-5, atol=1e-6)
    (l * 1.7).backward()
    for t, ref in zip(t4, g["kl_grads"]):
        assert torch.allclose(t.grad, torch.from_numpy(ref), rtol=1e-5, atol=1e-7)


def test_posterior_encoder_oracle_matches_reference(golden_dir):
    from oracle import xvapitch as oxv
    g = _g(golden_dir)
    B, CSP, CO, H, T, L, K, CIN = [int(v) for v in g["pe_cfg"]]
    z, mean, logs, _ = oxv.posterior_encoder(_sd(g, "pe_sd/"), torch.from_numpy(g["pe_x"]), torch.from_numpy(g["wn_lens"]), torch.from_numpy(g["pe_g"]),
                                            torch.from_numpy(g["pe_eps"]), CO, hidden=H, kernel_size=K, dilation_rate=1, num_layers=L)
    for a, k in ((z, "pe_z"), (mean, "pe_mean"), (logs, "pe_logs")):
        assert torch.allclose(a, torch.from_numpy(g[k]), rtol=1e-5, atol=1e-6), k


def test_rel_transformer_oracle_matches_reference_golden():
    """oracle/xvapitch.py:rel_transformer vs the vectors recorded from the reference RelativePositionTransformer (glow_tts.py:373-485)."""
    import os
    import numpy as np
    import torch
    from oracle import xvapitch as oxv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xvapitch_transformer.npz"))
    B, Cc, Fh, H, L, K, W, T = (int(v) for v in g["cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    sd = {k[3:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("sd/")}
    x = torch.from_numpy(g["x"]).requires_grad_(True)
    y = oxv.rel_transformer(sd, x, x_mask, H, L, K, W)
    assert torch.allclose(y, torch.from_numpy(g["y"]), rtol=1e-5, atol=1e-5)
    (y * torch.from_numpy(g["r"])).sum().backward()
    assert torch.allclose(x.grad, torch.from_numpy(g["dx"]), rtol=1e-4, atol=1e-5)
    for k in g.files:
        if k.startswith("grad/"):
            assert torch.allclose(sd[k[5:]].grad, torch.from_numpy(g[k]), rtol=1e-4, atol=1e-5), k


def test_dds_conv_oracle_matches_reference_golden():
    """oracle/xvapitch.py:dds_conv vs the vectors recorded from the reference DilatedDepthSeparableConv (sdp.py:40-93)."""
    import os
    import numpy as np
    import torch
    from oracle import xvapitch as oxv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xvapitch_sdp.npz"))
    B, Cc, T, K, L = (int(v) for v in g["dds_cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    sd = {k[7:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("dds_sd/")}
    x = torch.from_numpy(g["dds_x"]).requires_grad_(True); gg = torch.from_numpy(g["dds_g"]).requires_grad_(True)
    y = oxv.dds_conv(sd, x, x_mask, gg, K, L)
    assert torch.allclose(y, torch.from_numpy(g["dds_y"]), rtol=1e-5, atol=1e-6)
    (y * torch.from_numpy(g["dds_r"])).sum().backward()
    assert torch.allclose(x.grad, torch.from_numpy(g["dds_dx"]), rtol=1e-4, atol=1e-5) and torch.allclose(gg.grad, torch.from_numpy(g["dds_dg"]), rtol=1e-4, atol=1e-5)
    for k in g.files:
        if k.startswith("dds_grad/"):
            assert torch.allclose(sd[k[9:]].grad, torch.from_numpy(g[k]), rtol=1e-4, atol=1e-5), k


def test_conv_flow_oracle_matches_reference_golden():
    """oracle/xvapitch.py:conv_flow + rq_spline vs the vectors recorded from the reference ConvFlow (sdp.py:116-176, util.py:203-391)."""
    import os
    import numpy as np
    import torch
    from oracle import xvapitch as oxv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xvapitch_sdp.npz"))
    B, Hh, T, K, L, NB = (int(v) for v in g["cf_cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    sd = {k[6:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("cf_sd/")}
    z = torch.from_numpy(g["cf_z"]).requires_grad_(True); cond = torch.from_numpy(g["cf_g"]).requires_grad_(True)
    y, ld = oxv.conv_flow(sd, z, x_mask, cond, Hh, K, L, NB)
    assert torch.allclose(y, torch.from_numpy(g["cf_y"]), rtol=1e-5, atol=1e-5) and torch.allclose(ld, torch.from_numpy(g["cf_logdet"]), rtol=1e-5, atol=1e-4)
    ((y * torch.from_numpy(g["cf_rz"])).sum() + (ld * torch.from_numpy(g["cf_rl"])).sum()).backward()
    assert torch.allclose(z.grad, torch.from_numpy(g["cf_dz"]), rtol=1e-4, atol=1e-5)
    for k in g.files:
        if k.startswith("cf_grad/"):
            assert torch.allclose(sd[k[8:]].grad, torch.from_numpy(g[k]), rtol=1e-4, atol=1e-4), k


def test_sdp_oracle_matches_reference_golden():
    """oracle/xvapitch.py:sdp_forward vs the vectors recorded from the reference StochasticDurationPredictor (sdp.py:179-310, training direction)."""
    import os
    import numpy as np
    import torch
    from oracle import xvapitch as oxv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xvapitch_sdp.npz"))
    B, Cin, Hs, Cg, Cl, T, K = (int(v) for v in g["sdp_cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    sd = {k[7:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("sdp_sd/")}
    x = torch.from_numpy(g["sdp_x"]).requires_grad_(True)
    nll = oxv.sdp_forward(sd, x, x_mask, torch.from_numpy(g["sdp_dr"]), torch.from_numpy(g["sdp_noise"]), Hs, K, 4, g=torch.from_numpy(g["sdp_g"]),
                          lang_emb=torch.from_numpy(g["sdp_lang"]))
    assert torch.allclose(nll, torch.from_numpy(g["sdp_nll"]), rtol=1e-5, atol=1e-3)
    (nll * torch.from_numpy(g["sdp_r"])).sum().backward()
    ref = torch.from_numpy(g["sdp_dx"])
    assert float((x.grad - ref).norm() / ref.norm()) < 2e-4
    for k in g.files:
        if k.startswith("sdp_grad/"):
            r = torch.from_numpy(g[k])
            assert float((sd[k[9:]].grad - r).norm() / r.norm().clamp_min(1e-12)) < 2e-4, k


@pytest.mark.parametrize("tag", ["", "p_"])
def test_acoustic_losses_oracle_matches_reference_train_step_golden(tag):
    """oracle/xvapitch.py:acoustic_losses vs the vectors recorded from the reference's own xVAPitch.train_step (model.py:681-870; the generator
    oracle/gen_golden_xvapitch_acoustic.py compiled the method from its source lines in memory and ran it) — with --pitch 0 and with --pitch 1
    (tag p_): outputs, MAS path, the losses and the parameter gradients (first run: all in full; second: norms + samples, pitch tensors in full)."""
    from oracle import golden_util, xvapitch as oxv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xvapitch_acoustic.npz"))
    cfg = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
    t = lambda k: torch.from_numpy(g[k])
    o = oxv.acoustic_losses(leaves, t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("dvec"), t("lids"), t(tag + "eps"), t(tag + "noise"), cfg,
                            pitch_padded=t("pitch") if tag else None)
    for k in [f[len(tag) + 4:] for f in g.files if f.startswith(tag + "out/")]:
        assert torch.allclose(o[k].detach(), t(tag + "out/" + k), rtol=1e-4, atol=1e-5), k
    assert np.array_equal(o["attn"].numpy().astype(np.uint8), g[tag + "attn"])
    for k in ("loss_kl", "loss_duration") + (("loss_pitch",) if tag else ()):
        assert abs(float(o[k].detach()) - float(g[tag + k])) < 1e-3, k
    o["loss"].backward()
    mine = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items() if v.requires_grad}
    n = 0
    for k in g.files:
        if k.startswith(tag + "grad/"):
            ref = t(k)
            got = mine[k[len(tag) + 5:]]
            if float(ref.norm()) < 1e-5 * ref.numel() ** 0.5:                      # mathematically zero (conv_k.bias, an unused branch)
                assert float(got.norm()) < 1e-4, k
                continue
            assert float((got - ref).norm() / ref.norm()) < 2e-4, k
            n += 1
    assert n > (30 if tag else 400)
    if tag:
        keys = [str(k) for k in g["p_grad_keys"]]
        live = [k for k, nr in zip(keys, g["p_grad_norms"]) if nr >= 1e-5 * mine[k].numel() ** 0.5]
        errs = [e for e in golden_util.check_samples(mine, keys, g["p_grad_samples"], g["p_grad_offsets"], 256) if e[1] in live]
        assert len(errs) > 450 and errs[0][0] < 2e-4, errs[:4]
        for k, nr in zip(keys, g["p_grad_norms"]):
            assert abs(float(mine[k].norm()) - float(nr)) < 2e-4 * float(nr) + 1e-5, k


def test_vits_decoder_oracle_matches_reference_golden():
    """oracle/hifigan.py:vits_decoder vs the vectors recorded from the reference HifiganGenerator (python/xvapitch/hifigan.py:156-262 built as
    model.py:134-149): waveform 1e-5, gradients at the fixture's LeakyReLU-gate bound (1e-2, see oracle/gen_golden_vits_decoder.py)."""
    from oracle import golden_util, hifigan as ohg
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vits_decoder.npz"))
    seed, B, Cin, Cc, T = (int(v) for v in g["cfg"])
    sd = ohg.init_vits_decoder_sd(seed, Cin, Cc)
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["sd_checksum"])) < 1e-3
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    z = torch.from_numpy(g["z"]).requires_grad_(True)
    y = ohg.vits_decoder(leaves, z, torch.from_numpy(g["g"]).unsqueeze(-1))
    assert torch.allclose(y.detach(), torch.from_numpy(g["y"]), rtol=1e-5, atol=1e-6)
    (y * torch.from_numpy(g["r"])).sum().backward()
    mine = {k: v.grad for k, v in leaves.items()}
    errs = golden_util.check_samples(mine, [str(k) for k in g["grad_keys"]], g["grad_samples"], g["grad_offsets"], 512)
    assert len(errs) == 233 and errs[0][0] < 1e-2, errs[:4]
    assert float((z.grad - torch.from_numpy(g["dz"])).norm() / torch.from_numpy(g["dz"]).norm()) < 1e-2


def test_generator_pass_oracle_matches_reference_train_step_golden():
    """The CPU restatement of the generator pass (oracle.xvapitch.acoustic_losses + oracle.hifigan.vits_decoder + oracle.mel.mel_m3) vs the vectors
    recorded from the reference's own train_step with the reference HifiganGenerator as decoder (oracle/gen_golden_xvapitch_genpass.py)."""
    import torch.nn.functional as F
    from oracle import golden_util, hifigan as ohg, mel as omel, xvapitch as oxv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xvapitch_genpass.npz"))
    cfg = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    t = lambda k: torch.from_numpy(g[k])
    leaves = {k[3:]: t(k).clone().requires_grad_(t(k).is_floating_point()) for k in g.files if k.startswith("sd/")}
    dl = {k: v.requires_grad_(True) for k, v in ohg.init_vits_decoder_sd(int(g["dec_seed"]), cfg["latent"], cfg["dvec"]).items()}
    o = oxv.acoustic_losses(leaves, t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("dvec"), t("lids"), t("eps"), t("noise"), cfg, pitch_padded=t("pitch"))
    S = int(g["seg"])
    wav_hat = ohg.vits_decoder(dl, oxv.segment(o["z"], t("slice_ids"), S), F.normalize(t("dvec")).unsqueeze(-1))
    assert torch.allclose(wav_hat.detach(), t("model_outputs"), rtol=1e-4, atol=1e-5)
    seg = oxv.segment(t("wav"), t("slice_ids") * 256, S * 256)
    loss_mel = F.l1_loss(omel.mel_m3(seg.squeeze(1)), omel.mel_m3(wav_hat.squeeze(1)), reduction="none").mean() * 45
    for k, v in (("loss_mel", loss_mel), ("loss_kl", o["loss_kl"]), ("loss_duration", o["loss_duration"]), ("loss_pitch", o["loss_pitch"])):
        assert abs(float(v.detach()) - float(g[k])) < 1e-3 * abs(float(g[k])), k
    (o["loss"] + loss_mel).backward()
    mine = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items() if v.requires_grad}
    mine.update({"waveform_decoder." + k: v.grad for k, v in dl.items()})
    keys = [str(k) for k in g["grad_keys"]]
    live = set(k for k, nr in zip(keys, g["grad_norms"]) if nr >= 1e-5 * mine[k].numel() ** 0.5)
    errs = [e for e in golden_util.check_samples(mine, keys, g["grad_samples"], g["grad_offsets"], 256) if e[1] in live]
    assert len(errs) > 690 and errs[0][0] < 1e-2, errs[:4]


def test_vits_discriminator_oracle_matches_reference_golden():
    """oracle/hifigan.py:vits_disc + the loss restatements vs the vectors recorded from the reference VitsDiscriminator / loss functions."""
    from oracle import golden_util, hifigan as ohg
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vits_disc.npz"))
    seed, B, seg = (int(v) for v in g["cfg"])
    leaves = {k: v.requires_grad_(True) for k, v in ohg.init_vits_disc_sd(seed).items()}
    y = torch.from_numpy(g["y"]).unsqueeze(1)
    yh = torch.from_numpy(g["y_hat"]).unsqueeze(1).requires_grad_(True)
    rs, fr, gs, fg = ohg.vits_disc(leaves, y, yh.detach())
    ld = ohg.discriminator_loss(rs, gs)
    assert abs(float(ld.detach()) - float(g["loss_disc"])) < 1e-4 * float(g["loss_disc"])
    ld.backward()
    errs = golden_util.check_samples({k: v.grad for k, v in leaves.items()}, [str(k) for k in g["grad_keys"]], g["grad_samples"], g["grad_offsets"], 512)
    assert len(errs) == 111 and errs[0][0] < 1e-4, errs[:4]
    rs, fr, gs, fg = ohg.vits_disc({k: v.detach() for k, v in leaves.items()}, y, yh)
    lg, lf = ohg.generator_loss(gs), ohg.feature_loss([[t.detach() for t in f] for f in fr], fg)
    assert abs(float(lg.detach()) - float(g["loss_gen"])) < 1e-4 * float(g["loss_gen"]) and abs(float(lf.detach()) - float(g["loss_feat"])) < 1e-4 * float(g["loss_feat"])
    (lg + lf).backward()
    ref = torch.from_numpy(g["d_wav"]).unsqueeze(1)
    assert float((yh.grad - ref).norm() / ref.norm()) < 1e-3


def test_oracle_dropout_sites_against_reference_train_mode_golden():
    """tests/golden/xvapitch_dropout.npz was recorded from the REFERENCE modules in train mode with nn.Dropout.forward replaced by the
    keyed-hash mask of the call's site over the tensor's flat order (oracle/gen_golden_xvapitch_dropout.py).  The oracle with the same hook at
    its sites — attention weights, attention output, feed-forward hidden, feed-forward output per transformer layer (glow_tts.py:204,473,344,477);
    after the second GELU of each layer of the duration predictor's `convs` / `post_convs` (sdp.py:90,227,237) — must reproduce outputs and
    gradients: this pins WHERE the oracle (and through it the HIP path, tests/test_xvapitch_gpu.py) drops to where the reference does."""
    import os
    import numpy as np
    import torch
    from oracle import xvapitch as oxv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xvapitch_dropout.npz"))
    seed = int(g["seed"][0])
    lens = torch.from_numpy(g["lens"])
    for tag in ("te", "pp"):
        B, Cc, Co, Fh, H, L, K, W, T = (int(v) for v in g[tag + "_cfg"])
        x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
        sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith(tag + "_sd/")}
        x = torch.from_numpy(g[tag + "_x"]).requires_grad_(True)
        hook = oxv.HashDrop(float(g[tag + "_p"][0]), seed, layout="flat")
        y = oxv.rel_transformer(sd, x, x_mask, H, L, K, W, drop=hook)
        assert torch.allclose(y, torch.from_numpy(g[tag + "_y"]), rtol=1e-5, atol=1e-5)
        y0 = oxv.rel_transformer(sd, x, x_mask, H, L, K, W)
        assert float((y0 - y).abs().max()) > 1e-2                                     # and it is not the eval-mode output
        (y * torch.from_numpy(g[tag + "_r"])).sum().backward()
        assert torch.allclose(x.grad, torch.from_numpy(g[tag + "_dx"]), rtol=1e-4, atol=1e-5)
        names = [k[len(tag) + 6:] for k in g.files if k.startswith(tag + "_grad/")]
        assert len(names) >= 18 * L - 6
        for n in names:
            assert torch.allclose(sd[n].grad, torch.from_numpy(g["%s_grad/%s" % (tag, n)]), rtol=1e-4, atol=1e-5), (tag, n)
    B, Cin, Hh, Cg, Cl, Ts = (int(v) for v in g["sdp_cfg"])
    lens2 = torch.from_numpy(g["sdp_lens"])
    m_ = (torch.arange(Ts)[None, :] < lens2[:, None]).float().unsqueeze(1)
    sd = {k[7:]: torch.from_numpy(g[k]).requires_grad_(True) for k in g.files if k.startswith("sdp_sd/")}
    x = torch.from_numpy(g["sdp_x"]).requires_grad_(True)
    t = lambda k: torch.from_numpy(g[k])
    nll = oxv.sdp_forward(sd, x, m_, t("sdp_dr"), t("sdp_noise"), Hh, 3, 4, g=t("sdp_g"), lang_emb=t("sdp_le"), drop=oxv.HashDrop(0.5, seed + 1, layout="flat"))
    assert torch.allclose(nll, t("sdp_nll"), rtol=1e-4, atol=1e-3)
    nll.sum().backward()
    assert torch.allclose(x.grad, t("sdp_dx"), rtol=1e-3, atol=1e-4)
    for k in g.files:
        if k.startswith("sdp_grad/"):
            assert torch.allclose(sd[k[9:]].grad, t(k), rtol=1e-3, atol=1e-4), k


def test_infer_oracle_matches_reference_infer_golden():
    """oracle/xvapitch.py:infer (text encoder, duration predictor in reverse — inverse splines —, path expansion, pitch branch, flow in reverse,
    waveform decoder) and rq_spline_inverse vs the vectors recorded from the reference's own xVAPitch.infer and
    piecewise_rational_quadratic_transform(inverse=True) (oracle/gen_golden_xvapitch_infer.py): durations exact, waveform 1e-4."""
    from oracle import hifigan as ohg, xvapitch as oxv
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "xvapitch_infer.npz"))
    c = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    K = 10
    y, h = torch.from_numpy(g["spline/y"]), torch.from_numpy(g["spline/h"])
    ws = float(g["spline/wh_scale"])
    x = oxv.rq_spline_inverse(y, h[:, :K] * ws, h[:, K:2 * K] * ws, h[:, 2 * K:], float(g["spline/bound"]))
    assert torch.allclose(x, torch.from_numpy(g["spline/x"]), rtol=1e-5, atol=1e-5)
    fwd, _ = oxv.rq_spline(x, h[:, :K] * ws, h[:, K:2 * K] * ws, h[:, 2 * K:], float(g["spline/bound"]))
    assert torch.allclose(fwd, y, atol=1e-4)                                       # the forward map undoes it
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    dl = ohg.init_vits_decoder_sd(int(g["dec_seed"]), c["latent"], c["dvec"])
    for case in (1, 2):                                                            # case 0 (40 frames) is the GPU test's; these keep the CPU suite short
        pre = "c%d/" % case
        with torch.no_grad():
            o = oxv.infer(sd, torch.from_numpy(g[pre + "tokens"]), torch.from_numpy(g[pre + "dvec"]), torch.from_numpy(g[pre + "lid"]),
                          torch.from_numpy(g[pre + "noise"]), c, lambda z, gg: ohg.vits_decoder(dl, z, gg), pacing=float(g[pre + "pacing"]))
        assert torch.equal(o["w_ceil"], torch.from_numpy(g[pre + "w_ceil"]))
        assert torch.allclose(o["wav"], torch.from_numpy(g[pre + "wav"]), rtol=1e-4, atol=2e-5)
