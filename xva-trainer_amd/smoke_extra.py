"""Second half of __graft_entry__.smoke(): one tiny FastPitch train step and one tiny HiFi-GAN D+G iteration on cuda:0, each checked
against the CPU oracle (oracle/ is test infrastructure: imported here only as the checker)."""
import torch


def run():
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import engine as E, params as P
    sd = ofp.init_state_dict(3)
    batch = ofp.synth_batch(2, 9, 30, 4)
    out_ref = ofp.forward(sd, batch, 3)
    loss_ref, _ = ofp.loss(out_ref, batch, 3)
    for compute, tol in (("fp32", 1e-3), ("bf16", 5e-2)):
        eng = E.FastPitchEngine("cuda:0", compute)
        flat = torch.zeros(eng.total, device="cuda:0")
        P.to_flat(sd, eng.table, flat)
        grads = torch.zeros_like(flat)
        b = E.DeviceBatch.from_dict(batch, "cuda:0")
        losses = eng.fwd_loss_bwd(flat, grads, b, 3).cpu()
        mel = eng.outputs(b, 3)["mel_out"].float().cpu()
        err = ((mel - out_ref[0]).abs().max() / out_ref[0].abs().max()).item()
        lerr = abs(losses[0].item() - loss_ref.item()) / abs(loss_ref.item())
        assert err < tol and lerr < tol, "FastPitch %s parity failed: mel %g loss %g" % (compute, err, lerr)
        assert torch.isfinite(grads).all() and grads.abs().max().item() > 0
        print("smoke: FastPitch %s step vs oracle: mel rel err %.3g, loss rel err %.3g" % (compute, err, lerr))
    # HiFi-GAN: generator forward vs the oracle on one short segment, then one full D+G iteration must stay finite
    from oracle import hifigan as ohg
    from xva_trainer_amd.hifigan.step import HifiganStep
    st = HifiganStep("cuda:0", "fp32")
    g_sd = ohg.init_generator_sd(1)
    st.load_state_dicts(g_sd, ohg.init_mpd_sd(2), ohg.init_msd_sd(3))
    x, y, ym = ohg.synth_batch(1, 4)
    wav = st.eng.generator_forward(st.flat_g, x.cuda()).cpu()
    with torch.no_grad():
        ref = ohg.generator(g_sd, x).reshape(wav.shape)
    err = ((wav - ref).abs().max() / ref.abs().max()).item()
    assert err < 1e-3, "HiFi-GAN generator parity failed: %g" % err
    out = st.train_step(x.cuda(), y.cuda(), ym.cuda())
    assert torch.isfinite(out["loss_mel"]).item() and torch.isfinite(out["loss_disc_all"]).item()
    print("smoke: HiFi-GAN generator vs oracle rel err %.3g; D+G iteration loss_mel %.4f loss_disc_all %.4f"
          % (err, out["loss_mel"].item(), out["loss_disc_all"].item()))
