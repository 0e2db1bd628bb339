"""Parity at BASELINE.json's FULL sizes (configs[1]: FastPitch B = 32 x 150 tokens x 860 frames; configs[2]: HiFi-GAN B = 64 x 8192
samples) through size-independent properties — the CPU oracle pins the same code paths at sizes it finishes in seconds
(test_fastpitch_gpu.py, test_hifigan_gpu.py); here the full-size launches (other tile choices, split-K factors, grid remaps) must
be CONSISTENT with them:
  * batch-composition invariance: an item's outputs do not depend on what else is in the batch (the small batch is the
    oracle-pinned regime);
  * the analytic gradient is the derivative of the loss: central finite difference along a random parameter direction;
  * throughput mode (bf16) vs parity mode (fp32) at the full size stay inside the documented bf16 bound."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _fp_setup(compute, B, Tt, Tm, seed=11):
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import engine as E
    from fp_util import build_engine
    sd = ofp.init_state_dict(3)
    batch = ofp.synth_batch(B, Tt, Tm, seed)
    eng, flat, grads = build_engine(sd, compute)
    return ofp, E, eng, flat, grads, batch


def _sub_batch(batch, idx):
    """Items `idx` of a collated batch, re-padded to their own maxima (TTSCollate layout: sorted by text length desc)."""
    idx = sorted(idx)
    tt = int(batch["in_lens"][idx].max()); tm = int(batch["mel_lens"][idx].max())
    out = {}
    for k, v in batch.items():
        if not torch.is_tensor(v):
            out[k] = v
            continue
        w = v[idx]
        if k in ("text", "durs"):
            w = w[:, :tt]
        elif k in ("mel_tgt", "pitch"):
            w = w[..., :tm]
        elif k == "energy":
            w = w[:, :tm]
        out[k] = w.contiguous()
    return out


def test_fastpitch_full_size_batch_invariance_and_modes():
    ofp, E, eng, flat, grads, batch = _fp_setup("fp32", 32, 150, 860)
    b = E.DeviceBatch.from_dict(batch, "cuda")
    assert (b.B, b.Tt, b.Tm) == (32, 150, 860)
    losses = eng.fwd_loss_bwd(flat, grads, b, 3).clone()
    out = {k: v.clone() for k, v in eng.outputs(b, 3).items() if torch.is_tensor(v)}
    assert torch.equal(out["dec_lens"].cpu().long(), batch["mel_lens"])
    # the same items in a batch of three (the regime the oracle pins): identical predictions up to fp32 summation order
    idx = [0, 13, 31]
    sb = _sub_batch(batch, idx)
    b3 = E.DeviceBatch.from_dict(sb, "cuda")
    eng.forward(flat, b3, 3)
    o3 = eng.outputs(b3, 3)
    for j, i in enumerate(idx):
        tm, tt = int(batch["mel_lens"][i]), int(batch["in_lens"][i])
        a, r = out["mel_out"][i, :tm].float(), o3["mel_out"][j, :tm].float()
        assert ((a - r).abs().max() / r.abs().max()).item() < 2e-5
        for key in ("pitch_pred", "energy_pred"):
            a, r = out[key][i][..., :tt], o3[key][j][..., :tt]
            assert ((a - r).abs().max() / r.abs().max().clamp_min(1e-6)).item() < 2e-5
    # throughput mode at the full size vs the parity mode
    from fp_util import build_engine
    sd = ofp.init_state_dict(3)
    eng16, flat16, grads16 = build_engine(sd, "bf16")
    l16 = eng16.fwd_loss_bwd(flat16, grads16, b, 3)
    mel16 = eng16.outputs(b, 3)["mel_out"].float()
    assert abs(l16[0].item() - losses[0].item()) < 2e-2 * abs(losses[0].item())
    assert ((mel16 - out["mel_out"].float()).norm() / out["mel_out"].float().norm()).item() < 3e-2
    cos = torch.nn.functional.cosine_similarity(grads16.double(), grads.double(), dim=0).item()
    assert cos > 0.99, cos


@pytest.mark.parametrize("stage", [2, 3])
def test_fastpitch_full_size_gradient_is_the_loss_derivative(stage):
    """(L(theta + eps v) - L(theta - eps v)) / (2 eps) = <grad, v> at B = 32 x 860 in the exact-fp32 mode."""
    ofp, E, eng, flat, grads, batch = _fp_setup("fp32", 32, 150, 860, seed=12)
    b = E.DeviceBatch.from_dict(batch, "cuda")
    eng.fwd_loss_bwd(flat, grads, b, stage)
    def loss_at(t):
        eng.forward(t, b, stage)
        eng.loss_partials(b, stage)
        return eng.loss_grads(b, stage)[0].item()
    torch.manual_seed(5)
    # directions with a derivative far above the fp32 noise of the loss: the gradient restricted to a random half of the parameters
    for trial in range(2):
        m = (torch.rand_like(flat) < 0.5).float() if trial else torch.ones_like(flat)
        v = grads * m
        v /= v.norm()
        gv = (grads.double() * v.double()).sum().item()
        eps = 2e-3
        fd = (loss_at(flat + eps * v) - loss_at(flat - eps * v)) / (2 * eps)
        assert abs(fd - gv) < 2e-2 * abs(gv), (trial, fd, gv)


def test_hifigan_full_size_generator_batch_invariance():
    from oracle import hifigan as ohg
    from xva_trainer_amd.hifigan import engine as HE
    g_sd = ohg.init_generator_sd(4321)
    eng = HE.HifiganEngine("cuda", "fp32")
    flat = torch.zeros(eng.total[HE.G], device="cuda")
    HE.to_flat(g_sd, eng.table[HE.G], flat)
    torch.manual_seed(3)
    mel = (torch.randn(64, 80, 32) * 2 - 5).clamp(-11.5, 2.0).cuda()
    wav = eng.generator_forward(flat, mel).clone()
    assert wav.shape == (64, 8192)
    idx = [0, 29, 63]
    w3 = eng.generator_forward(flat, mel[idx].contiguous())
    assert ((wav[idx] - w3).abs().max() / w3.abs().max()).item() < 2e-5
    # bf16 throughput mode at the full size vs the parity mode
    e16 = HE.HifiganEngine("cuda", "bf16")
    w16 = e16.generator_forward(flat, mel)
    assert ((w16 - wav).norm() / wav.norm()).item() < 5e-2


def test_hifigan_full_size_discriminator_gradient_is_the_loss_derivative():
    """d(loss_disc)/d(theta_D) at B = 64 x 8192 against a central finite difference along a random direction (fp32 mode)."""
    from oracle import hifigan as ohg
    from xva_trainer_amd.hifigan import engine as HE
    eng = HE.HifiganEngine("cuda", "fp32")
    flat = torch.zeros(eng.total[HE.D], device="cuda")
    HE.to_flat(ohg.init_mpd_sd(78), eng.table[HE.D], flat, "mpd.")
    HE.to_flat(ohg.init_msd_sd(79), eng.table[HE.D], flat, "msd.")
    torch.manual_seed(9)
    y = (torch.rand(64, 8192, device="cuda") * 2 - 1) * 0.5
    yh = (torch.rand(64, 8192, device="cuda") * 2 - 1) * 0.5
    grads = torch.zeros_like(flat)
    eng.disc_forward(flat.clone(), y, yh)        # (the forward advances the spectral-norm power-iteration buffers held in the parameter buffer)
    eng.disc_backward_d(flat.clone(), grads)
    # direction over the weight-normed discriminators (5 MPD periods, MSD scales 1 and 2); the spectral-normed scale 0 treats its
    # power-iteration vectors as constants in backward (like torch), which a finite difference would not
    keep = torch.zeros_like(flat)
    for name, off, n, shape, kind in eng.table[HE.D]:
        if not name.startswith("msd.discriminators.0."):
            keep[off:off + n] = 1.0
    for trial in range(2):
        m = keep * ((torch.rand_like(flat) < 0.5).float() if trial else 1.0)
        v = grads * m
        v /= v.norm()
        gv = (grads.double() * v.double()).sum().item()
        eps = 2e-3
        lp = eng.disc_forward(flat + eps * v, y, yh)[0].item()
        lm = eng.disc_forward(flat - eps * v, y, yh)[0].item()
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - gv) < 2e-2 * abs(gv), (trial, fd, gv)


def test_hifigan_full_size_bf16_step_is_the_weighted_sum_of_its_items():
    """The BENCHMARKED schedule — bf16, B = 64 x 8192, every engine call of the D + G iteration (generator forward / backward, the 8
    discriminators' forward, D-step backward, G-step backward-data, the mel L1 gradient) — must be consistent with the regime the oracle
    pins: every loss is a batch mean of per-item terms, so with a batch of 22 / 21 / 21 copies of three clips every gradient equals the
    same weighted mean of the three single-clip (B = 1) gradients.  Tile shapes, split-K factors and resident-conv grids differ between B = 1 and
    B = 64, i.e. the fp32 summation order inside a product differs, and an element that rounds to the other bf16 neighbour moves by 4e-3: through ~50
    conv layers that leaves a few 1e-3 of L2 noise (the same effect tests/test_fastpitch_gpu.py quantifies), which is the bound used here."""
    from oracle import hifigan as ohg
    from xva_trainer_amd import mel as pmel
    from xva_trainer_amd.hifigan import engine as HE
    from xva_trainer_amd.hifigan.step import HifiganStep
    st = HifiganStep("cuda", "bf16")
    st.load_state_dicts(ohg.init_generator_sd(1), ohg.init_mpd_sd(2), ohg.init_msd_sd(3))
    u0 = st.flat_d.clone()                                   # the spectral-norm buffers advance with every forward: restore them per run
    x3, y3, ym3 = [t.cuda() for t in ohg.synth_batch(3, 4)]
    counts = [22, 21, 21]
    rep = torch.tensor(sum(([i] * c for i, c in enumerate(counts)), []), device="cuda")

    def run(x, y, ym):
        eng = st.eng
        st.flat_d.copy_(u0)
        yg = eng.generator_forward(st.flat_g, x)
        ld = eng.disc_forward(st.flat_d, y, yg, losses="d")
        gd = torch.zeros_like(st.flat_d)
        eng.disc_backward_d(st.flat_d, gd)
        st.flat_d.copy_(u0)
        lg = eng.disc_forward(st.flat_d, y, yg, losses="g")
        d_wav = eng.disc_backward_g(st.flat_d)
        lm, _ = pmel.mel_l1_loss_backward(yg, ym, d_wav, scale=45.0, accumulate=True)
        gg = torch.zeros_like(st.flat_g)
        eng.generator_backward(st.flat_g, gg, d_wav)
        torch.cuda.synchronize()
        return {"yg": yg.clone(), "ld": ld[0].item(), "lgen": lg[1].item(), "lfm": lg[2].item(), "lmel": lm[0].item(), "gd": gd, "gg": gg,
                "d_wav": d_wav.clone()}

    full = run(x3[rep].contiguous(), y3[rep].contiguous(), ym3[rep].contiguous())
    assert full["yg"].shape == (64, 8192)
    singles = [run(x3[i:i + 1].contiguous(), y3[i:i + 1].contiguous(), ym3[i:i + 1].contiguous()) for i in range(3)]
    w = [c / 64.0 for c in counts]
    for i, c in enumerate(counts):                            # per-item tensors: identical rows whatever the batch
        j = sum(counts[:i])
        assert ((full["yg"][j] - singles[i]["yg"][0]).norm() / singles[i]["yg"][0].norm()).item() < 5e-3
        a, r = full["d_wav"][j] * 64.0, singles[i]["d_wav"][0]
        assert ((a - r).norm() / r.norm()).item() < 5e-3, i
    for k in ("ld", "lgen", "lfm", "lmel"):
        ref = sum(wi * s[k] for wi, s in zip(w, singles))
        assert abs(full[k] - ref) < 1e-3 * abs(ref), (k, full[k], ref)
    nd, ng = st.eng.trainable[HE.D], st.eng.trainable[HE.G]
    for key, n in (("gd", nd), ("gg", ng)):
        ref = sum(wi * s[key][:n].double() for wi, s in zip(w, singles))
        rel = ((full[key][:n].double() - ref).norm() / ref.norm()).item()
        assert rel < 5e-3, (key, rel)
        for which, tbl in ((HE.D, "gd"), (HE.G, "gg")):
            if tbl != key:
                continue
            for b, e in HE.bucket_ranges(which):
                nb = ref[b:e].norm().item()
                assert ((full[key][b:e].double() - ref[b:e]).norm().item() / nb) < 1e-2, (key, b)


@pytest.mark.parametrize("products", ["exact", "split_planes", "f16"])
def test_fastpitch_full_length_against_the_oracle(products):
    """(f16, round 6: the fp16-operand mode — the same schedule on single IEEE-half operand tensors and one-pass v_mfma_f32_16x16x32_f16 products, an fp32 residual
    stream, loss-scaled fp16 gradient buffers: outputs and loss at north_star's 1e-3, the whole gradient at 5e-3.)
    (split_planes, round 5: the same case with fp32 storage and split-bf16 products on the planes path — the feed-forward and attention products of every layer as
    three-pass `planes` launches at the full 862-row key count, the 1 724-row reductions of the weight gradients; outputs and loss at north_star's 1e-3, the whole
    gradient at 2e-3.)
    The CPU oracle AT FULL LENGTH: B = 2 clips of 150 tokens x 860 frames (BASELINE configs[1]'s sequence lengths: 14 key blocks in
    the attention, the 27 584-row GEMM grids' tile shapes per item, 862-row LayerNorm / loss reductions), parity mode (fp32, north_star's
    1e-3): mel / pitch / energy predictions, the loss and every parameter gradient against oracle/fastpitch.py's forward + autograd — the
    full-size tests above are properties only, test_against_oracle_ragged stops at 210 frames."""
    from oracle import fastpitch as ofp
    from fp_util import build_engine, grad_report, rel
    from xva_trainer_amd.fastpitch.engine import DeviceBatch
    torch.manual_seed(0)
    stage = 3
    sd = ofp.init_state_dict(5)
    batch = ofp.synth_batch(2, 150, 860, 21)
    assert int(batch["mel_lens"].max()) == 860 and int(batch["in_lens"].max()) == 150
    names = ofp.trainable_names(sd.keys(), stage)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd); work.update(leaves)
    out_ref = ofp.forward(work, batch, stage)
    loss_ref, _ = ofp.loss(out_ref, batch, stage)
    loss_ref.backward()
    ref_grads = {k: v.grad for k, v in leaves.items() if v.grad is not None}
    from xva_trainer_amd import _lib
    old_mode = _lib.lib.xva_gemm_set_fp32_products(1 if products == "split_planes" else 0)
    try:
        eng, flat, grads = build_engine(sd, "f16" if products == "f16" else "fp32")
        b = DeviceBatch.from_dict(batch, "cuda")
        grads.zero_()
        losses = eng.fwd_loss_bwd(flat, grads, b, stage).cpu()
        out = eng.outputs(b, stage)
        if products == "f16":
            grads.mul_(eng.grad_inv_scale)
    finally:
        _lib.lib.xva_gemm_set_fp32_products(old_mode)
    assert rel(out["mel_out"], out_ref[0]) < 1e-3
    assert rel(out["pitch_pred"], out_ref[4]) < 1e-3
    assert rel(out["energy_pred"], out_ref[6]) < 1e-3
    assert torch.equal(out["dec_lens"].cpu().long(), batch["mel_lens"])
    assert abs(losses[0].item() - loss_ref.item()) < 1e-3 * abs(loss_ref.item())
    if products == "f16":
        from xva_trainer_amd.fastpitch import params as P
        mine = P.from_flat(grads, eng.table)
        a = torch.cat([mine[k].double().cpu().flatten() for k in ref_grads])
        r = torch.cat([ref_grads[k].double().flatten() for k in ref_grads])
        assert ((a - r).norm() / r.norm()).item() < 5e-3
        bad, worst = grad_report(eng, grads, ref_grads, 4e-2)
        assert not bad, bad[:10]
        return
    if products == "split_planes":        # products carry ~1e-5 each: more gates within rounding of zero than in the exact mode (tests/test_fastpitch_gpu.py: 6e-3 on elements)
        from xva_trainer_amd.fastpitch import params as P
        mine = P.from_flat(grads, eng.table)
        a = torch.cat([mine[k].double().cpu().flatten() for k in ref_grads])
        r = torch.cat([ref_grads[k].double().flatten() for k in ref_grads])
        assert ((a - r).norm() / r.norm()).item() < 2e-3
        bad, worst = grad_report(eng, grads, ref_grads, 6e-3)
        assert all(rr < 3e-2 for _, rr in bad) and len(bad) <= 0.05 * len(ref_grads), bad[:10]
        return
    # Gradients: 2e-3 per tensor as at the small sizes — except where a ReLU gate sits within rounding of zero.  With these seeds ONE
    # element of the energy predictor's first ConvReLUNorm has an fp32 pre-activation of -8e-8 in the oracle and +eps here (measured: the
    # only element of that tensor differing by more than 1e-3 of its maximum; tools/fp_grad_report.py 3 fp32 2,150,860): its gate is open
    # on one side and closed on the other, which moves that predictor's layer-0 gradients and, through d(encoder output), the encoder's
    # small-norm tensors by up to 5e-3 — a property of ReLU at 0, not of either implementation (the fp64 oracle shares the fp32 oracle's
    # side by luck of rounding).  So: every tensor within 1e-2, at least 95 % of them within 2e-3, and the whole gradient within 1e-3.
    bad, worst = grad_report(eng, grads, ref_grads, 2e-3)
    assert all(r < 1e-2 for _, r in bad), bad[:10]
    assert len(bad) <= 0.05 * len(ref_grads), bad[:10]
    from xva_trainer_amd.fastpitch import params as P
    mine = P.from_flat(grads, eng.table)
    a = torch.cat([mine[k].double().cpu().flatten() for k in ref_grads])
    r = torch.cat([ref_grads[k].double().flatten() for k in ref_grads])
    assert ((a - r).norm() / r.norm()).item() < 1e-3
