"""Run-to-run reproducibility of the engines on their default stream lanes (full-size BASELINE shapes).  Stored activations and outputs must be bit-identical between
repeated runs on the same inputs (same dropout seed); sums that end in fp32 atomics (losses, bias / LayerNorm gradients) may differ in the last bits — reported
as relative differences.  A sporadic, larger difference is a race (round 5: the packed-fp32 conv0 kernel of the discriminators was found this way).

    python tools/determinism_sweep.py [runs=6]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xva_trainer_amd import synthetic
from xva_trainer_amd.fastpitch import engine as E, params as P

R = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = "cuda:0"


def nrel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def report(name, runs):
    """runs: list of dicts name -> tensor"""
    worst = {}
    for r in runs[1:]:
        for k, v in r.items():
            d = 0.0 if torch.equal(v, runs[0][k]) else max(nrel(v, runs[0][k]), 1e-30)
            worst[k] = max(worst.get(k, 0.0), d)
    exact = [k for k, d in worst.items() if d == 0.0]
    other = sorted(((d, k) for k, d in worst.items() if d > 0.0), reverse=True)
    print("%s: %d runs, %d tensors bit-identical; differing: %s" % (name, len(runs), len(exact), ", ".join("%s %.2e" % (k, d) for d, k in other[:12]) or "none"))


# ---- FastPitch, every compute mode
from xva_trainer_amd import _lib
for compute, products in (("bf16", 0), ("fp32", 0), ("fp32", 1)):          # products 1: fp32 storage, split-bf16 products on planes (the mode that meets 1e-3 at speed)
    old_products = _lib.lib.xva_gemm_set_fp32_products(products)
    eng = E.FastPitchEngine(dev, compute, p_dropout=0.1, seed=1234)
    flat = torch.zeros(eng.total, device=dev)
    P.default_init_(flat, eng.table, seed=1234)
    batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(32, 150, 860, 1234), dev)
    runs = []
    for r in range(R):
        eng.step = 0                                        # the same dropout masks every run
        grads = torch.zeros_like(flat)
        losses = eng.fwd_loss_bwd(flat, grads, batch, 3)
        o = eng.outputs(batch, 3)
        torch.cuda.synchronize()
        d = {k: v.float().clone() for k, v in o.items() if torch.is_tensor(v)}
        d["losses"] = losses.clone()
        for name, off, numel, shape in [(t[0], t[1], t[2], t[3]) for t in eng.table][:400]:
            d["grad:" + name] = grads[off:off + numel].clone()
        runs.append(d)
    report("FastPitch %s%s step (B = 32 x 150 x 860, stage 3, dropout 0.1)" % (compute, " + split-bf16 products" if products else ""), runs)
    _lib.lib.xva_gemm_set_fp32_products(old_products)
    del eng, flat, batch, runs
    torch.cuda.empty_cache()

# ---- HiFi-GAN: generator forward slots, waveform, D + G gradients
from xva_trainer_amd.hifigan.step import HifiganStep
from xva_trainer_amd.hifigan import engine as HE
from xva_trainer_amd import mel as pmel
for hg_mode, hg_B in (("bf16", 64), ("fp32", 16)):
    st = HifiganStep(dev, hg_mode)
    bench.init_hifigan_weights(st)
    x, y, y_mel = bench.hifigan_inputs(hg_B, 0, dev)
    eng = st.eng
    runs = []
    pd0 = st.flat_d.clone()
    for r in range(R):
        st.flat_d.copy_(pd0)                                    # the spectral-norm buffers advance every pass: start every run from the same ones
        d = {}
        yg = eng.generator_forward(st.flat_g, x)
        d["waveform"] = yg.clone()
        for rb in range(12):
            for m in range(3):
                d["xt1.%d.%d" % (rb, m)] = HE._slot(eng, "xt1", rb, m).float().clone()
        for i in range(4):
            d["xs.%d" % i] = HE._slot(eng, "xs", i).float().clone()
        ld = eng.disc_forward(st.flat_d, y, yg, losses="d")
        for dd in range(5):
            for i in range(1, 7):
                d["mpd.%d.%d" % (dd, i)] = HE._slot(eng, "mpd", dd, i).float().clone()
        for sc in range(3):
            for i in range(1, 8):
                d["msd.%d.%d" % (sc, i)] = HE._slot(eng, "msd", sc, 0, i).float().clone()
        gd = torch.zeros_like(st.flat_d)
        eng.disc_backward_d(st.flat_d, gd)
        d["loss_d"] = ld.clone(); d["grads_d"] = gd.clone()
        lg = eng.disc_forward(st.flat_d, y, yg, losses="g")
        dw = eng.disc_backward_g(st.flat_d)
        d["d_wav (G step)"] = dw.clone(); d["loss_g"] = lg.clone()
        pmel.mel_l1_loss_backward(yg, y_mel, dw, scale=45.0, accumulate=True)
        gg = torch.zeros_like(st.flat_g)
        eng.generator_backward(st.flat_g, gg, dw)
        d["grads_g"] = gg.clone()
        torch.cuda.synchronize()
        runs.append(d)
    report("HiFi-GAN D + G passes (B = %d x 8192, %s)" % (hg_B, hg_mode), runs)
    del st, eng, runs
    torch.cuda.empty_cache()


# ---- xVAPitch C5 iteration on its five streams (the benchmarked schedule: eager_disc, late_join), fixed random draws
from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
from xva_trainer_amd.xvapitch.decoder import VitsDecoder
from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
from xva_trainer_amd.xvapitch.train_step import XVAPitchStep
from xva_trainer_amd.xvapitch import ops as xops
B, Tt, Ty, VOCAB, LANGS, SEG = 16, 100, 400, 256, 31, 32
gen = torch.Generator().manual_seed(5)
ac = AcousticTrainPath(VOCAB, LANGS, pitch=True, compute="bf16", device=torch.device(dev), dropout_p=0.1, sdp_dropout_p=0.5)
dec, D = VitsDecoder(192, 512, compute="bf16", device=torch.device(dev)), VitsDiscriminator(compute="bf16", device=torch.device(dev))
for e in (dec, D):
    sd = {k: torch.randn(shape, generator=gen) * 0.02 for k, (off, numel, shape) in e.table.items()}
    for k in list(sd):
        if k.endswith("weight_g"):
            v = sd[k[:-1] + "v"]
            sd[k] = v.reshape(v.size(0), -1).norm(dim=1).reshape(sd[k].shape)
    e.load_state_dict(sd)
step = XVAPitchStep(GeneratorPass(ac, dec, SEG), D)
xl = torch.randint(Tt // 2, Tt + 1, (B,), generator=gen); xl[0] = Tt
yl0 = torch.randint(max(Ty // 2, SEG + 1), Ty + 1, (B,), generator=gen); yl0[0] = Ty
tok = torch.randint(1, VOCAB, (B, Tt), generator=gen) * (torch.arange(Tt)[None, :] < xl[:, None])
wl = (yl0 - 1) * 256 + torch.randint(0, 256, (B,), generator=gen)
wavs = torch.rand(B, (Ty - 1) * 256 + 255, generator=gen) * 0.1 - 0.05
wavs = wavs * (torch.arange(wavs.size(1))[None, :] < wl[:, None])
dv, li = torch.randn(B, 512, generator=gen), torch.randint(0, LANGS, (B,), generator=gen)
pit = (torch.rand(B, 1, Ty, generator=gen) * 3 - 1.2).clamp_min(0) * (torch.arange(Ty)[None, None, :] < yl0[:, None, None])
eps, noi = torch.randn(B, 192, Ty, generator=gen), torch.randn(B, 2, Tt, generator=gen)
ids = (torch.rand(B, generator=gen) * (yl0 - SEG + 1)).long()
c = lambda t: t.contiguous().to(dev)
LOSSES = ("loss", "loss_kl", "loss_duration", "loss_pitch", "loss_mel", "loss_gen", "loss_feat")
runs = []
for r in range(R):
    ac.train(True)
    ac.set_dropout_seed(20240905)
    step.gen.zero_grad(); D.zero_grad()
    y, yl, wav = step.gen.batch_from_wav(c(wavs), c(wl))
    o = step.generator_pass(c(tok), c(xl), y, yl, wav, c(dv), c(li), pitch_padded=c(pit), eps=c(eps), noise=c(noi), slice_ids=c(ids), eager_disc=True)
    o["loss"].backward()
    ld = step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
    torch.cuda.synchronize()
    xops.raise_deferred()
    d = {k: o[k].detach().float().reshape(-1).clone() for k in LOSSES}
    d["loss_disc"] = ld.detach().float().reshape(-1).clone()
    d["decoded segment"] = o["model_outputs"].detach().float().clone()
    d["grads: acoustic (all)"] = torch.cat([v.detach().float().flatten() for k, v in sorted(ac.grads().items()) if v is not None])
    d["grads: decoder"] = dec.grad.detach().float().clone()
    d["grads: discriminator"] = D.grad.detach().float().clone()
    runs.append(d)
report("xVAPitch C5 iteration (B = 16 x 100 x 400, five streams, eager_disc + late_join)", runs)
