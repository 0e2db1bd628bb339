"""CPU oracle for the HiFi-GAN v1 training step (floating point: torch fp32 on CPU, autograd for gradients).

Functional restatement driven by state_dicts with the REFERENCE's keys (old-style weight_norm: `weight_g` / `weight_v`;
spectral_norm on MSD discriminator 0: `weight_orig` / `weight_u` / `weight_v`).  Restates:
  ResBlock1, Generator                      python/hifigan/models.py:17-128
  DiscriminatorP, MultiPeriodDiscriminator  python/hifigan/models.py:140-200
  DiscriminatorS, MultiScaleDiscriminator   python/hifigan/models.py:203-260
  feature_loss / discriminator_loss / generator_loss   python/hifigan/models.py:263-294
  the D + G optimisation step               python/hifigan/xva_train.py:479-515
  torch.optim.AdamW (lr 2e-4, betas (0.8, 0.99), eps 1e-8, weight_decay 0.01)   python/hifigan/xva_train.py:298-300
  torch.nn.utils.weight_norm / spectral_norm (1 power iteration per training forward, eps 1e-12)
"""
import math

import torch
import torch.nn.functional as F

from oracle import mel as omel

LRELU_SLOPE = 0.1
UPSAMPLE_RATES = [8, 8, 2, 2]
UPSAMPLE_KERNELS = [16, 16, 4, 4]
UPSAMPLE_INITIAL = 512
RES_KERNELS = [3, 7, 11]
RES_DILATIONS = [[1, 3, 5], [1, 3, 5], [1, 3, 5]]
PERIODS = [2, 3, 5, 7, 11]


def get_padding(kernel_size, dilation=1):
    return int((kernel_size * dilation - dilation) / 2)


# ------------------------------------------------------------------ reparametrisations ----
def wn_weight(sd, pre):
    """torch.nn.utils.weight_norm (dim=0): w = g * v / ||v||, norm over all dims but 0."""
    v, g = sd[pre + "weight_v"], sd[pre + "weight_g"]
    dims = tuple(range(1, v.dim()))
    return v * (g / torch.sqrt((v * v).sum(dim=dims, keepdim=True)))


def sn_weight(sd, pre, training=True, eps=1e-12):
    """torch.nn.utils.spectral_norm.compute_weight: one power iteration IN PLACE on weight_u / weight_v (no grad), then
    sigma = u . (W v) with u, v detached copies; weight = weight_orig / sigma."""
    w = sd[pre + "weight_orig"]
    u, v = sd[pre + "weight_u"], sd[pre + "weight_v"]
    wm = w.reshape(w.size(0), -1)
    if training:
        with torch.no_grad():
            v.copy_(F.normalize(torch.mv(wm.t(), u), dim=0, eps=eps))
            u.copy_(F.normalize(torch.mv(wm, v), dim=0, eps=eps))
        u, v = u.clone(), v.clone()
    sigma = torch.dot(u, torch.mv(wm, v))
    return w / sigma


# ------------------------------------------------------------------ generator ----
def generator(sd, x):
    """x: (B, 80, T) mel -> (B, 1, T * 256) waveform in (-1, 1)."""
    x = F.conv1d(x, wn_weight(sd, "conv_pre."), sd["conv_pre.bias"], padding=3)
    nk = len(RES_KERNELS)
    for i, (u, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNELS)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, wn_weight(sd, "ups.%d." % i), sd["ups.%d.bias" % i], stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            r = _resblock1(sd, "resblocks.%d." % (i * nk + j), x, RES_KERNELS[j], RES_DILATIONS[j])
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)   # default slope 0.01 (models.py:124)
    x = F.conv1d(x, wn_weight(sd, "conv_post."), sd["conv_post.bias"], padding=3)
    return torch.tanh(x)


def _resblock1(sd, pre, x, k, dil):
    for m in range(3):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, wn_weight(sd, "%sconvs1.%d." % (pre, m)), sd["%sconvs1.%d.bias" % (pre, m)], padding=get_padding(k, dil[m]),
                      dilation=dil[m])
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, wn_weight(sd, "%sconvs2.%d." % (pre, m)), sd["%sconvs2.%d.bias" % (pre, m)], padding=get_padding(k, 1))
        x = xt + x
    return x


# ------------------------------------------------------------------ discriminators ----
def disc_p(sd, pre, x, period):
    fmap = []
    b, c, t = x.shape
    if t % period != 0:
        n_pad = period - (t % period)
        x = F.pad(x, (0, n_pad), "reflect")
        t = t + n_pad
    x = x.view(b, c, t // period, period)
    for i in range(5):
        stride = (3, 1) if i < 4 else (1, 1)
        x = F.conv2d(x, wn_weight(sd, "%sconvs.%d." % (pre, i)), sd["%sconvs.%d.bias" % (pre, i)], stride=stride, padding=(2, 0))
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv2d(x, wn_weight(sd, pre + "conv_post."), sd[pre + "conv_post.bias"], padding=(1, 0))
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


MSD_CFG = [(1, 128, 15, 1, 1, 7), (128, 128, 41, 2, 4, 20), (128, 256, 41, 2, 16, 20), (256, 512, 41, 4, 16, 20),
           (512, 1024, 41, 4, 16, 20), (1024, 1024, 41, 1, 16, 20), (1024, 1024, 5, 1, 1, 2)]


def disc_s(sd, pre, x, spectral, training=True):
    fmap = []
    wfn = (lambda p: sn_weight(sd, p, training)) if spectral else (lambda p: wn_weight(sd, p))
    for i, (cin, cout, k, s, g, p) in enumerate(MSD_CFG):
        x = F.conv1d(x, wfn("%sconvs.%d." % (pre, i)), sd["%sconvs.%d.bias" % (pre, i)], stride=s, padding=p, groups=g)
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv1d(x, wfn(pre + "conv_post."), sd[pre + "conv_post.bias"], padding=1)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def mpd(sd, y, y_hat):
    rs, gs, fr, fg = [], [], [], []
    for i, p in enumerate(PERIODS):
        r, f1 = disc_p(sd, "discriminators.%d." % i, y, p)
        g, f2 = disc_p(sd, "discriminators.%d." % i, y_hat, p)
        rs.append(r); gs.append(g); fr.append(f1); fg.append(f2)
    return rs, gs, fr, fg


def msd(sd, y, y_hat, training=True):
    rs, gs, fr, fg = [], [], [], []
    for i in range(3):
        if i != 0:
            y = F.avg_pool1d(y, 4, 2, padding=2)
            y_hat = F.avg_pool1d(y_hat, 4, 2, padding=2)
        r, f1 = disc_s(sd, "discriminators.%d." % i, y, i == 0, training)
        g, f2 = disc_s(sd, "discriminators.%d." % i, y_hat, i == 0, training)
        rs.append(r); gs.append(g); fr.append(f1); fg.append(f2)
    return rs, gs, fr, fg


# ------------------------------------------------------------------ losses ----
def feature_loss(fmap_r, fmap_g):
    loss = 0
    for dr, dg in zip(fmap_r, fmap_g):
        for rl, gl in zip(dr, dg):
            loss = loss + torch.mean(torch.abs(rl - gl))
    return loss * 2


def discriminator_loss(drs, dgs):
    loss = 0
    for dr, dg in zip(drs, dgs):
        loss = loss + torch.mean((1 - dr) ** 2) + torch.mean(dg ** 2)
    return loss


def generator_loss(dgs):
    loss = 0
    for dg in dgs:
        loss = loss + torch.mean((1 - dg) ** 2)
    return loss


def mel_for_loss(y):
    """mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, fmax_for_loss=None) (xva_train.py:480)."""
    return omel.mel_m2(y, fmax=None)


# ------------------------------------------------------------------ optimiser ----
def adamw_step(params, grads, state, lr=2e-4, betas=(0.8, 0.99), eps=1e-8, weight_decay=0.01):
    b1, b2 = betas
    for k, p in params.items():
        g = grads.get(k)
        if g is None:
            continue
        st = state.setdefault(k, {"step": 0, "exp_avg": torch.zeros_like(p), "exp_avg_sq": torch.zeros_like(p)})
        st["step"] += 1
        t = st["step"]
        p.mul_(1 - lr * weight_decay)
        st["exp_avg"].mul_(b1).add_(g, alpha=1 - b1)
        st["exp_avg_sq"].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
        denom = (st["exp_avg_sq"].sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(st["exp_avg"], denom, value=-lr / bc1)


def _leaves(sd):
    """Trainable tensors of a state_dict (everything but the spectral-norm power-iteration buffers)."""
    return [k for k in sd if not (k.endswith("weight_u") or (k.endswith("weight_v") and (k[:-1] + "u") in sd))]


def train_step(g_sd, mpd_sd, msd_sd, x_mel, y_wav, y_mel, opt_g, opt_d, lr=2e-4):
    """One reference iteration (xva_train.py:479-515): G forward, D step (MPD + MSD on detached fake), G step
    (45 * L1 mel + feature matching + LSGAN).  Mutates the state_dicts / optimizer states.  Returns a dict of scalars and
    the gradient dicts (before the optimizer steps)."""
    gl = {k: g_sd[k].detach().clone().requires_grad_(True) for k in _leaves(g_sd)}
    y = y_wav.unsqueeze(1)
    y_g_hat = generator({**g_sd, **gl}, x_mel)
    y_g_hat_mel = mel_for_loss(y_g_hat.squeeze(1))

    # ---- discriminator step
    pl = {k: mpd_sd[k].detach().clone().requires_grad_(True) for k in _leaves(mpd_sd)}
    sl = {k: msd_sd[k].detach().clone().requires_grad_(True) for k in _leaves(msd_sd)}
    mw = dict(msd_sd); mw.update(sl)      # weight_u / weight_v buffers stay shared (updated in place)
    r, g, _, _ = mpd({**mpd_sd, **pl}, y, y_g_hat.detach())
    loss_disc_f = discriminator_loss(r, g)
    r, g, _, _ = msd(mw, y, y_g_hat.detach())
    loss_disc_s = discriminator_loss(r, g)
    loss_disc_all = loss_disc_s + loss_disc_f
    loss_disc_all.backward()
    d_grads = {"mpd." + k: v.grad for k, v in pl.items()}
    d_grads.update({"msd." + k: v.grad for k, v in sl.items()})
    with torch.no_grad():
        params = {"mpd." + k: mpd_sd[k] for k in pl}
        params.update({"msd." + k: msd_sd[k] for k in sl})
        adamw_step(params, d_grads, opt_d, lr=lr)

    # ---- generator step (discriminators already updated, as in the reference)
    loss_mel = F.l1_loss(y_mel, y_g_hat_mel) * 45
    _, g_f, fr_f, fg_f = mpd(mpd_sd, y, y_g_hat)
    _, g_s, fr_s, fg_s = msd(msd_sd, y, y_g_hat)
    loss_fm_f = feature_loss(fr_f, fg_f)
    loss_fm_s = feature_loss(fr_s, fg_s)
    loss_gen_f = generator_loss(g_f)
    loss_gen_s = generator_loss(g_s)
    loss_gen_all = loss_gen_s + loss_gen_f + loss_fm_s + loss_fm_f + loss_mel
    loss_gen_all.backward()
    g_grads = {k: v.grad for k, v in gl.items()}
    with torch.no_grad():
        adamw_step({k: g_sd[k] for k in gl}, g_grads, opt_g, lr=lr)
    out = {"loss_disc_all": float(loss_disc_all), "loss_disc_f": float(loss_disc_f), "loss_disc_s": float(loss_disc_s),
           "loss_gen_all": float(loss_gen_all), "loss_mel": float(loss_mel), "loss_fm_f": float(loss_fm_f), "loss_fm_s": float(loss_fm_s),
           "loss_gen_f": float(loss_gen_f), "loss_gen_s": float(loss_gen_s), "mel_error": float(loss_mel) / 45}
    return out, g_grads, d_grads, y_g_hat.detach()


# ------------------------------------------------------------------ seeded state_dicts / inputs ----
def _wn(sd, pre, shape, g, std=None, bias=True, transpose=False):
    """weight_norm parametrisation of a conv with weight `shape` (dim 0 = out channels, or in channels for ConvTranspose)."""
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    if std is None:
        v = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
    else:
        v = torch.randn(shape, generator=g) * std
    dims = tuple(range(1, len(shape)))
    sd[pre + "bias"] = (torch.rand(shape[1] if transpose else shape[0], generator=g) * 2 - 1) / math.sqrt(fan_in)
    sd[pre + "weight_g"] = torch.sqrt((v * v).sum(dim=dims, keepdim=True)) * (0.75 + 0.5 * torch.rand([shape[0]] + [1] * (len(shape) - 1), generator=g))
    sd[pre + "weight_v"] = v


def init_generator_sd(seed):
    """Reference key order (Generator.state_dict()): conv_pre, ups.*, resblocks.*, conv_post.  init_weights uses N(0, 0.01) for
    ups / resblocks / conv_post (utils.py:23-26); a larger std keeps the synthetic network's activations O(1)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    _wn(sd, "conv_pre.", (512, 80, 7), g)
    ch = UPSAMPLE_INITIAL
    for i, (u, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNELS)):
        _wn(sd, "ups.%d." % i, (ch, ch // 2, k), g, std=0.04, transpose=True)
        ch //= 2
    ch = UPSAMPLE_INITIAL
    for i in range(4):
        ch //= 2
        for j, k in enumerate(RES_KERNELS):
            for name in ("convs1", "convs2"):
                for m in range(3):
                    _wn(sd, "resblocks.%d.%s.%d." % (i * 3 + j, name, m), (ch, ch, k), g, std=0.04)
    _wn(sd, "conv_post.", (1, 32, 7), g, std=0.04)
    return sd


def init_mpd_sd(seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    chans = [(1, 32), (32, 128), (128, 512), (512, 1024), (1024, 1024)]
    for d in range(5):
        for i, (ci, co) in enumerate(chans):
            _wn(sd, "discriminators.%d.convs.%d." % (d, i), (co, ci, 5, 1), g)
        _wn(sd, "discriminators.%d.conv_post." % d, (1, 1024, 3, 1), g)
    return sd


def init_msd_sd(seed):
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for d in range(3):
        for i, (ci, co, k, s, gr, p) in enumerate(MSD_CFG + [(1024, 1, 3, 1, 1, 1)]):
            pre = "discriminators.%d.%s" % (d, "convs.%d." % i if i < 7 else "conv_post.")
            shape = (co, ci // gr, k)
            if d == 0:
                fan_in = shape[1] * shape[2]
                w = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
                sd[pre + "bias"] = (torch.rand(co, generator=g) * 2 - 1) / math.sqrt(fan_in)
                sd[pre + "weight_orig"] = w
                sd[pre + "weight_u"] = F.normalize(torch.randn(co, generator=g), dim=0, eps=1e-12)
                sd[pre + "weight_v"] = F.normalize(torch.randn(shape[1] * shape[2], generator=g), dim=0, eps=1e-12)
            else:
                _wn(sd, pre, shape, g)
    return sd


def synth_batch(B, seed, segment=8192):
    """(x mel (B,80,segment/256), y wav (B,segment), y_mel for the loss (fmax=None)) like MelDataset.__getitem__
    (meldataset.py:340-373): peak-normalised * 0.95 crops of the synthetic clips."""
    import numpy as np
    wav = np.stack([omel.peak_normalize(omel.synth_wave(segment, seed + i)) * 0.95 for i in range(B)]).astype(np.float32)
    y = torch.from_numpy(wav)
    x = omel.mel_m2(y, fmax=8000)
    y_mel = omel.mel_m2(y, fmax=None)
    return x, y, y_mel


# ------------------------------------------------------------------ storage = "bf16": one layer at a time ----
# The throughput mode stores activations and effective (weight-/spectral-normed) weights as bf16 and accumulates in fp32.  Rounding after
# each of ~50 layers does not compose into a tight end-to-end bound (the rounded network is a different, chaotic function of its input: see
# DESIGN.md §5), so the storage-dtype parity is checked TEACHER-FORCED: each function below restates ONE layer of the reference (same
# models.py lines as above) on operands that are already bf16 values — the engine's own stored input of that layer — with the products
# and sums in fp64 and ONE rounding where the engine stores.  The only differences left are fp32-vs-fp64 summation order, i.e. a rare
# 1-ulp flip of the stored bf16 value.
SLOPE32 = float(torch.tensor(LRELU_SLOPE, dtype=torch.float32))


def bf16r(x):
    """round to nearest-even bf16, returned as fp64"""
    return x.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def lrelu64(x, slope=SLOPE32):
    return torch.where(x > 0, x, x * slope)


def layer_conv(x, w, b, stride=1, padding=0, dilation=1, groups=1, weight_fp32=False):
    """Conv1d on bf16-valued x (N, C, T) with the bf16-rounded effective weight (fp32 for the 1-channel boundary convolutions, which the
    engine runs outside the matrix pipe); returns the UNROUNDED fp64 result."""
    wq = w.double() if weight_fp32 else bf16r(w)
    return F.conv1d(x.double(), wq, b.double(), stride=stride, padding=padding, dilation=dilation, groups=groups)


def layer_ups(sd, i, prev):
    """lrelu + ups[i] (models.py:115-116) on the stored input `prev` (N, C, T): returns (u, lrelu(u)) as stored (both rounded from the same
    fp32 value: the activated copy is NOT the LeakyReLU of the rounded one)."""
    u, k = UPSAMPLE_RATES[i], UPSAMPLE_KERNELS[i]
    a = bf16r(lrelu64(prev.double()))
    v = F.conv_transpose1d(a, bf16r(wn_weight(sd, "ups.%d." % i)), sd["ups.%d.bias" % i].double(), stride=u, padding=(k - u) // 2)
    return bf16r(v), bf16r(lrelu64(v))


def layer_res_c1(sd, rb, m, xact):
    """xt1 = lrelu(c1(lrelu(x))) (models.py:43-45), stored activated; xact = the stored lrelu(x)."""
    k, dil = RES_KERNELS[rb % 3], RES_DILATIONS[rb % 3][m]
    pre = "resblocks.%d.convs1.%d." % (rb, m)
    return bf16r(lrelu64(layer_conv(xact, wn_weight(sd, pre), sd[pre + "bias"], padding=get_padding(k, dil), dilation=dil)))


def layer_res_c2(sd, rb, m, xt1, xcur):
    """x <- c2(xt1) + x (models.py:46-47): the UNROUNDED new x (the engine stores bf16(x) and bf16(lrelu(x)) from it, and for the last
    block of a resblock feeds it into the running mean of the stage)."""
    k = RES_KERNELS[rb % 3]
    pre = "resblocks.%d.convs2.%d." % (rb, m)
    return layer_conv(xt1, wn_weight(sd, pre), sd[pre + "bias"], padding=get_padding(k, 1)) + xcur.double()


# ------------------------------------------------------------------ the VITS waveform decoder (xVAPitch) ----
def vits_decoder(sd, x, g=None):
    """HifiganGenerator.forward (python/xvapitch/hifigan.py:233-262) as xVAPitch builds it (python/xvapitch/model.py:134-149): the v1
    generator above on `in_channels` latent channels, conv_pre / conv_post WITHOUT weight norm, conv_post without bias, and
    o = conv_pre(x) + cond_layer(g) with g (B, cond, 1) the speaker vector.  x: (B, in, T) -> (B, 1, T * 256)."""
    x = F.conv1d(x, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    if g is not None:
        x = x + F.conv1d(g, sd["cond_layer.weight"], sd["cond_layer.bias"])
    nk = len(RES_KERNELS)
    for i, (u, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNELS)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, wn_weight(sd, "ups.%d." % i), sd["ups.%d.bias" % i], stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            r = _resblock1(sd, "resblocks.%d." % (i * nk + j), x, RES_KERNELS[j], RES_DILATIONS[j])
            xs = r if xs is None else xs + r
        x = xs / nk
    x = F.leaky_relu(x)
    x = F.conv1d(x, sd["conv_post.weight"], None, padding=3)
    return torch.tanh(x)


def init_vits_decoder_sd(seed, in_channels=192, cond_channels=512):
    """Seeded state_dict in the reference key order (HifiganGenerator.state_dict() after the two remove_weight_norm calls of
    hifigan.py:226-230: conv_pre.bias / .weight, ups.*, resblocks.*, conv_post.weight, cond_layer.weight / .bias)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    sd["conv_pre.bias"] = (torch.rand(UPSAMPLE_INITIAL, generator=g) * 2 - 1) / math.sqrt(in_channels * 7)
    sd["conv_pre.weight"] = (torch.rand(UPSAMPLE_INITIAL, in_channels, 7, generator=g) * 2 - 1) / math.sqrt(in_channels * 7)
    ch = UPSAMPLE_INITIAL
    for i, (u, k) in enumerate(zip(UPSAMPLE_RATES, UPSAMPLE_KERNELS)):
        _wn(sd, "ups.%d." % i, (ch, ch // 2, k), g, std=0.04, transpose=True)
        ch //= 2
    ch = UPSAMPLE_INITIAL
    for i in range(4):
        ch //= 2
        for j, k in enumerate(RES_KERNELS):
            for name in ("convs1", "convs2"):
                for m in range(3):
                    _wn(sd, "resblocks.%d.%s.%d." % (i * 3 + j, name, m), (ch, ch, k), g, std=0.04)
    sd["conv_post.weight"] = torch.randn(1, 32, 7, generator=g) * 0.04
    if cond_channels:
        sd["cond_layer.weight"] = (torch.rand(UPSAMPLE_INITIAL, cond_channels, 1, generator=g) * 2 - 1) / math.sqrt(cond_channels)
        sd["cond_layer.bias"] = (torch.rand(UPSAMPLE_INITIAL, generator=g) * 2 - 1) / math.sqrt(cond_channels)
    return sd


# ------------------------------------------------------------------ VitsDiscriminator (xVAPitch) ----
VITS_S_CFG = [(1, 16, 15, 1, 1, 7), (16, 64, 41, 4, 4, 20), (64, 256, 41, 4, 16, 20), (256, 1024, 41, 4, 64, 20), (1024, 1024, 41, 4, 256, 20),
              (1024, 1024, 5, 1, 1, 2)]


def vits_disc_s(sd, pre, x):
    """DiscriminatorS.forward of python/xvapitch/model.py:1548-1587 (weight norm)."""
    fmap = []
    for i, (cin, cout, k, s, g, p) in enumerate(VITS_S_CFG):
        x = F.conv1d(x, wn_weight(sd, "%sconvs.%d." % (pre, i)), sd["%sconvs.%d.bias" % (pre, i)], stride=s, padding=p, groups=g)
        x = F.leaky_relu(x, LRELU_SLOPE)
        fmap.append(x)
    x = F.conv1d(x, wn_weight(sd, pre + "conv_post."), sd[pre + "conv_post.bias"], padding=1)
    fmap.append(x)
    return torch.flatten(x, 1, -1), fmap


def vits_disc(sd, y, y_hat):
    """VitsDiscriminator.forward (python/xvapitch/model.py:1609-1640): nets.0 the scale discriminator, nets.1-5 periods 2, 3, 5, 7, 11.
    Returns (scores real, feats real, scores fake, feats fake)."""
    rs, fr, gs, fg = [], [], [], []
    for n in range(6):
        pre = "nets.%d." % n
        f = (lambda t: vits_disc_s(sd, pre, t)) if n == 0 else (lambda t: disc_p(sd, pre, t, PERIODS[n - 1]))
        r, f1 = f(y)
        g, f2 = f(y_hat)
        rs.append(r); fr.append(f1); gs.append(g); fg.append(f2)
    return rs, fr, gs, fg


def init_vits_disc_sd(seed):
    """Seeded state_dict in the reference key order (VitsDiscriminator.state_dict())."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for i, (ci, co, k, s, gr, p) in enumerate(VITS_S_CFG + [(1024, 1, 3, 1, 1, 1)]):
        _wn(sd, "nets.0.%s" % ("convs.%d." % i if i < 6 else "conv_post."), (co, ci // gr, k), g)
    chans = [(1, 32), (32, 128), (128, 512), (512, 1024), (1024, 1024)]
    for d in range(5):
        for i, (ci, co) in enumerate(chans):
            _wn(sd, "nets.%d.convs.%d." % (d + 1, i), (co, ci, 5, 1), g)
        _wn(sd, "nets.%d.conv_post." % (d + 1), (1, 1024, 3, 1), g)
    return sd
