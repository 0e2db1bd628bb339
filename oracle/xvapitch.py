"""CPU oracle for the xVAPitch-only blocks built so far (TEST INFRASTRUCTURE ONLY — never imported by the package).

Functional torch / numpy restatements driven by state_dicts with the REFERENCE's keys; oracle/gen_golden_xvapitch.py asserts them equal
to the reference's own modules and records tests/golden/xvapitch_blocks.npz.
  wn()                WN.forward + fused_add_tanh_sigmoid_multiply   python/xvapitch/wavenet.py:5-12,92-109
  coupling()          ResidualCouplingBlock.forward (mean_only)      python/xvapitch/model.py:1519-1535
  posterior_encoder() PosteriorEncoder.forward                       python/xvapitch/model.py:1462-1475
  maximum_path()      monotonic alignment search                     python/xvapitch/util.py:14-53
  segment()           util.py:166-178 ;  kl_loss()  VitsGeneratorLoss.kl_loss  python/xvapitch/losses.py:87-104
"""
import numpy as np
import torch
import torch.nn.functional as F


def _wn_weight(sd, pre):
    v, g = sd[pre + "weight_v"], sd[pre + "weight_g"]
    return g * v / v.reshape(v.size(0), -1).norm(dim=1).reshape(-1, 1, 1)          # torch.nn.utils.weight_norm (old API, dim 0)


def wn(sd, x, x_mask, g=None, hidden=None, kernel_size=5, dilation_rate=1, num_layers=4, pre=""):
    """x (B, H, T), x_mask (B, 1, T), g (B, c_in, 1) or None -> (B, H, T)."""
    H = hidden or x.size(1)
    output = torch.zeros_like(x)
    if g is not None:
        g = F.conv1d(g, _wn_weight(sd, pre + "cond_layer."), sd[pre + "cond_layer.bias"])
    for i in range(num_layers):
        d = dilation_rate ** i
        x_in = F.conv1d(x, _wn_weight(sd, pre + "in_layers.%d." % i), sd[pre + "in_layers.%d.bias" % i], dilation=d, padding=(kernel_size * d - d) // 2)
        g_l = g[:, i * 2 * H:(i + 1) * 2 * H, :] if g is not None else torch.zeros_like(x_in)
        in_act = x_in + g_l
        acts = torch.tanh(in_act[:, :H]) * torch.sigmoid(in_act[:, H:])
        rs = F.conv1d(acts, _wn_weight(sd, pre + "res_skip_layers.%d." % i), sd[pre + "res_skip_layers.%d.bias" % i])
        if i < num_layers - 1:
            x = (x + rs[:, :H]) * x_mask
            output = output + rs[:, H:]
        else:
            output = output + rs
    return output * x_mask


def coupling(sd, x, x_mask, g=None, reverse=False, **wn_args):
    half = x.size(1) // 2
    x0, x1 = x[:, :half], x[:, half:]
    h = F.conv1d(x0, sd["pre.weight"], sd["pre.bias"]) * x_mask
    h = wn(sd, h, x_mask, g, pre="enc.", **wn_args)
    m = F.conv1d(h, sd["post.weight"], sd["post.bias"]) * x_mask
    x1 = (x1 - m) * x_mask if reverse else m + x1 * x_mask
    return torch.cat([x0, x1], 1)


def posterior_encoder(sd, x, x_lengths, g, eps, out_channels, **wn_args):
    """PosteriorEncoder.forward (model.py:1462-1475) with the N(0, 1) draw passed in.  Returns z, mean, log_scale, x_mask."""
    x_mask = (torch.arange(x.size(2))[None, :] < x_lengths[:, None]).to(x.dtype).unsqueeze(1)
    h = F.conv1d(x, sd["pre.weight"], sd["pre.bias"]) * x_mask
    h = wn(sd, h, x_mask, g, pre="enc.", **wn_args)
    stats = F.conv1d(h, sd["proj.weight"], sd["proj.bias"]) * x_mask
    mean, log_scale = torch.split(stats, out_channels, dim=1)
    return (mean + eps * torch.exp(log_scale)) * x_mask, mean, log_scale, x_mask


def maximum_path(value, mask):
    """value, mask (B, t_x, t_y) numpy -> 0/1 path, the reference's loop."""
    value = value * mask
    mask = mask.astype(bool)
    b, t_x, t_y = value.shape
    direction = np.zeros(value.shape, dtype=np.int64)
    v = np.zeros((b, t_x), dtype=np.float32)
    x_range = np.arange(t_x, dtype=np.float32).reshape(1, -1)
    for j in range(t_y):
        v0 = np.pad(v, [[0, 0], [1, 0]], mode="constant", constant_values=-np.inf)[:, :-1]
        max_mask = v >= v0
        v_max = np.where(max_mask, v, v0)
        direction[:, :, j] = max_mask
        v = np.where(x_range <= j, v_max + value[:, :, j], -np.inf)
    direction = np.where(mask, direction, 1)
    path = np.zeros(value.shape, dtype=np.float32)
    index = mask[:, :, 0].sum(1).astype(np.int64) - 1
    rng = np.arange(b)
    for j in reversed(range(t_y)):
        path[rng, index, j] = 1
        index = index + direction[rng, index, j] - 1
    return path * mask.astype(np.float32)


def segment(x, idx, S):
    return torch.stack([x[i, :, int(idx[i]):int(idx[i]) + S] for i in range(x.size(0))])


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    kl = logs_p - logs_q - 0.5 + 0.5 * ((z_p - m_p) ** 2) * torch.exp(-2.0 * logs_p)
    kl_sw = kl * z_mask
    return kl_sw.sum() / z_mask.sum(), kl_sw


def rel_transformer(sd, x, x_mask, num_heads, num_layers, kernel_size, window, drop=None, site0=0):
    """RelativePositionTransformer.forward (python/xvapitch/glow_tts.py:463-484; layer_norm_type "2", heads sharing the relative
    embeddings, in = hidden = out) — `drop(site, tensor) -> tensor`, when given, stands for nn.Dropout at the module's four calls per layer, in
    the reference's order: site0 + 4 i + 0 the attention weights (:204), + 1 the attention block's output (:473), + 2 the feed-forward hidden
    activation (:344), + 3 the feed-forward output (:477); None = eval mode — with the attention's relative terms (:173-292) written directly: for r = j - i in
    [-window, window] the scores get q_i . emb_rel_k[r + window] / sqrt(dk) and the output gets p[i][j] emb_rel_v[r + window]; the
    reference reaches the same numbers through zero-padded embeddings and two pad / reshape index shifts."""
    B, Cc, T = x.shape
    H, dk = num_heads, Cc // num_heads
    attn_mask = x_mask.unsqueeze(2) * x_mask.unsqueeze(-1)                                   # (B, 1, T, T)
    idx = torch.arange(T)
    rel = idx[None, :] - idx[:, None]                                                        # j - i
    inwin = (rel.abs() <= window)
    ridx = (rel + window).clamp(0, 2 * window)
    pad_l, pad_r = (kernel_size - 1) // 2, kernel_size // 2
    for i in range(num_layers):
        a, f = "attn_layers.%d." % i, "ffn_layers.%d." % i
        x = x * x_mask
        q = F.conv1d(x, sd[a + "conv_q.weight"], sd[a + "conv_q.bias"]).view(B, H, dk, T).transpose(2, 3)
        k = F.conv1d(x, sd[a + "conv_k.weight"], sd[a + "conv_k.bias"]).view(B, H, dk, T).transpose(2, 3)
        v = F.conv1d(x, sd[a + "conv_v.weight"], sd[a + "conv_v.bias"]).view(B, H, dk, T).transpose(2, 3)
        ek, ev = sd[a + "emb_rel_k"][0], sd[a + "emb_rel_v"][0]                              # (2w + 1, dk)
        scores = q @ k.transpose(-2, -1)
        qe = q @ ek.t()                                                                      # (B, H, T, 2w + 1)
        scores = scores + torch.where(inwin, qe.gather(-1, ridx.expand(B, H, T, T)), torch.zeros(()))
        scores = (scores / dk ** 0.5).masked_fill(attn_mask == 0, -1e4)
        p = torch.softmax(scores, dim=-1)
        if drop is not None:
            p = drop(site0 + 4 * i, p)
        o = p @ v
        pw = torch.zeros(B, H, T, 2 * window + 1).scatter_add(-1, ridx.expand(B, H, T, T), p * inwin)
        o = o + pw @ ev
        o = o.transpose(2, 3).contiguous().view(B, Cc, T)
        y = F.conv1d(o, sd[a + "conv_o.weight"], sd[a + "conv_o.bias"])
        if drop is not None:
            y = drop(site0 + 4 * i + 1, y)
        x = F.layer_norm((x + y).transpose(1, -1), (Cc,), sd["norm_layers_1.%d.gamma" % i], sd["norm_layers_1.%d.beta" % i], 1e-5).transpose(1, -1)
        last = i == num_layers - 1
        h = torch.relu(F.conv1d(F.pad(x * x_mask, (pad_l, pad_r)), sd[f + "conv_1.weight"], sd[f + "conv_1.bias"]))
        if drop is not None:
            h = drop(site0 + 4 * i + 2, h)
        y = F.conv1d(F.pad(h * x_mask, (pad_l, pad_r)), sd[f + "conv_2.weight"], sd[f + "conv_2.bias"]) * x_mask
        if drop is not None:
            y = drop(site0 + 4 * i + 3, y)
        if last and "proj.weight" in sd:                                                     # hidden != out (:479-480)
            x = F.conv1d(x, sd["proj.weight"], sd["proj.bias"])
        Co = x.size(1)
        if Co != 1 or not last:                                                              # out_channels == 1: the stack returns proj(x) (:482)
            x = F.layer_norm((x + y).transpose(1, -1), (Co,), sd["norm_layers_2.%d.gamma" % i], sd["norm_layers_2.%d.beta" % i], 1e-5).transpose(1, -1)
    return x * x_mask


def dds_conv(sd, x, x_mask, g=None, kernel_size=3, num_layers=3, pre="", drop=None, site0=0):
    """DilatedDepthSeparableConv.forward (python/xvapitch/sdp.py:76-93); `drop(site0 + layer, y)` stands for nn.Dropout (:90), None = off."""
    Cc = x.size(1)
    if g is not None:
        x = x + g
    for i in range(num_layers):
        d = kernel_size ** i
        y = F.conv1d(x * x_mask, sd["%sconvs_sep.%d.weight" % (pre, i)], sd["%sconvs_sep.%d.bias" % (pre, i)], groups=Cc, dilation=d,
                     padding=(kernel_size * d - d) // 2)
        y = F.gelu(F.layer_norm(y.transpose(1, -1), (Cc,), sd["%snorms_1.%d.gamma" % (pre, i)], sd["%snorms_1.%d.beta" % (pre, i)], 1e-5).transpose(1, -1))
        y = F.conv1d(y, sd["%sconvs_1x1.%d.weight" % (pre, i)], sd["%sconvs_1x1.%d.bias" % (pre, i)])
        y = F.gelu(F.layer_norm(y.transpose(1, -1), (Cc,), sd["%snorms_2.%d.gamma" % (pre, i)], sd["%snorms_2.%d.beta" % (pre, i)], 1e-5).transpose(1, -1))
        if drop is not None:
            y = drop(site0 + i, y)
        x = x + y
    return x * x_mask


def rq_spline(x, uw, uh, ud, bound=5.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """piecewise_rational_quadratic_transform(inverse=False, tails="linear") (python/xvapitch/util.py:203-391): x (...), uw / uh (..., K),
    ud (..., K - 1) -> y, log|det|.  Outside [-bound, bound]: identity; inside: the rational-quadratic map of the bin that holds x, with knot
    derivatives softplus(ud) + min_d inside and exactly 1 at the two boundary knots."""
    K = uw.size(-1)
    inside = (x >= -bound) & (x <= bound)

    def edges(u, m):
        w = m + (1 - m * K) * torch.softmax(u, -1)
        c = F.pad(torch.cumsum(w, -1), (1, 0)) * (2 * bound) - bound
        c = torch.cat([torch.full_like(c[..., :1], -bound), c[..., 1:-1], torch.full_like(c[..., :1], bound)], -1)
        return c, c[..., 1:] - c[..., :-1]
    cw, w = edges(uw, min_w)
    ch, hh = edges(uh, min_h)
    const = float(np.log(np.exp(1 - min_d) - 1))
    d = min_d + F.softplus(torch.cat([torch.full_like(ud[..., :1], const), ud, torch.full_like(ud[..., :1], const)], -1))
    loc = cw.clone()
    loc[..., -1] = loc[..., -1] + 1e-6
    xc = x.clamp(-bound, bound)
    k = ((xc[..., None] >= loc).sum(-1) - 1).clamp(0, K - 1)[..., None]
    g = lambda t: t.gather(-1, k)[..., 0]
    cwk, wk, chk, hk, dk, dk1 = g(cw), g(w), g(ch), g(hh), g(d), g(d[..., 1:])
    th = (xc - cwk) / wk
    om = th * (1 - th)
    dl = hk / wk
    den = dl + (dk + dk1 - 2 * dl) * om
    y = chk + hk * (dl * th ** 2 + dk * om) / den
    ld = torch.log(dl ** 2 * (dk1 * th ** 2 + 2 * dl * om + dk * (1 - th) ** 2)) - 2 * torch.log(den)
    return torch.where(inside, y, x), torch.where(inside, ld, torch.zeros_like(ld))


def conv_flow(sd, x, x_mask, g, hidden, kernel_size=3, num_layers=3, num_bins=10, tail_bound=5.0, pre=""):
    """ConvFlow.forward (python/xvapitch/sdp.py:144-176), forward direction, 2-channel input."""
    x0, x1 = x[:, :1], x[:, 1:]
    h = F.conv1d(x0, sd[pre + "pre.weight"], sd[pre + "pre.bias"])
    h = dds_conv(sd, h, x_mask, g, kernel_size, num_layers, pre=pre + "convs.")
    h = F.conv1d(h, sd[pre + "proj.weight"], sd[pre + "proj.bias"]) * x_mask
    b, c, t = x0.shape
    h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    y1, ld = rq_spline(x1, h[..., :num_bins] / hidden ** 0.5, h[..., num_bins:2 * num_bins] / hidden ** 0.5, h[..., 2 * num_bins:], tail_bound)
    return torch.cat([x0, y1], 1) * x_mask, (ld * x_mask).sum((1, 2))


def sdp_forward(sd, x, x_mask, dr, noise, hidden, kernel_size=3, num_flows=4, g=None, lang_emb=None, drop=None, site0=0):
    """StochasticDurationPredictor.forward, training direction (python/xvapitch/sdp.py:247-310): negative log-likelihood (B,) of the durations
    `dr` with variational dequantisation; `noise` is the N(0, 1) draw of :281.  `drop`: nn.Dropout of `convs` (sites site0 + 0..2) and
    `post_convs` (site0 + 3..5), the two DilatedDepthSeparableConv built with dropout_p (:227,237); the flows' have none (:144)."""
    import math
    x = F.conv1d(x, sd["pre.weight"], sd["pre.bias"])
    if g is not None:
        x = x + F.conv1d(g, sd["cond.weight"], sd["cond.bias"])
    if lang_emb is not None:
        x = x + F.conv1d(lang_emb, sd["cond_lang.weight"], sd["cond_lang.bias"])
    x = dds_conv(sd, x, x_mask, None, kernel_size, 3, pre="convs.", drop=drop, site0=site0)
    x = F.conv1d(x, sd["proj.weight"], sd["proj.bias"]) * x_mask
    h = F.conv1d(dr, sd["post_pre.weight"], sd["post_pre.bias"])
    h = dds_conv(sd, h, x_mask, None, kernel_size, 3, pre="post_convs.", drop=drop, site0=site0 + 3)
    h = F.conv1d(h, sd["post_proj.weight"], sd["post_proj.bias"]) * x_mask

    def affine(pre, z):
        return (z * torch.exp(sd[pre + "log_scale"]) + sd[pre + "translation"]) * x_mask, (sd[pre + "log_scale"] * x_mask).sum((1, 2))
    noise = noise * x_mask
    z_q, ld_q = noise, 0.0
    for idx in range(num_flows + 1):
        if idx == 0:
            z_q, ld = affine("post_flows.0.", z_q)
        else:
            z_q, ld = conv_flow(sd, z_q, x_mask, x + h, hidden, kernel_size, 3, pre="post_flows.%d." % idx)
            z_q = torch.flip(z_q, [1])
        ld_q = ld_q + ld
    z_u, z_v = z_q[:, :1], z_q[:, 1:]
    u = torch.sigmoid(z_u) * x_mask
    z0 = (dr - u) * x_mask
    ld_q = ld_q + ((F.logsigmoid(z_u) + F.logsigmoid(-z_u)) * x_mask).sum((1, 2))
    nll_post = (-0.5 * (math.log(2 * math.pi) + noise ** 2) * x_mask).sum((1, 2)) - ld_q
    z0 = torch.log(torch.clamp_min(z0, 1e-5)) * x_mask
    ld_tot = (-z0).sum((1, 2))
    z = torch.cat([z0, z_v], 1)
    for idx in range(num_flows + 1):
        if idx == 0:
            z, ld = affine("flows.0.", z)
        else:
            z, ld = conv_flow(sd, z, x_mask, x, hidden, kernel_size, 3, pre="flows.%d." % idx)
            z = torch.flip(z, [1])
        ld_tot = ld_tot + ld
    nll_flow = (0.5 * (math.log(2 * math.pi) + z ** 2) * x_mask).sum((1, 2)) - ld_tot
    return nll_flow + nll_post


def rq_spline_inverse(y, uw, uh, ud, bound=5.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """piecewise_rational_quadratic_transform(inverse=True, tails="linear") (python/xvapitch/util.py:203-350): the bin is searched over the
    cumulative heights (:322), x is the root 2c / (-b - sqrt(b^2 - 4ac)) of the bin's quadratic (:325-340).  Returns x only (the sampling
    direction of ConvFlow discards log|det|, sdp.py:174-176)."""
    K = uw.size(-1)
    inside = (y >= -bound) & (y <= bound)

    def edges(u, m):
        w = m + (1 - m * K) * torch.softmax(u, -1)
        c = F.pad(torch.cumsum(w, -1), (1, 0)) * (2 * bound) - bound
        c = torch.cat([torch.full_like(c[..., :1], -bound), c[..., 1:-1], torch.full_like(c[..., :1], bound)], -1)
        return c, c[..., 1:] - c[..., :-1]
    cw, w = edges(uw, min_w)
    ch, hh = edges(uh, min_h)
    const = float(np.log(np.exp(1 - min_d) - 1))
    d = min_d + F.softplus(torch.cat([torch.full_like(ud[..., :1], const), ud, torch.full_like(ud[..., :1], const)], -1))
    loc = ch.clone()
    loc[..., -1] = loc[..., -1] + 1e-6
    yc = y.clamp(-bound, bound)
    k = ((yc[..., None] >= loc).sum(-1) - 1).clamp(0, K - 1)[..., None]
    g = lambda t: t.gather(-1, k)[..., 0]
    cwk, wk, chk, hk, dk, dk1 = g(cw), g(w), g(ch), g(hh), g(d), g(d[..., 1:])
    dl = hk / wk
    dy = yc - chk
    a = dy * (dk + dk1 - 2 * dl) + hk * (dl - dk)
    b = hk * dk - dy * (dk + dk1 - 2 * dl)
    c = -dl * dy
    root = (2 * c) / (-b - torch.sqrt((b * b - 4 * a * c).clamp_min(0)))
    return torch.where(inside, root * wk + cwk, y)


def conv_flow_reverse(sd, x, x_mask, g, hidden, kernel_size=3, num_layers=3, num_bins=10, tail_bound=5.0, pre=""):
    """ConvFlow.forward(reverse=True) (python/xvapitch/sdp.py:144-176)."""
    x0, x1 = x[:, :1], x[:, 1:]
    h = F.conv1d(x0, sd[pre + "pre.weight"], sd[pre + "pre.bias"])
    h = dds_conv(sd, h, x_mask, g, kernel_size, num_layers, pre=pre + "convs.")
    h = F.conv1d(h, sd[pre + "proj.weight"], sd[pre + "proj.bias"]) * x_mask
    b, c, t = x0.shape
    h = h.reshape(b, c, -1, t).permute(0, 1, 3, 2)
    x1 = rq_spline_inverse(x1, h[..., :num_bins] / hidden ** 0.5, h[..., num_bins:2 * num_bins] / hidden ** 0.5, h[..., 2 * num_bins:], tail_bound)
    return torch.cat([x0, x1], 1) * x_mask


def sdp_reverse(sd, x, x_mask, noise, hidden, kernel_size=3, num_flows=4, g=None, lang_emb=None, noise_scale=1.0):
    """StochasticDurationPredictor.forward(reverse=True) (python/xvapitch/sdp.py:247-276,311-321): log w (B, 1, T) sampled by running
    z = noise * noise_scale backwards through the flows (list reversed, its second-to-last entry dropped, a channel flip BEFORE each flow).
    `noise` (B, 2, T) is the N(0, 1) draw of :313."""
    x = F.conv1d(x, sd["pre.weight"], sd["pre.bias"])
    if g is not None:
        x = x + F.conv1d(g, sd["cond.weight"], sd["cond.bias"])
    if lang_emb is not None:
        x = x + F.conv1d(lang_emb, sd["cond_lang.weight"], sd["cond_lang.bias"])
    x = dds_conv(sd, x, x_mask, None, kernel_size, 3, pre="convs.")
    x = F.conv1d(x, sd["proj.weight"], sd["proj.bias"]) * x_mask
    order = list(reversed(range(num_flows + 1)))                     # flows[4], [3], [2], [1], [0] (= ElementwiseAffine)
    order = order[:-2] + [order[-1]]                                 # :312 "remove a useless vflow"
    z = noise * noise_scale
    for idx in order:
        z = torch.flip(z, [1])
        if idx == 0:
            z = (z - sd["flows.0.translation"]) * torch.exp(-sd["flows.0.log_scale"]) * x_mask       # ElementwiseAffine reverse :112-113
        else:
            z = conv_flow_reverse(sd, z, x_mask, x, hidden, kernel_size, 3, pre="flows.%d." % idx)
    return z[:, :1]


def generate_path(duration, mask):
    """util.py:849-864: duration (B, Tt) whole frames per symbol, mask (B, Tt, Ty) -> the 0 / 1 monotonic path."""
    b, t_x, t_y = mask.shape
    cum = torch.cumsum(duration, 1).reshape(b * t_x)
    path = (torch.arange(t_y)[None, :] < cum[:, None]).to(mask.dtype).reshape(b, t_x, t_y)
    path = path - F.pad(path, (0, 0, 1, 0))[:, :-1]
    return path * mask


def infer(sd, tokens, d_vector, language_id, noise, cfg, decoder, pacing=1.0, pe_scaling=0.1, noise_scale_dp=0.333, length_scale=1.0):
    """xVAPitch.infer (python/xvapitch/model.py:417-599) on the switches xVAPitchModel sets (xva_train.py:1424-1428: --pitch 1, pe_scaling 0.1,
    --energy / --ow_flow / --expanded_flow 0; flc 0, lang_w 1): text encoder -> duration predictor sampled in reverse (noise_scale_dp 0.333,
    model.py:73) -> w = ceil(exp(logw) * mask * length_scale * pacing) -> path -> prior statistics expanded along it -> m_p += pitch_emb(expanded
    pitch prediction) * pe_scaling (:498-515) -> z_p = m_p (inference_noise_scale is set to 0, :549) -> flow in reverse -> waveform decoder.
    tokens (1, Tt) int64, d_vector (dvec,), language_id scalar, noise (1, 2, Tt); decoder(z (1, C, Ty), g (1, dvec, 1)) -> (1, 1, Ty * 256).
    Returns dict(w_ceil, y_lengths, m_p, z, wav)."""
    import math
    Cc = cfg["latent"]

    def sub(pre):
        return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    B, Tt = tokens.shape
    g = F.normalize(d_vector.unsqueeze(0)).unsqueeze(-1)                                                   # _set_cond_input :918
    lang_emb = sd["emb_l.weight"][language_id.reshape(1)].unsqueeze(-1)                                    # :431-432 (lang_w 1)
    te = sub("text_encoder.")
    x_emb = te["emb.weight"][tokens] * math.sqrt(Cc)
    x = torch.cat([x_emb, lang_emb.transpose(2, 1).expand(B, Tt, -1)], -1).transpose(1, 2)
    x_mask = torch.ones(B, 1, Tt)
    x = rel_transformer(sub("text_encoder.encoder."), x * x_mask, x_mask, cfg["heads"], cfg["te_layers"], 3, 4)          # :438
    stats = F.conv1d(x, te["proj.weight"], te["proj.bias"]) * x_mask                                                   # :439
    m_p, logs_p = torch.split(stats, Cc, dim=1)
    logw = sdp_reverse(sub("duration_predictor."), x, x_mask, noise, Cc, 3, 4, g=g, lang_emb=lang_emb, noise_scale=noise_scale_dp)   # :443
    w_ceil = torch.ceil(torch.exp(logw) * x_mask * length_scale * pacing)                                              # :445-447
    y_lengths = torch.clamp_min(torch.sum(w_ceil, [1, 2]), 1).long()                                                   # :452
    Ty = int(y_lengths.max())
    y_mask = torch.ones(B, 1, Ty)
    attn = generate_path(w_ceil.squeeze(1), torch.ones(B, Tt, Ty))                                                     # :455-456
    m_p = torch.matmul(attn.transpose(1, 2), m_p.transpose(1, 2)).transpose(1, 2)                                      # :458
    pin = torch.cat([x.permute(0, 2, 1), g.transpose(2, 1).expand(B, Tt, -1)], -1).transpose(1, -1)                    # :498, model.py:1338-1340
    pitch_pred = rel_transformer(sub("pitch_predictor.encoder."), pin * x_mask, x_mask, cfg["heads"], 3, 3, 4)         # (1, 1, Tt)
    reps = w_ceil.reshape(Tt).long()
    pitch_exp = torch.repeat_interleave(pitch_pred.reshape(Tt), reps).reshape(1, 1, Ty)                                # expand_pitch_energy :935-958
    m_p = m_p + F.conv1d(pitch_exp, sd["pitch_emb.weight"], sd["pitch_emb.bias"], padding=1) * pe_scaling              # :511-515
    z = m_p                                                                                                            # :549-550 (noise scale 0)
    for i in reversed(range(cfg["num_flows"])):                                                                        # ResidualCouplingBlocks reverse :1415-1419
        z = torch.flip(z, [1])
        z = coupling(sub("flow.flows.%d." % i), z, y_mask, g, reverse=True, hidden=Cc, kernel_size=5, dilation_rate=1, num_layers=cfg["flow_layers"])
    wav = decoder(z * y_mask, g)                                                                                       # :593-597
    return {"w_ceil": w_ceil, "y_lengths": y_lengths, "m_p": m_p, "z": z * y_mask, "wav": wav, "logw": logw, "pitch_pred": pitch_pred}


def average_pitch(pitch, durs):
    """model.py:1005-1023: mean of the non-zero frame values under each symbol's duration span.  pitch (B, 1, Ty), durs (B, Tt) -> (B, 1, Tt)."""
    ends = torch.cumsum(durs, dim=1).long()
    starts = F.pad(ends[:, :-1], (1, 0))
    nz = F.pad(torch.cumsum(pitch != 0.0, dim=2), (1, 0))
    cs = F.pad(torch.cumsum(pitch, dim=2), (1, 0))
    dcs, dce = starts[:, None, :], ends[:, None, :]
    sums = (torch.gather(cs, 2, dce) - torch.gather(cs, 2, dcs)).float()
    n = (torch.gather(nz, 2, dce) - torch.gather(nz, 2, dcs)).float()
    return torch.where(n == 0.0, n, sums / n)


def acoustic_losses(sd, tokens, x_lengths, y, y_lengths, d_vectors, language_ids, eps, noise, cfg, pitch_padded=None, pe_scaling=0.1, drop=None):
    """xVAPitch.train_step (python/xvapitch/model.py:681-870) followed by the KL and duration terms of VitsGeneratorLoss.forward
    (losses.py:213-220), on the reference's default switches (--pitch / --energy / --flc / --ow_flow / --mltts_rc 0, detach_dp_input True,
    lang_w 1; `drop(site, tensor)` = nn.Dropout of the text encoder (sites 1000 + 4 layer + k), the pitch predictor (2000 + ...) and the duration
    predictor (3000 + 0..5), None = eval mode) and WITHOUT the waveform decoder / discriminator branch (:852-853 — that branch is the HiFi-GAN path).
    sd: state_dict with the reference's keys (emb_l.*, text_encoder.*, posterior_encoder.*, flow.flows.i.*, duration_predictor.*).
    cfg: latent, lang_dim, heads, te_layers, pe_layers, flow_layers, num_flows.  Returns a dict of the intermediate tensors and losses.
    pitch_padded (B, 1, Ty): the --pitch 1 branch the shipped trainer runs (xva_train.py:1421-1425, pe_scaling 0.1): z_p -= pitch_emb(pitch) *
    pe_scaling (:752-755), per-symbol pitch targets by average_pitch over ceil(durations) (:817-834), the pitch predictor on the detached text
    encoding + speaker vector (:836), and the pitch term of losses.py:224-241 with its broadcast (mask (B, Tt, 1) -> unsqueeze(1) against a
    (B, 1, Tt) error: the product is (B, B, Tt, Tt), so sum / mask.sum() is the UNMASKED sum of squared errors)."""
    import math
    Cc = cfg["latent"]

    def sub(pre):
        return {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    B, Tt = tokens.shape
    g = F.normalize(d_vectors).unsqueeze(-1)                                                              # _set_cond_input :918
    lang_emb = sd["emb_l.weight"][language_ids].unsqueeze(-1)                                              # :695-696 (lang_w = 1)
    z, m_q, logs_q, y_mask = posterior_encoder(sub("posterior_encoder."), y, y_lengths, g, eps, Cc, hidden=Cc, kernel_size=5, dilation_rate=1,
                                               num_layers=cfg["pe_layers"])                                # :698
    te = sub("text_encoder.")
    x_emb = te["emb.weight"][tokens] * math.sqrt(Cc)                                                       # TextEncoder.forward :1152
    x = torch.cat([x_emb, lang_emb.transpose(2, 1).expand(B, Tt, -1)], -1).transpose(1, 2)                 # :1158-1163
    x_mask = (torch.arange(Tt)[None, :] < x_lengths[:, None]).to(x.dtype).unsqueeze(1)
    x = rel_transformer(sub("text_encoder.encoder."), x * x_mask, x_mask, cfg["heads"], cfg["te_layers"], 3, 4, drop=drop, site0=1000)
    stats = F.conv1d(x, te["proj.weight"], te["proj.bias"]) * x_mask                                       # stats=True branch :1148-1150
    m_p, logs_p = torch.split(stats, Cc, dim=1)
    lang_emb = lang_emb.detach()                                                                           # :722
    z_p = z
    for i in range(cfg["num_flows"]):                                                                      # ResidualCouplingBlocks.forward :1406-1420
        z_p = torch.flip(coupling(sub("flow.flows.%d." % i), z_p, y_mask, g, hidden=Cc, kernel_size=5, dilation_rate=1, num_layers=cfg["flow_layers"]), [1])
    if pitch_padded is not None:
        z_p = z_p - F.conv1d(pitch_padded, sd["pitch_emb.weight"], sd["pitch_emb.bias"], padding=1) * pe_scaling   # :752-755
    attn_mask = x_mask.detach().unsqueeze(-1) * y_mask.unsqueeze(2)                                        # :763
    with torch.no_grad():                                                                                  # :765-776
        o_scale = torch.exp(-2 * logs_p)
        logp1 = torch.sum(-0.5 * math.log(2 * math.pi) - logs_p, [1]).unsqueeze(-1)
        logp2 = torch.einsum("klm, kln -> kmn", [o_scale, -0.5 * (z_p ** 2)])
        logp3 = torch.einsum("klm, kln -> kmn", [m_p * o_scale, z_p])
        logp4 = torch.sum(-0.5 * (m_p ** 2) * o_scale, [1]).unsqueeze(-1)
        logp = logp2 + logp3 + logp1 + logp4
        attn = torch.from_numpy(maximum_path(logp.numpy(), attn_mask.squeeze(1).numpy())).to(logp.dtype).unsqueeze(1)
    attn_durations = attn.sum(3)                                                                           # :792
    nll = sdp_forward(sub("duration_predictor."), x.detach(), x_mask, attn_durations, noise, Cc, 3, 4, g=g.detach(), lang_emb=lang_emb, drop=drop,
                      site0=3000)                                                                          # :795-803
    loss_duration = nll / torch.sum(x_mask)                                                                # :814
    m_p_e = torch.einsum("klmn, kjm -> kjn", [attn, m_p])                                                  # :846-847
    logs_p_e = torch.einsum("klmn, kjm -> kjn", [attn, logs_p])
    loss_kl, _ = kl_loss(z_p, logs_q, m_p_e, logs_p_e, y_mask)                                             # losses.py:213
    loss_dur = torch.sum(loss_duration.float())                                                            # losses.py:220
    extra = {}
    loss_pitch = 0.0
    if pitch_padded is not None:
        w_ceil = torch.ceil(attn_durations * x_mask).squeeze(1)                                            # :817-819
        mask = (torch.arange(Tt)[None, :] < x_lengths[:, None])[..., None]                                 # :824
        with torch.no_grad():
            pitch_tgt = average_pitch(pitch_padded, w_ceil)                                                # :829
        pin = torch.cat([x.permute(0, 2, 1).detach(), g.transpose(2, 1).expand(B, Tt, -1)], -1).transpose(1, -1)   # :836, model.py:1338-1340
        pitch_pred = rel_transformer(sub("pitch_predictor.encoder."), pin * x_mask, x_mask, cfg["heads"], 3, 3, 4, drop=drop, site0=2000)
        lp = F.mse_loss(pitch_tgt, pitch_pred, reduction="none") * mask.unsqueeze(1)                       # losses.py:227-228
        loss_pitch = lp.sum() / mask.sum() / pitch_pred.shape[0] * 0.1                                     # :236-241 (pitch_predictor_loss_scale 0.1, :55)
        extra = {"pitch_tgt": pitch_tgt, "pitch_pred": pitch_pred, "loss_pitch": loss_pitch}
    return {**extra, "z": z, "m_q": m_q, "logs_q": logs_q, "x": x, "m_p": m_p_e, "logs_p": logs_p_e, "z_p": z_p, "attn": attn.squeeze(1), "logp": logp,
            "loss_kl": loss_kl, "loss_duration": loss_dur, "loss": loss_kl + loss_dur + loss_pitch}


class HashDrop:
    """The `drop` hooks of this module over the keyed hash of xva-trainer_amd/csrc/xva_common.h (oracle/fastpitch.py HashDropout restates it in
    numpy): drop(site, x) = x * (0 | 1 / (1 - p(site))).  layout "flat": element index = position in x's own contiguous order (the stand-in the
    reference's nn.Dropout is patched with when the goldens are made: oracle/gen_golden_xvapitch_dropout.py); layout "hip": the index the HIP path
    uses for the same element — attention weights (B, H, T, T): flat; a (B, C, T) activation of the transformers: (row of the time-major
    sequence = b * (T + 2 pad) + pad + t) * C + c; a (B, C, T) activation of the duration predictor, held (B, T, C) there: (b * T + t) * C + c."""

    def __init__(self, p_of_site, seed, layout="flat", pad=8, btc_sites=range(3000, 4000)):
        from oracle.fastpitch import HashDropout
        self.p_of_site, self.seed, self.layout, self.pad, self.btc = p_of_site, seed, layout, pad, btc_sites
        self._h = {}
        self._HD = HashDropout

    def __call__(self, site, x):
        import numpy as np
        p = self.p_of_site(site) if callable(self.p_of_site) else self.p_of_site
        if p <= 0:
            return x
        h = self._h.setdefault(p, self._HD(p, self.seed))
        if self.layout == "flat" or x.dim() == 4:
            idx = np.arange(x.numel(), dtype=np.uint64).reshape(tuple(x.shape))
        else:
            B, Cc, T = x.shape
            b, c, t = np.meshgrid(np.arange(B, dtype=np.uint64), np.arange(Cc, dtype=np.uint64), np.arange(T, dtype=np.uint64), indexing="ij")
            if site in self.btc:
                idx = (b * np.uint64(T) + t) * np.uint64(Cc) + c
            else:
                idx = (b * np.uint64(T + 2 * self.pad) + np.uint64(self.pad) + t) * np.uint64(Cc) + c
        return x * torch.from_numpy(h._mult(site, idx)).to(x.dtype)
