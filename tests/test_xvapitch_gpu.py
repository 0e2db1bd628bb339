"""GPU: the xVAPitch-only blocks on libxvahip (xva-trainer_amd/xvapitch: WN gated stack, mean-only ResidualCouplingBlock, maximum_path,
segment, kl_loss) against the vectors recorded from the reference's own modules (tests/golden/xvapitch_blocks.npz) — outputs and every
gradient at 1e-3 in the exact-fp32 mode, index work bit-exact — and against the CPU oracle on other sizes; bf16 mode at a looser bound."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def _g(golden_dir):
    return np.load(os.path.join(golden_dir, "xvapitch_blocks.npz"))


def _sd(g, pre):
    return {k[len(pre):]: torch.from_numpy(g[k]) for k in g.files if k.startswith(pre)}


def _mask(lens, T, device="cpu"):
    return (torch.arange(T, device=device)[None, :] < torch.as_tensor(lens, device=device)[:, None]).float().unsqueeze(1)


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


@pytest.mark.parametrize("compute,tol", [("fp32", RTOL), ("bf16", 3e-2)])
def test_wn_against_reference_golden(golden_dir, compute, tol):
    from xva_trainer_amd.xvapitch.wn import WN
    g = _g(golden_dir)
    B, H, T, CIN, L, K = [int(v) for v in g["wn_cfg"]]
    wn = WN(H, H, K, 1, L, c_in_channels=CIN, compute=compute)
    sd = _sd(g, "wn_sd/")
    assert set(wn.state_dict()) == set(sd) and all(tuple(wn.state_dict()[k].shape) == tuple(v.shape) for k, v in sd.items())
    wn.load_state_dict(sd)
    x = torch.from_numpy(g["wn_x"]).cuda().requires_grad_(True)
    cond = torch.from_numpy(g["wn_g"]).cuda().requires_grad_(True)
    y = wn(x, _mask(g["wn_lens"], T, "cuda"), g=cond)
    assert _rel(y, torch.from_numpy(g["wn_y"])) < tol
    wn.zero_grad()
    (y * torch.from_numpy(g["wn_r"]).cuda()).sum().backward()
    assert _rel(x.grad, torch.from_numpy(g["wn_dx"])) < tol
    assert _rel(cond.grad, torch.from_numpy(g["wn_dg"])) < tol
    mine = wn.grads()
    for k, ref in _sd(g, "wn_grad/").items():
        assert _rel(mine[k], ref) < (tol if compute == "fp32" else 6e-2), k
    if compute == "fp32":
        dead = y[1, :, int(g["wn_lens"][1]):]
        assert float(dead.abs().max()) == 0.0                       # output * x_mask


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
@pytest.mark.parametrize("cond", [True, False], ids=["conditioned", "no conditioning"])
def test_wn_engine_calls_equal_the_per_primitive_sequencing(compute, cond):
    """csrc/xvp_wn.hip (xva_xvp_wn_forward / _backward: the layer loop as two C calls, the weight gradients on the engine's side stream) against the Python
    sequencing of the same kernels (wn.py forward_seq / backward_seq, which the reference golden above pins): output, d x, d g and every parameter gradient."""
    from xva_trainer_amd.xvapitch import wn as wmod
    B, H, T, CIN, L, K = 3, 192, 53, 64, 4, 5
    torch.manual_seed(9)
    lens = np.array([53, 20, 41])
    x0, g0, r = torch.randn(B, H, T).cuda(), torch.randn(B, CIN, 1).cuda(), torch.randn(B, H, T).cuda()
    res = {}
    for engine in (0, 1):
        wn = wmod.WN(H, H, K, 1, L, c_in_channels=CIN if cond else 0, compute=compute, seed=2)
        wn.zero_grad()
        x = x0.clone().requires_grad_(True)
        gg = g0.clone().requires_grad_(True) if cond else None
        old, wmod._WN_ENGINE = wmod._WN_ENGINE, engine
        try:
            y = wn(x, _mask(lens, T, "cuda"), g=gg)
            (y * r).sum().backward()
        finally:
            wmod._WN_ENGINE = old
        torch.cuda.synchronize()
        res[engine] = (y.detach().clone(), x.grad.clone(), gg.grad.clone() if cond else None, {k: v.clone() for k, v in wn.grads().items()})
    (y0, dx0, dg0, gr0), (y1, dx1, dg1, gr1) = res[0], res[1]
    tol = 1e-6 if compute == "fp32" else 1e-6          # same kernels, same operands: only atomically summed vectors (the conditioning gradient) may re-associate
    assert float(y0.abs().max()) > 0 and _rel(y1, y0) < tol and _rel(dx1, dx0) < tol, (_rel(y1, y0), _rel(dx1, dx0))
    if cond:
        assert _rel(dg1, dg0) < 1e-5
    worst = sorted(((_rel(gr1[k], gr0[k]), k) for k in gr0), reverse=True)
    assert worst[0][0] < 1e-5, worst[:4]


def test_wn_engine_weight_gradient_lane_in_a_process_that_switches_it_on():
    """XVA_XVP_WN_LANE=1 (csrc/xvp_wn.hip: the weight-gradient products of the backward loop on the engine's side stream, ordered by events) is decided once per
    process: the engine-against-sequencing test above, re-run in a child process with the switch on."""
    import subprocess, sys
    if os.environ.get("XVA_XVP_WN_LANE") == "1":
        pytest.skip("already inside the child process")
    env = dict(os.environ, XVA_XVP_WN_LANE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k", "test_wn_engine_calls_equal", "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "4 passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]


def test_coupling_against_reference_golden(golden_dir):
    from xva_trainer_amd.xvapitch.wn import ResidualCouplingBlock
    g = _g(golden_dir)
    B, CH, H, T, L, K = [int(v) for v in g["cp_cfg"]]
    blk = ResidualCouplingBlock(CH, H, K, 1, L, mean_only=True, compute="fp32")
    sd = _sd(g, "cp_sd/")
    assert set(blk.state_dict()) == set(sd)
    blk.load_state_dict({k: v.cuda() for k, v in sd.items()})
    m = _mask(g["wn_lens"], T, "cuda")
    x = torch.from_numpy(g["cp_x"]).cuda().requires_grad_(True)
    y, logdet = blk(x, m)
    assert _rel(y, torch.from_numpy(g["cp_y"])) < RTOL and float(logdet.abs().max()) == 0.0
    blk.zero_grad()
    (y * torch.from_numpy(g["cp_r"]).cuda()).sum().backward()
    assert _rel(x.grad, torch.from_numpy(g["cp_dx"])) < RTOL
    mine = blk.grads()
    for k, ref in _sd(g, "cp_grad/").items():
        assert _rel(mine[k], ref) < RTOL, k
    with torch.no_grad():
        rev = blk(torch.from_numpy(g["cp_x"]).cuda(), m, reverse=True)
    assert _rel(rev, torch.from_numpy(g["cp_yrev"])) < RTOL
    with pytest.raises(NotImplementedError):
        ResidualCouplingBlock(CH, H, K, 1, L, mean_only=False)


def test_wn_against_oracle_xvapitch_shapes():
    """The posterior-encoder / flow shape of xVAPitch (hidden 192, kernel 5, 4 layers) on a ragged batch, no conditioning."""
    from oracle import xvapitch as oxv
    from xva_trainer_amd.xvapitch.wn import WN
    torch.manual_seed(3)
    B, H, T, L, K = 3, 192, 210, 4, 5
    wn = WN(H, H, K, 1, L, compute="fp32", seed=5)
    sd = {k: v.cpu() for k, v in wn.state_dict().items()}
    lens = [210, 151, 64]
    x = torch.randn(B, H, T)
    r = torch.randn(B, H, T)
    xo = x.clone().requires_grad_(True)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = oxv.wn(sdo, xo, _mask(lens, T), None, hidden=H, kernel_size=K, dilation_rate=1, num_layers=L)
    (yo * r).sum().backward()
    xg = x.cuda().requires_grad_(True)
    y = wn(xg, _mask(lens, T, "cuda"))
    wn.zero_grad()
    (y * r.cuda()).sum().backward()
    assert _rel(y, yo) < RTOL and _rel(xg.grad, xo.grad) < RTOL
    mine = wn.grads()
    for k in sdo:
        assert _rel(mine[k], sdo[k].grad) < 2e-3, k


def test_maximum_path_segment_kl_against_reference_golden(golden_dir):
    from xva_trainer_amd.xvapitch import ops
    g = _g(golden_dir)
    path = ops.maximum_path(torch.from_numpy(g["mp_value"]).cuda(), torch.from_numpy(g["mp_mask"]).cuda())
    assert np.array_equal(path.cpu().numpy(), g["mp_path"])                                    # index work: bit-exact, ties included
    x = torch.from_numpy(g["sg_x"]).cuda().requires_grad_(True)
    seg = ops.segment(x, torch.from_numpy(g["sg_idx"]).cuda(), 4)
    assert torch.equal(seg.cpu(), torch.from_numpy(g["sg_out"]))
    w = torch.arange(seg.numel(), device="cuda", dtype=torch.float32).reshape(seg.shape)
    (seg * w).sum().backward()
    ref = torch.zeros_like(x)
    for i, s in enumerate(g["sg_idx"]):
        ref[i, :, int(s):int(s) + 4] = w[i]
    assert torch.equal(x.grad, ref)
    t4 = [torch.from_numpy(a).cuda().requires_grad_(True) for a in g["kl_in"]]
    l, sw = ops.kl_loss(*t4, torch.from_numpy(g["kl_mask"]).cuda())
    assert abs(l.item() - float(g["kl_loss"])) < RTOL * abs(float(g["kl_loss"])) and _rel(sw, torch.from_numpy(g["kl_sw"])) < 1e-5
    (l * 1.7).backward()
    for t, ref in zip(t4, g["kl_grads"]):
        assert _rel(t.grad, torch.from_numpy(ref)) < 1e-5


@pytest.mark.parametrize("B,tx,ty", [(1, 1, 1), (5, 64, 300), (8, 150, 860)])
def test_maximum_path_against_oracle(B, tx, ty):
    """Up to the C2 / C5 sizes (150 symbols x 860 frames), ragged lengths, quantised values so that ties occur."""
    from oracle import xvapitch as oxv
    from xva_trainer_amd.xvapitch import ops
    rng = np.random.RandomState(B + tx)
    xl = np.maximum(1, rng.randint(1, tx + 1, size=B)); xl[0] = tx
    yl = np.array([max(int(x), int(rng.randint(1, ty + 1))) for x in xl]); yl = np.minimum(yl, ty); yl[0] = ty
    xl = np.minimum(xl, yl)
    mask = ((np.arange(tx)[None, :, None] < xl[:, None, None]) & (np.arange(ty)[None, None, :] < yl[:, None, None])).astype(np.float32)
    value = (np.round(rng.randn(B, tx, ty) * 4) / 4).astype(np.float32)
    ref = oxv.maximum_path(value, mask)
    out = ops.maximum_path(torch.from_numpy(value).cuda(), torch.from_numpy(mask).cuda()).cpu().numpy()
    assert np.array_equal(out, ref)
    assert np.array_equal(out.sum(1)[mask[:, 0, :] > 0], np.ones(int(mask[:, 0, :].sum())))   # one symbol per live frame


def test_rand_segments_draws_inside_the_clips():
    from xva_trainer_amd.xvapitch import ops
    torch.manual_seed(0)
    x = torch.arange(4 * 3 * 50, dtype=torch.float32).reshape(4, 3, 50).cuda()
    lens = torch.tensor([50, 32, 40, 33], device="cuda")
    seg, idx = ops.rand_segments(x, lens, segment_size=32)
    assert seg.shape == (4, 3, 32) and bool((idx >= 0).all()) and bool((idx + 32 <= lens).all())
    for i in range(4):
        assert torch.equal(seg[i], x[i, :, int(idx[i]):int(idx[i]) + 32])


def test_rand_segments_draws_what_the_reference_draws_and_defers_the_short_clip_check():
    """util.py:165-178: `(torch.rand([B]).type_as(x) * max_idxs).long()` from the CPU generator; a clip shorter than the segment is an AssertionError —
    raised here when the iteration's values are next read (ops.raise_deferred), with the start clamped so that the gather stays inside the tensor."""
    from xva_trainer_amd.xvapitch import ops
    ops.raise_deferred()
    x = torch.randn(4, 3, 50).cuda()
    lens = torch.tensor([50, 32, 40, 33], device="cuda")
    torch.manual_seed(7)
    _, idx = ops.rand_segments(x, lens, segment_size=32)
    torch.manual_seed(7)
    want = (torch.rand([4]) * (lens.cpu() - 32 + 1)).long()
    assert torch.equal(idx.cpu(), want)
    flags = ops.deferred_flags(x.device).cpu()
    assert flags.tolist() == [0.0]
    ops.raise_deferred(flags)                              # nothing set, list emptied
    assert not ops.DEFERRED_CHECKS
    seg, idx = ops.rand_segments(x, torch.tensor([50, 31, 40, 33], device="cuda"), segment_size=32)
    assert seg.shape == (4, 3, 32) and int(idx[1]) == 0
    with pytest.raises(AssertionError, match="shorter than the segment size"):
        ops.raise_deferred()
    assert not ops.DEFERRED_CHECKS


def test_kl_loss_backward_takes_the_upstream_gradient_from_the_device():
    """The loss weight (x upstream gradient) reaches xva_kl_loss_bwd without a host round trip; gradients against autograd on the formula."""
    from xva_trainer_amd.xvapitch import ops
    torch.manual_seed(1)
    B, H, T = 3, 8, 37
    t = [torch.randn(B, H, T, device="cuda").requires_grad_(True) for _ in range(4)]
    mask = (torch.arange(T, device="cuda")[None, None, :] < torch.tensor([37, 20, 5], device="cuda")[:, None, None]).float()
    l, _ = ops.kl_loss(t[0], t[1], t[2], t[3], mask)
    (l * 2.5).backward()
    got = [a.grad.clone() for a in t]
    r = [a.detach().clone().requires_grad_(True) for a in t]
    kl = r[3] - r[1] - 0.5 + 0.5 * (r[0] - r[2]) ** 2 * torch.exp(-2.0 * r[3])
    ((kl * mask).sum() / mask.sum() * 2.5).backward()
    for a, b in zip(got, r):
        assert torch.allclose(a, b.grad, rtol=1e-5, atol=1e-7)


def test_posterior_encoder_against_reference_golden(golden_dir):
    """PosteriorEncoder (model.py:1427-1475): conv1x1 -> conditioned WN -> conv1x1 -> z = (m + eps * exp(logs)) * mask with the reference's own
    N(0, 1) draw; z / m / logs and every gradient of a scalar over all three outputs."""
    from xva_trainer_amd.xvapitch.wn import PosteriorEncoder
    g = _g(golden_dir)
    B, CSP, CO, H, T, L, K, CIN = [int(v) for v in g["pe_cfg"]]
    pe = PosteriorEncoder(CSP, CO, H, K, 1, L, cond_channels=CIN, compute="fp32")
    sd = _sd(g, "pe_sd/")
    assert set(pe.state_dict()) == set(sd)
    pe.load_state_dict({k: v.cuda() for k, v in sd.items()})
    x = torch.from_numpy(g["pe_x"]).cuda().requires_grad_(True)
    cond = torch.from_numpy(g["pe_g"]).cuda().requires_grad_(True)
    z, mean, logs, x_mask = pe(x, torch.from_numpy(g["wn_lens"]).cuda(), g=cond, eps=torch.from_numpy(g["pe_eps"]).cuda())
    assert _rel(z, torch.from_numpy(g["pe_z"])) < RTOL and _rel(mean, torch.from_numpy(g["pe_mean"])) < RTOL and _rel(logs, torch.from_numpy(g["pe_logs"])) < RTOL
    assert x_mask.shape == (B, 1, T) and int(x_mask.sum()) == int(g["wn_lens"].sum())
    pe.zero_grad()
    loss = (z * torch.from_numpy(g["pe_rz"]).cuda()).sum() + (mean * torch.from_numpy(g["pe_rm"]).cuda()).sum() + 0.5 * (logs * torch.from_numpy(g["pe_rl"]).cuda()).sum()
    loss.backward()
    assert _rel(x.grad, torch.from_numpy(g["pe_dx"])) < RTOL and _rel(cond.grad, torch.from_numpy(g["pe_dg"])) < RTOL
    mine = pe.grads()
    for k, ref in _sd(g, "pe_grad/").items():
        assert _rel(mine[k], ref) < RTOL, k


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def test_rel_transformer_against_reference_golden(golden_dir):
    """xvapitch/transformer.py:RelativePositionTransformer (the text encoder's stack: relative-position attention, LayerNorm2, k = 3
    feed-forward convolutions; glow_tts.py:59-485) vs the vectors recorded from the REFERENCE module: output, input gradient and all
    36 parameter gradients (incl. the relative key / value embeddings) at 1e-3, masked positions exactly zero."""
    from xva_trainer_amd.xvapitch.transformer import RelativePositionTransformer
    g = np.load(os.path.join(golden_dir, "xvapitch_transformer.npz"))
    B, Cc, Fh, H, L, K, W, T = (int(v) for v in g["cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1).cuda()
    tr = RelativePositionTransformer(Cc, Cc, Cc, Fh, H, L, kernel_size=K, dropout_p=0.0, rel_attn_window_size=W, layer_norm_type="2")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    assert set(tr.state_dict()) == set(sd)
    tr.load_state_dict(sd)
    tr.zero_grad()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = tr(x, x_mask)
    assert _rel(y, torch.from_numpy(g["y"])) < 1e-3
    dead = y.detach() * (1 - x_mask)
    assert float(dead.abs().max()) == 0.0
    (y * torch.from_numpy(g["r"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(x.grad, torch.from_numpy(g["dx"])) < 1e-3
    grads = tr.grads()
    # conv_k's bias adds the same q_i . b to every score of a row: softmax does not see it, its gradient is zero up to rounding (1e-8 in the
    # reference as well) and has no relative error to speak of
    names = [k[5:] for k in g.files if k.startswith("grad/")]
    for n in names:
        if n.endswith("conv_k.bias"):
            assert float(grads[n].abs().max()) < 1e-5 and float(np.abs(g["grad/" + n]).max()) < 1e-5
    worst = sorted(((_rel(grads[n], torch.from_numpy(g["grad/" + n])), n) for n in names if not n.endswith("conv_k.bias")), reverse=True)
    print("rel transformer worst gradients:", worst[:4])
    assert len(names) == 18 * L and worst[0][0] < 1e-3, worst[:4]


@pytest.mark.parametrize("B,T,Cc,H", [(2, 64, 192, 2), (1, 150, 196, 2), (3, 9, 8, 1)])
def test_rel_transformer_against_oracle(B, T, Cc, H):
    """Other shapes (text-encoder width 192 / 196, a sequence shorter than the attention window, one head) against the CPU oracle."""
    from oracle import xvapitch as oxv
    from xva_trainer_amd.xvapitch.transformer import RelativePositionTransformer
    gen = torch.Generator().manual_seed(5)
    tr = RelativePositionTransformer(Cc, Cc, Cc, 64, H, 2, kernel_size=3, rel_attn_window_size=4, layer_norm_type="2", seed=3)
    sd = tr.state_dict()
    lens = torch.tensor([T] + [max(1, T // (i + 2)) for i in range(B - 1)])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    x = torch.randn(B, Cc, T, generator=gen)
    r = torch.randn(B, Cc, T, generator=gen)
    xo = x.clone().requires_grad_(True)
    leaves = {k: v.cpu().clone().requires_grad_(True) for k, v in sd.items()}
    yo = oxv.rel_transformer(leaves, xo, x_mask, H, 2, 3, 4)
    (yo * r).sum().backward()
    tr.zero_grad()
    xg = x.cuda().requires_grad_(True)
    y = tr(xg, x_mask.cuda())
    (y * r.cuda()).sum().backward()
    assert _rel(y, yo.detach()) < 1e-3 and _rel(xg.grad, xo.grad) < 1e-3
    for k, gr in tr.grads().items():
        if k.endswith("conv_k.bias"):
            assert float(gr.abs().max()) < 1e-4                  # mathematically zero (see above)
        else:
            assert _rel(gr, leaves[k].grad) < 2e-3, k


def test_rel_transformer_pitch_encoder_form_against_reference_golden(golden_dir):
    """The pitch / energy encoders' stack (model.py:1292-1305): out_channels = 1, so `proj` maps the last layer's attention block to one
    channel and the stack returns it (glow_tts.py:479-482); the last layer's feed-forward network and second norm do not reach the output and
    receive no gradient, in the reference as here."""
    from xva_trainer_amd.xvapitch.transformer import RelativePositionTransformer
    g = np.load(os.path.join(golden_dir, "xvapitch_transformer.npz"))
    B, Cc, Fh, H, L, K, W, T = (int(v) for v in g["p_cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1).cuda()
    tr = RelativePositionTransformer(Cc, 1, Cc, Fh, H, L, kernel_size=K, dropout_p=0.0, rel_attn_window_size=W, layer_norm_type="2")
    sd = {k[5:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("p_sd/")}
    assert set(tr.state_dict()) == set(sd) and "proj.weight" in sd
    tr.load_state_dict(sd)
    tr.zero_grad()
    x = torch.from_numpy(g["p_x"]).cuda().requires_grad_(True)
    y = tr(x, x_mask)
    assert tuple(y.shape) == (B, 1, T) and _rel(y, torch.from_numpy(g["p_y"])) < 1e-3
    (y * torch.from_numpy(g["p_r"]).cuda()).sum().backward()
    assert _rel(x.grad, torch.from_numpy(g["p_dx"])) < 1e-3
    grads = tr.grads()
    have = {k[7:] for k in g.files if k.startswith("p_grad/")}
    for n, gr in grads.items():
        if n in have:
            if n.endswith("conv_k.bias"):
                assert float(gr.abs().max()) < 1e-5
            else:
                assert _rel(gr, torch.from_numpy(g["p_grad/" + n])) < 1e-3, n
        else:
            assert float(gr.abs().max()) == 0.0, n + " does not reach the output"
    assert {"proj.weight", "proj.bias"} <= have and "ffn_layers.1.conv_2.weight" not in have


def test_dds_conv_against_reference_golden(golden_dir):
    """xvapitch/sdp.py:DilatedDepthSeparableConv (depthwise dilated conv, LayerNorm2, exact GELU, 1x1 conv: every step a HIP primitive under
    torch autograd; python/xvapitch/sdp.py:40-93) vs the REFERENCE module: output, d x, d g and all 24 parameter gradients at 1e-3."""
    from xva_trainer_amd.xvapitch.sdp import DilatedDepthSeparableConv
    g = np.load(os.path.join(golden_dir, "xvapitch_sdp.npz"))
    B, Cc, T, K, L = (int(v) for v in g["dds_cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1).cuda()
    m = DilatedDepthSeparableConv(Cc, K, L)
    sd = {k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("dds_sd/")}
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd)
    x = torch.from_numpy(g["dds_x"]).cuda().requires_grad_(True)
    cond = torch.from_numpy(g["dds_g"]).cuda().requires_grad_(True)
    y = m(x, x_mask, g=cond)
    assert _rel(y, torch.from_numpy(g["dds_y"])) < 1e-3
    assert float((y.detach() * (1 - x_mask)).abs().max()) == 0.0
    (y * torch.from_numpy(g["dds_r"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(x.grad, torch.from_numpy(g["dds_dx"])) < 1e-3 and _rel(cond.grad, torch.from_numpy(g["dds_dg"])) < 1e-3
    worst = sorted(((_rel(m.p[k[9:]].grad, torch.from_numpy(g[k])), k[9:]) for k in g.files if k.startswith("dds_grad/")), reverse=True)
    print("DDSConv worst gradients:", worst[:3])
    assert len(worst) == 8 * L and worst[0][0] < 1e-3, worst[:3]


@pytest.mark.parametrize("with_g", [True, False], ids=["conditioned", "no conditioning"])
@pytest.mark.parametrize("p_drop", [0.0, 0.5])
def test_dds_engine_calls_equal_the_per_primitive_sequencing(with_g, p_drop):
    """csrc/xvp_dds.hip (xva_xvp_dds_forward / _backward) against the Python sequencing of the same kernels (sdp.py _dds_fwd / _dds_bwd, which the reference golden
    above pins): output, d x, d g and every parameter gradient, with the dropout masks of one seed.  Twice through the backward of the same module: the second
    pass finds the parameters' .grad in place and accumulates into it directly (sdp._gbuf)."""
    from xva_trainer_amd.xvapitch import sdp as smod
    B, T, Cc, K, L = 3, 41, 192, 3, 3
    torch.manual_seed(3)
    lens = torch.tensor([41, 17, 30])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1).cuda()
    x0, g0, r = torch.randn(B, Cc, T).cuda(), torch.randn(B, Cc, T).cuda(), torch.randn(B, Cc, T).cuda()
    res = {}
    for engine in (0, 1):
        m = smod.DilatedDepthSeparableConv(Cc, K, L, dropout_p=p_drop, seed=4, dropout_site_base=7)
        m.drop_seed = 991
        old, smod._DDS_ENGINE = smod._DDS_ENGINE, engine
        try:
            for _ in range(2):
                x = x0.clone().requires_grad_(True)
                cond = g0.clone().requires_grad_(True) if with_g else None
                y = m(x, x_mask, g=cond)
                (y * r).sum().backward()
        finally:
            smod._DDS_ENGINE = old
        torch.cuda.synchronize()
        res[engine] = (y.detach().clone(), x.grad.clone(), cond.grad.clone() if with_g else None, {k: v.grad.clone() for k, v in m.p.items()})
    (y0, dx0, dg0, gr0), (y1, dx1, dg1, gr1) = res[0], res[1]
    assert float(y0.abs().max()) > 0 and _rel(y1, y0) < 1e-6 and _rel(dx1, dx0) < 1e-6
    assert float((y1 * (1 - x_mask)).abs().max()) == 0.0
    if with_g:
        assert _rel(dg1, dg0) < 1e-6
    worst = sorted(((_rel(gr1[k], gr0[k]), k) for k in gr0), reverse=True)
    assert len(worst) == 8 * L and worst[0][0] < 1e-5, worst[:4]


def test_conv_flow_against_reference_golden(golden_dir):
    """xvapitch/sdp.py:ConvFlow (pre, DDSConv with conditioning, proj, the rational-quadratic spline with linear tails — xva_rq_spline_fwd/bwd;
    python/xvapitch/sdp.py:116-176, util.py:203-391) vs the REFERENCE module on inputs that reach into the tails: transformed variable, per-item
    log-determinant, d z, d g and all 28 parameter gradients at 1e-3."""
    from xva_trainer_amd.xvapitch.sdp import ConvFlow
    g = np.load(os.path.join(golden_dir, "xvapitch_sdp.npz"))
    B, Hh, T, K, L, NB = (int(v) for v in g["cf_cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1).cuda()
    m = ConvFlow(2, Hh, K, L, num_bins=NB)
    sd = {k[6:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("cf_sd/")}
    assert set(m.state_dict()) == set(sd)
    m.load_state_dict(sd)
    z = torch.from_numpy(g["cf_z"]).cuda().requires_grad_(True)
    cond = torch.from_numpy(g["cf_g"]).cuda().requires_grad_(True)
    y, ld = m(z, x_mask, g=cond)
    assert _rel(y, torch.from_numpy(g["cf_y"])) < 1e-3 and _rel(ld, torch.from_numpy(g["cf_logdet"])) < 1e-3
    ((y * torch.from_numpy(g["cf_rz"]).cuda()).sum() + (ld * torch.from_numpy(g["cf_rl"]).cuda()).sum()).backward()
    torch.cuda.synchronize()
    assert _rel(z.grad, torch.from_numpy(g["cf_dz"])) < 1e-3 and _rel(cond.grad, torch.from_numpy(g["cf_dg"])) < 1e-3
    worst = sorted(((_rel(m.p[k[8:]].grad, torch.from_numpy(g[k])), k[8:]) for k in g.files if k.startswith("cf_grad/")), reverse=True)
    print("ConvFlow worst gradients:", worst[:3])
    assert len(worst) == 4 + 8 * L and worst[0][0] < 1e-3, worst[:3]


def test_stochastic_duration_predictor_against_reference_golden(golden_dir):
    """xvapitch/sdp.py:StochasticDurationPredictor, training direction (negative log-likelihood of the durations through 2 x (ElementwiseAffine + 4
    ConvFlows), variational dequantisation, speaker and language conditioning; python/xvapitch/sdp.py:179-310) vs the REFERENCE module with the same
    N(0, 1) draw: nll per item, d x, d g, d lang and all 290 parameter gradients at 1e-3."""
    from xva_trainer_amd.xvapitch.sdp import StochasticDurationPredictor
    g = np.load(os.path.join(golden_dir, "xvapitch_sdp.npz"))
    B, Cin, Hs, Cg, Cl, T, K = (int(v) for v in g["sdp_cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1).cuda()
    m = StochasticDurationPredictor(Cin, Hs, K, 0.0, 4, cond_channels=Cg, language_emb_dim=Cl)
    sd = {k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sdp_sd/")}
    assert set(m.state_dict()) == set(sd), sorted(set(m.state_dict()) ^ set(sd))[:8]
    m.load_state_dict(sd)
    x = torch.from_numpy(g["sdp_x"]).cuda().requires_grad_(True)
    gs = torch.from_numpy(g["sdp_g"]).cuda().requires_grad_(True)
    le = torch.from_numpy(g["sdp_lang"]).cuda().requires_grad_(True)
    nll = m(x, x_mask, torch.from_numpy(g["sdp_dr"]).cuda(), g=gs, lang_emb=le, noise=torch.from_numpy(g["sdp_noise"]).cuda())
    assert _rel(nll, torch.from_numpy(g["sdp_nll"])) < 1e-3, (nll, g["sdp_nll"])
    (nll * torch.from_numpy(g["sdp_r"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    assert _rel(x.grad, torch.from_numpy(g["sdp_dx"])) < 1e-3 and _rel(gs.grad, torch.from_numpy(g["sdp_dg"])) < 1e-3
    assert _rel(le.grad, torch.from_numpy(g["sdp_dlang"])) < 1e-3
    worst = sorted(((_rel(m.p[k[9:]].grad, torch.from_numpy(g[k])), k[9:]) for k in g.files if k.startswith("sdp_grad/")), reverse=True)
    print("SDP worst gradients:", worst[:4], "of", len(worst))
    assert len(worst) == len(sd) and worst[0][0] < 1e-3, worst[:4]


def test_residual_coupling_blocks_flow_and_its_inverse():
    """The flow stack (model.py:1358-1422): four mean-only coupling blocks with channel flips, against the block-wise pinned oracle composed
    the same way — output, d x, d g, parameter gradients — and reverse(forward(x)) = x on the unmasked positions (a flow is a bijection)."""
    from oracle import xvapitch as oxv
    from xva_trainer_amd.xvapitch.wn import ResidualCouplingBlocks
    B, CH, H, T, K, L, CIN = 2, 16, 32, 33, 5, 2, 8
    gen = torch.Generator().manual_seed(2)
    fl = ResidualCouplingBlocks(CH, H, K, 1, L, num_flows=4, cond_channels=CIN, seed=11)
    sd = {k: v.cpu() for k, v in fl.state_dict().items()}
    lens = torch.tensor([33, 20])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    x = torch.randn(B, CH, T, generator=gen); g = torch.randn(B, CIN, 1, generator=gen); r = torch.randn(B, CH, T, generator=gen)
    xo, go = x.clone().requires_grad_(True), g.clone().requires_grad_(True)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    y = xo
    for i in range(4):
        sub = {k[len("flows.%d." % i):]: v for k, v in leaves.items() if k.startswith("flows.%d." % i)}
        y = torch.flip(oxv.coupling(sub, y, x_mask, g=go, hidden=H, kernel_size=K, dilation_rate=1, num_layers=L), [1])
    (y * r).sum().backward()
    fl.zero_grad()
    xg, gg = x.cuda().requires_grad_(True), g.cuda().requires_grad_(True)
    out = fl(xg, x_mask.cuda(), g=gg)
    (out * r.cuda()).sum().backward()
    assert _rel(out, y.detach()) < 1e-3 and _rel(xg.grad, xo.grad) < 1e-3 and _rel(gg.grad, go.grad) < 1e-3
    for k, gr in fl.grads().items():
        if leaves[k].grad is not None and float(leaves[k].grad.abs().max()) > 0:
            assert _rel(gr, leaves[k].grad) < 2e-3, k
    with torch.no_grad():
        back = fl(out.detach(), x_mask.cuda(), g=gg.detach(), reverse=True)
    assert _rel(back * x_mask.cuda(), x * x_mask) < 1e-4


@pytest.mark.parametrize("tag", ["", "p_"])
def test_acoustic_train_path_against_reference_train_step_golden(golden_dir, tag):
    """xvapitch/acoustic.py:AcousticTrainPath — embeddings, TextEncoder, PosteriorEncoder (41 spectrogram bins: a channel count that is no multiple
    of 4, like the model's 513), flow, MAS, StochasticDurationPredictor, prior expansion, KL + duration losses; with tag p_ also the --pitch 1
    branch the shipped trainer runs (pitch_emb on z_p, average_pitch targets, pitch predictor, pitch loss) — against the vectors recorded from
    the REFERENCE's own xVAPitch.train_step (model.py:681-870) with the same two N(0, 1) draws: outputs and losses at 1e-3, the MAS path
    bit-exact, every parameter gradient at 2e-3 (relative L2; the second run's non-pitch tensors by their norms and 256 samples each)."""
    from oracle import golden_util
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    g = np.load(os.path.join(golden_dir, "xvapitch_acoustic.npz"))
    c = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    m = AcousticTrainPath(c["vocab"], c["langs"], latent_size=c["latent"], embedded_language_dim=c["lang_dim"], d_vector_dim=c["dvec"],
                          hidden_channels_ffn=c["ffn"], num_heads=c["heads"], text_layers=c["te_layers"], posterior_layers=c["pe_layers"],
                          flow_layers=c["flow_layers"], num_flows=c["num_flows"], spec_bins=c["spec_bins"], pitch=bool(tag))
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")}
    grads_all = [k[5:] for k in g.files if k.startswith("grad/")]
    want = set(k for k in grads_all if tag or not k.startswith("pitch_"))
    assert set(m.state_dict()) == want, sorted(set(m.state_dict()) ^ want)[:8]
    m.load_state_dict(sd)
    t = lambda k: torch.from_numpy(g[k]).cuda()
    m.zero_grad()
    o = m(t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("dvec"), t("lids"), eps=t(tag + "eps"), noise=t(tag + "noise"),
          pitch_padded=t("pitch") if tag else None)
    for k in [f[len(tag) + 4:] for f in g.files if f.startswith(tag + "out/")]:
        assert _rel(o[k], torch.from_numpy(g[tag + "out/" + k])) < 1e-3, (k, _rel(o[k], torch.from_numpy(g[tag + "out/" + k])))
    assert np.array_equal(o["attn"].cpu().numpy().astype(np.uint8), g[tag + "attn"])
    for k in ("loss_kl", "loss_duration") + (("loss_pitch",) if tag else ()):
        assert abs(float(o[k]) - float(g[tag + k])) < 1e-3 * abs(float(g[tag + k])), (k, float(o[k]), float(g[tag + k]))
    o["loss"].backward()
    torch.cuda.synchronize()
    mine = {k: v.detach().cpu() for k, v in m.grads().items()}
    worst = []
    for k in [f[len(tag) + 5:] for f in g.files if f.startswith(tag + "grad/")]:
        ref = torch.from_numpy(g[tag + "grad/" + k])
        if k not in mine:
            continue
        if float(ref.norm()) < 1e-5 * ref.numel() ** 0.5:                          # mathematically zero (conv_k.bias, unused last FFN): rounding noise on both sides
            assert float(mine[k].norm()) < 1e-3, k
            continue
        worst.append((_rel(mine[k].reshape(ref.shape), ref), k))
    worst.sort(reverse=True)
    print("acoustic path%s worst gradients:" % (" (pitch)" if tag else ""), worst[:5], "of", len(worst))
    assert len(worst) > (30 if tag else 400) and worst[0][0] < 2e-3, worst[:5]
    if tag:
        keys = [str(k) for k in g["p_grad_keys"]]
        live = set(k for k, nr in zip(keys, g["p_grad_norms"]) if nr >= 1e-5 * mine[k].numel() ** 0.5)
        errs = [e for e in golden_util.check_samples(mine, keys, g["p_grad_samples"], g["p_grad_offsets"], 256) if e[1] in live]
        print("sampled:", errs[:3], "of", len(errs))
        assert len(errs) > 450 and errs[0][0] < 2e-3, errs[:4]
        for k, nr in zip(keys, g["p_grad_norms"]):
            assert abs(float(mine[k].norm()) - float(nr)) < 2e-3 * float(nr) + 1e-4, k


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_vits_decoder_against_reference_golden(golden_dir, compute):
    """xvapitch/decoder.py:VitsDecoder (xva_vits_dec_forward / _backward: the HiFi-GAN v1 generator engine with a 192-channel latent input, plain
    conv_pre / conv_post, no conv_post bias, cond_layer on the speaker vector) against the vectors recorded from the REFERENCE HifiganGenerator
    (python/xvapitch/hifigan.py:156-262 built as model.py:134-149).  fp32: waveform 1e-4; d z and all 233 parameter gradients 1e-2 — the bound the
    fixture's generator derives (LeakyReLU gates within rounding of zero flip between any two fp32 evaluations; the same CPU restatement in fp64
    moves its own gradients by 2-4e-3).  bf16 storage: the waveform at 3e-2, gradient norms at 15 %, d z at 0.35 relative L2 (measured 9e-3 / 7 % / 0.20: end-to-end
    through ~50 bf16-stored layers against a random cotangent; the tight, teacher-forced per-layer bound of the shared engine is in test_hifigan_gpu.py)."""
    from oracle import golden_util, hifigan as ohg
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    g = np.load(os.path.join(golden_dir, "vits_decoder.npz"))
    seed, B, Cin, Cc, T = (int(v) for v in g["cfg"])
    sd = ohg.init_vits_decoder_sd(seed, Cin, Cc)
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["sd_checksum"])) < 1e-3
    dec = VitsDecoder(Cin, Cc, compute=compute)
    assert set(dec.state_dict()) == set(sd), sorted(set(dec.state_dict()) ^ set(sd))[:8]
    dec.load_state_dict(sd)
    z = torch.from_numpy(g["z"]).cuda().requires_grad_(True)
    gv = torch.from_numpy(g["g"]).cuda().unsqueeze(-1)
    y = dec(z, gv)
    ey = _rel(y, torch.from_numpy(g["y"]))
    dec.zero_grad()
    (y * torch.from_numpy(g["r"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    mine = {k: v.detach().cpu() for k, v in dec.grads().items()}
    keys = [str(k) for k in g["grad_keys"]]
    edz = _rel(z.grad, torch.from_numpy(g["dz"]))
    if compute == "fp32":
        errs = golden_util.check_samples(mine, keys, g["grad_samples"], g["grad_offsets"], 512)
        full = sorted(((_rel(mine[k[5:]], torch.from_numpy(g[k])), k[5:]) for k in g.files if k.startswith("grad/")), reverse=True)
        print("vits decoder fp32: y %.2e dz %.2e sampled worst %s full worst %s" % (ey, edz, errs[:3], full[:3]))
        assert ey < 1e-4 and edz < 1e-2
        assert len(errs) == 233 and errs[0][0] < 1e-2, errs[:4]
        assert full[0][0] < 1e-2, full[:4]
    else:
        nerr = sorted(((abs(float(mine[k].norm()) - float(n)) / float(n), k) for k, n in zip(keys, g["grad_norms"])), reverse=True)
        print("vits decoder bf16: y %.2e dz %.2e worst norm errors %s" % (ey, edz, nerr[:3]))
        assert ey < 3e-2 and edz < 0.35 and nerr[0][0] < 0.15, (ey, edz, nerr[:3])


def test_vits_decoder_without_conditioning_and_data_only_input():
    """cond_channels = 0 (no cond_layer tensors), a 256-channel latent (the `big` model), an input that does not require grad: the parameter
    gradients are still produced; against the CPU oracle."""
    from oracle import hifigan as ohg
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    sd = ohg.init_vits_decoder_sd(5, 256, 0)
    dec = VitsDecoder(256, 0)
    dec.load_state_dict(sd)
    gen = torch.Generator().manual_seed(6)
    z = torch.randn(1, 256, 8, generator=gen); r = torch.randn(1, 1, 2048, generator=gen)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = ohg.vits_decoder(leaves, z, None)
    (yo * r).sum().backward()
    dec.zero_grad()
    y = dec(z.cuda(), None)
    (y * r.cuda()).sum().backward()
    assert _rel(y, yo.detach()) < 1e-4
    worst = sorted(((_rel(v, leaves[k].grad), k) for k, v in dec.grads().items()), reverse=True)
    assert worst[0][0] < 1e-2, worst[:4]


def test_generator_pass_against_reference_train_step_golden(golden_dir):
    """xvapitch/generator_pass.py:GeneratorPass — AcousticTrainPath (--pitch 1) + rand_segments + VitsDecoder + segment + TorchSTFT-mel L1 x 45 —
    against the vectors recorded from the REFERENCE's own train_step (model.py:681-870) with the reference HifiganGenerator as waveform decoder and
    the non-adversarial terms of VitsGeneratorLoss (losses.py:187-241), same three random draws: the generated segment 1e-3, the four losses
    1e-3, and d(loss)/d(every parameter) of the acoustic modules AND the decoder (709 tensors, norms + 256 samples each; 7 in full) at 1e-2 — the
    LeakyReLU-gate bound of the decoder's backward (oracle/gen_golden_vits_decoder.py); the decoder gradient reaches the posterior encoder through d z."""
    from oracle import golden_util, hifigan as ohg
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
    g = np.load(os.path.join(golden_dir, "xvapitch_genpass.npz"))
    c = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    ac = AcousticTrainPath(c["vocab"], c["langs"], latent_size=c["latent"], embedded_language_dim=c["lang_dim"], d_vector_dim=c["dvec"],
                           hidden_channels_ffn=c["ffn"], num_heads=c["heads"], text_layers=c["te_layers"], posterior_layers=c["pe_layers"],
                           flow_layers=c["flow_layers"], num_flows=c["num_flows"], spec_bins=c["spec_bins"], pitch=True)
    ac.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
    dec = VitsDecoder(c["latent"], c["dvec"])
    dsd = ohg.init_vits_decoder_sd(int(g["dec_seed"]), c["latent"], c["dvec"])
    assert abs(sum(float(v.double().sum()) for v in dsd.values()) - float(g["dec_checksum"])) < 1e-3
    dec.load_state_dict(dsd)
    gp = GeneratorPass(ac, dec, spec_segment_size=int(g["seg"]))
    t = lambda k: torch.from_numpy(g[k]).cuda()
    gp.zero_grad()
    o = gp(t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("wav"), t("dvec"), t("lids"), pitch_padded=t("pitch"), eps=t("eps"), noise=t("noise"),
           slice_ids=t("slice_ids"))
    assert _rel(o["model_outputs"], torch.from_numpy(g["model_outputs"])) < 1e-3 and _rel(o["z_p"], torch.from_numpy(g["z_p"])) < 1e-3
    for k in ("loss_mel", "loss_kl", "loss_duration", "loss_pitch"):
        assert abs(float(o[k]) - float(g[k])) < 1e-3 * abs(float(g[k])), (k, float(o[k]), float(g[k]))
    o["loss"].backward()
    torch.cuda.synchronize()
    mine = {k: v.detach().cpu() for k, v in ac.grads().items()}
    mine.update({"waveform_decoder." + k: v.detach().cpu() for k, v in dec.grads().items()})
    keys = [str(k) for k in g["grad_keys"]]
    assert set(keys) == set(mine), sorted(set(keys) ^ set(mine))[:8]
    live = set(k for k, nr in zip(keys, g["grad_norms"]) if nr >= 1e-5 * mine[k].numel() ** 0.5)
    errs = [e for e in golden_util.check_samples(mine, keys, g["grad_samples"], g["grad_offsets"], 256) if e[1] in live]
    full = sorted(((_rel(mine[k[5:]], torch.from_numpy(g[k])), k[5:]) for k in g.files if k.startswith("grad/")), reverse=True)
    print("generator pass: sampled worst", errs[:4], "of", len(errs), "full worst", full[:3])
    assert len(errs) > 690 and errs[0][0] < 1e-2, errs[:4]
    assert full[0][0] < 1e-2, full[:3]
    # the draw path: without slice_ids the starts are drawn like the reference's rand_segments and stay inside the clips
    o2 = gp(t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("wav"), t("dvec"), t("lids"), pitch_padded=t("pitch"))
    ids = o2["slice_ids"].cpu()
    assert bool((ids >= 0).all()) and bool((ids + int(g["seg"]) <= torch.from_numpy(g["y_lens"])).all())


@pytest.mark.parametrize("compute", ["fp32", "bf16"])
def test_vits_discriminator_against_reference_golden(golden_dir, compute):
    """xvapitch/discriminator.py:VitsDiscriminator (xva_vits_disc_*: five period discriminators + the scale discriminator whose grouped k = 41
    convolutions — four input channels per group — run as dense products over block-diagonal weights) against the vectors recorded from the
    REFERENCE VitsDiscriminator and loss functions (model.py:1548-1640, losses.py:64-84,331-343).  fp32: the three losses 1e-4, d(loss_gen + loss_feat)/d y_hat of the G pass 1e-3, the 111 parameter
    gradients of the D pass 1e-3 except a handful of the first period discriminator's (≤ 5e-3 on 512 samples: LeakyReLU gates within rounding
    of zero, as for the decoder; the scale discriminator's land at 5e-7).  bf16: losses 2e-2, gradient norms 10 %."""
    from oracle import golden_util, hifigan as ohg
    from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
    g = np.load(os.path.join(golden_dir, "vits_disc.npz"))
    seed, B, seg = (int(v) for v in g["cfg"])
    sd = ohg.init_vits_disc_sd(seed)
    assert abs(sum(float(v.double().sum()) for v in sd.values()) - float(g["sd_checksum"])) < 1e-3
    D = VitsDiscriminator(compute=compute)
    assert list(D.state_dict()) == list(sd)                     # same tensors in the reference's order
    D.load_state_dict(sd)
    y, yh = torch.from_numpy(g["y"]).cuda(), torch.from_numpy(g["y_hat"]).cuda()
    D.zero_grad()
    loss_disc = float(D.d_pass(y, yh))
    loss_gen, loss_feat, d_wav = D.g_pass(y, yh)
    torch.cuda.synchronize()
    mine = {k: v.detach().cpu() for k, v in D.grads().items()}
    keys = [str(k) for k in g["grad_keys"]]
    rel = lambda a, b: abs(a - b) / abs(b)
    el = max(rel(loss_disc, float(g["loss_disc"])), rel(float(loss_gen), float(g["loss_gen"])), rel(float(loss_feat), float(g["loss_feat"])))
    edw = _rel(d_wav, torch.from_numpy(g["d_wav"]))
    if compute == "fp32":
        errs = golden_util.check_samples(mine, keys, g["grad_samples"], g["grad_offsets"], 512)
        full = sorted(((_rel(mine[k[5:]], torch.from_numpy(g[k])), k[5:]) for k in g.files if k.startswith("grad/")), reverse=True)
        print("vits disc fp32: losses %.2e d_wav %.2e sampled worst %s full worst %s" % (el, edw, errs[:3], full[:3]))
        assert el < 1e-4 and edw < 1e-3
        assert len(errs) == 111 and errs[0][0] < 5e-3 and errs[4][0] < 1e-3, errs[:6]
        assert full[0][0] < 2e-3, full[:4]
    else:
        nerr = sorted(((abs(float(mine[k].norm()) - float(n)) / float(n), k) for k, n in zip(keys, g["grad_norms"])), reverse=True)
        print("vits disc bf16: losses %.2e d_wav %.2e worst norm errors %s" % (el, edw, nerr[:3]))
        assert el < 2e-2 and edw < 0.1 and nerr[0][0] < 0.1, (el, edw, nerr[:3])


@pytest.mark.parametrize("eager_disc", [False, True], ids=["reference order", "discriminator pass on the branch stream"])
def test_c5_generator_and_discriminator_passes_against_reference_golden(golden_dir, eager_disc):
    """xvapitch/train_step.py:XVAPitchStep — BOTH passes of one xVAPitch iteration (BASELINE config C5) — against the vectors recorded from the
    reference's own code (oracle/gen_golden_xvapitch_c5.py: train_step + HifiganGenerator + VitsDiscriminator + the loss functions, assembled as
    model.py:272-384 / losses.py:187-300 do): the six generator-side losses, their total and loss_disc at 1e-3; d(total)/d(every generator
    parameter) (709 tensors: text encoder, posterior encoder, flow, duration + pitch predictors, embeddings, decoder) and d(loss_disc)/d(every
    discriminator parameter) (111) at 1e-2 on norms and 256 samples each (the LeakyReLU-gate bound of the decoder / discriminators)."""
    from oracle import golden_util, hifigan as ohg
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
    from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
    from xva_trainer_amd.xvapitch.train_step import XVAPitchStep
    g = np.load(os.path.join(golden_dir, "xvapitch_genpass.npz"))
    g5 = np.load(os.path.join(golden_dir, "xvapitch_c5.npz"))
    c = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    ac = AcousticTrainPath(c["vocab"], c["langs"], latent_size=c["latent"], embedded_language_dim=c["lang_dim"], d_vector_dim=c["dvec"],
                           hidden_channels_ffn=c["ffn"], num_heads=c["heads"], text_layers=c["te_layers"], posterior_layers=c["pe_layers"],
                           flow_layers=c["flow_layers"], num_flows=c["num_flows"], spec_bins=c["spec_bins"], pitch=True)
    ac.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
    dec = VitsDecoder(c["latent"], c["dvec"])
    dec.load_state_dict(ohg.init_vits_decoder_sd(int(g["dec_seed"]), c["latent"], c["dvec"]))
    D = VitsDiscriminator()
    dsd = ohg.init_vits_disc_sd(int(g5["disc_seed"]))
    assert abs(sum(float(v.double().sum()) for v in dsd.values()) - float(g5["disc_checksum"])) < 1e-3
    D.load_state_dict(dsd)
    step = XVAPitchStep(GeneratorPass(ac, dec, spec_segment_size=int(g["seg"])), D)
    t = lambda k: torch.from_numpy(g[k]).cuda()
    step.gen.zero_grad(); D.zero_grad()
    o = step.generator_pass(t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("wav"), t("dvec"), t("lids"), pitch_padded=t("pitch"), eps=t("eps"),
                            noise=t("noise"), slice_ids=t("slice_ids"), eager_disc=eager_disc)
    assert (step._eager is not None) == eager_disc          # the trainer's order: the discriminator pass has already run, inside the generator pass
    if eager_disc:
        step.gen.join()                                     # values are read BEFORE the backward pass here: the branch's stream has not been joined yet (late_join)
    assert _rel(o["model_outputs"], torch.from_numpy(g5["model_outputs"])) < 1e-3
    for k in ("loss_mel", "loss_kl", "loss_duration", "loss_pitch", "loss_gen", "loss_feat", "loss"):
        assert abs(float(o[k]) - float(g5[k])) < 1e-3 * abs(float(g5[k])), (k, float(o[k]), float(g5[k]))
    o["loss"].backward()
    loss_disc = step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
    torch.cuda.synchronize()
    assert abs(float(loss_disc) - float(g5["loss_disc"])) < 1e-3 * float(g5["loss_disc"]), (float(loss_disc), float(g5["loss_disc"]))

    def check(mine, tag, nmin):
        keys = [str(k) for k in g5[tag + "_keys"]]
        assert set(keys) == set(mine), sorted(set(keys) ^ set(mine))[:8]
        live = set(k for k, nr in zip(keys, g5[tag + "_norms"]) if nr >= 1e-5 * mine[k].numel() ** 0.5)
        errs = [e for e in golden_util.check_samples(mine, keys, g5[tag + "_samples"], g5[tag + "_offsets"], 256) if e[1] in live]
        nerr = sorted(((abs(float(mine[k].norm()) - float(nr)) / float(nr), k) for k, nr in zip(keys, g5[tag + "_norms"]) if k in live), reverse=True)
        print("C5 %s-pass gradients: sampled worst %s ; norm worst %s ; %d tensors" % (tag, errs[:3], nerr[:2], len(errs)))
        assert len(errs) >= nmin and errs[0][0] < 1e-2 and nerr[0][0] < 1e-2, (errs[:4], nerr[:4])
    mine = {k: v.detach().cpu() for k, v in ac.grads().items()}
    mine.update({"waveform_decoder." + k: v.detach().cpu() for k, v in dec.grads().items()})
    check(mine, "g", 690)
    check({k: v.detach().cpu() for k, v in D.grads().items()}, "d", 111)


def test_rel_transformer_mixed_mode_against_reference_golden(golden_dir):
    """compute="mixed" (the throughput mode of the two transformers: bf16 MFMA on the fp32-stored operands of the projections and feed-forward
    convolutions; attention and LayerNorm fp32) against the same reference vectors: output 2e-2 (measured 1.6e-3), input gradient 5e-2 (1.2e-2), parameter gradients 0.1 (worst 5.9e-2: a feed-forward bias, a sum of
    near-cancelling bf16-rounded products)."""
    from xva_trainer_amd.xvapitch.transformer import RelativePositionTransformer
    g = np.load(os.path.join(golden_dir, "xvapitch_transformer.npz"))
    B, Cc, Fh, H, L, K, W, T = (int(v) for v in g["cfg"])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1).cuda()
    tr = RelativePositionTransformer(Cc, Cc, Cc, Fh, H, L, kernel_size=K, dropout_p=0.0, rel_attn_window_size=W, layer_norm_type="2", compute="mixed")
    tr.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
    tr.zero_grad()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    y = tr(x, x_mask)
    (y * torch.from_numpy(g["r"]).cuda()).sum().backward()
    torch.cuda.synchronize()
    grads = tr.grads()
    names = [k[5:] for k in g.files if k.startswith("grad/") and not k.endswith("conv_k.bias")]
    worst = sorted(((_rel(grads[n], torch.from_numpy(g["grad/" + n])), n) for n in names), reverse=True)
    ey, edx = _rel(y, torch.from_numpy(g["y"])), _rel(x.grad, torch.from_numpy(g["dx"]))
    print("rel transformer (mixed): y %.2e dx %.2e worst gradients %s" % (ey, edx, worst[:3]))
    assert 1e-6 < ey < 2e-2 and edx < 5e-2 and worst[0][0] < 0.1, (ey, edx, worst[:4])      # > 1e-6: the bf16 products really ran


@pytest.mark.parametrize("cfg", [dict(C=196, Co=196, F=768, H=2, L=3, k=3, compute="fp32"), dict(C=196, Co=196, F=768, H=2, L=2, k=3, compute="mixed"),
                                 dict(C=708, Co=1, F=768, H=2, L=3, k=3, compute="mixed"), dict(C=64, Co=32, F=128, H=4, L=2, k=1, compute="fp32")],
                         ids=["text encoder fp32", "text encoder mixed", "pitch predictor (proj to 1 channel)", "proj to 32 channels, k 1"])
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_transformer_engine_calls_equal_the_per_primitive_sequencing(cfg, p_drop):
    """csrc/xvp_transformer.hip (xva_xvp_tr_forward / _backward: the whole stack as two C calls) against the Python sequencing of the same kernels
    (transformer.py forward_seq / backward_seq, which the reference goldens pin): output, input gradient and every parameter gradient, with the
    dropout masks of the same seed.  Same kernels, same summation order: equal to fp32 rounding of the few re-associated adds."""
    from xva_trainer_amd.xvapitch import transformer as tmod
    B, T = 3, 37
    torch.manual_seed(5)
    lens = torch.tensor([37, 20, 9])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1).cuda()
    x0 = torch.randn(B, cfg["C"], T).cuda()
    r = torch.randn(B, cfg["Co"], T).cuda()
    res = {}
    for engine in (False, True):
        tr = tmod.RelativePositionTransformer(cfg["C"], cfg["Co"], cfg["C"], cfg["F"], cfg["H"], cfg["L"], kernel_size=cfg["k"], dropout_p=p_drop,
                                              rel_attn_window_size=4, layer_norm_type="2", compute=cfg["compute"], seed=11, dropout_site_base=40)
        tr.set_dropout_seed(777)
        tr.zero_grad()
        x = x0.clone().requires_grad_(True)
        old, tmod._ENGINE = tmod._ENGINE, engine
        try:
            y = tr(x, x_mask)
            (y * r).sum().backward()
        finally:
            tmod._ENGINE = old
        torch.cuda.synchronize()
        res[engine] = (y.detach().clone(), x.grad.clone(), {k: v.clone() for k, v in tr.grads().items()})
    (y0, dx0, g0), (y1, dx1, g1) = res[False], res[True]
    assert float(y0.abs().max()) > 0 and float(dx0.abs().max()) > 0
    # The engine splits its long-K products along K (fp32 re-association, 3e-7 on the product).  In "mixed" mode every product rounds its operands to bf16:
    # a last-bit difference upstream flips roundings downstream, so two correct evaluations agree to bf16 noise, not to fp32 noise.
    ty, tg = (1e-5, 1e-5) if cfg["compute"] == "fp32" else (2e-3, 2e-2)
    assert _rel(y1, y0) < ty and _rel(dx1, dx0) < tg, (_rel(y1, y0), _rel(dx1, dx0))
    # conv_k.bias: the softmax does not see a bias of the keys — its gradient is rounding noise around zero (the reference golden test skips it too)
    worst = sorted(((_rel(g1[k], g0[k]) if float(g0[k].abs().max()) > 0 else float(g1[k].abs().max()), k) for k in g0 if not k.endswith("conv_k.bias")), reverse=True)
    assert set(g0) == set(g1) and worst[0][0] < tg, worst[:4]
    assert all(float(g1[k].abs().max()) < 1e-3 * float(g1[k.replace("conv_k", "conv_q")].abs().max()) for k in g1 if k.endswith("conv_k.bias"))
    dead = [k for k in g0 if float(g0[k].abs().max()) == 0]
    assert all(float(g1[k].abs().max()) == 0 for k in dead)            # out_channels == 1: the last layer's feed-forward / norm2 get no gradient in either


def test_c5_optimizer_step_matches_torch_adamw(golden_dir):
    """XVAPitchStep.optimizer_step (three flat xva_adamw_step launches) against torch.optim.AdamW — the class the reference trainer constructs
    (python/xvapitch/training_util.py:56-57: betas 0.8 / 0.99, eps 1e-9, weight decay 0.01) — run on the CPU over the same parameters and the
    gradients the two passes produced: every parameter of the acoustic modules, the decoder and the discriminator after two steps at 1e-6."""
    from oracle import hifigan as ohg
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
    from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
    from xva_trainer_amd.xvapitch.train_step import XVAPitchStep
    g = np.load(os.path.join(golden_dir, "xvapitch_genpass.npz"))
    c = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    ac = AcousticTrainPath(c["vocab"], c["langs"], latent_size=c["latent"], embedded_language_dim=c["lang_dim"], d_vector_dim=c["dvec"],
                           hidden_channels_ffn=c["ffn"], num_heads=c["heads"], text_layers=c["te_layers"], posterior_layers=c["pe_layers"],
                           flow_layers=c["flow_layers"], num_flows=c["num_flows"], spec_bins=c["spec_bins"], pitch=True)
    ac.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
    dec = VitsDecoder(c["latent"], c["dvec"]); dec.load_state_dict(ohg.init_vits_decoder_sd(int(g["dec_seed"]), c["latent"], c["dvec"]))
    D = VitsDiscriminator(); D.load_state_dict(ohg.init_vits_disc_sd(3))
    step = XVAPitchStep(GeneratorPass(ac, dec, spec_segment_size=int(g["seg"])), D)
    t = lambda k: torch.from_numpy(g[k]).cuda()
    mods = (("ac", ac), ("dec", dec), ("D", D))
    ref_p = {n + "/" + k: v.detach().cpu().clone().requires_grad_(True) for n, m in mods for k, v in m.state_dict().items()}
    opt_g = torch.optim.AdamW([v for k, v in ref_p.items() if not k.startswith("D/")], lr=1e-3, betas=[0.8, 0.99], eps=1e-09, weight_decay=0.01)
    opt_d = torch.optim.AdamW([v for k, v in ref_p.items() if k.startswith("D/")], lr=2e-4, betas=[0.8, 0.99], eps=1e-09, weight_decay=0.01)
    for it in range(2):
        step.gen.zero_grad(); D.zero_grad()
        o = step.generator_pass(t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("wav"), t("dvec"), t("lids"), pitch_padded=t("pitch"), eps=t("eps"),
                                noise=t("noise"), slice_ids=t("slice_ids"))
        o["loss"].backward()
        step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
        torch.cuda.synchronize()
        for n, m in mods:
            for k, gr in m.grads().items():
                # the pitch predictor's last feed-forward block / second LayerNorm never reach its output: p.grad stays None in the reference
                dead = n == "ac" and k.startswith(("pitch_predictor.encoder.ffn_layers.2.", "pitch_predictor.encoder.norm_layers_2.2."))
                ref_p[n + "/" + k].grad = None if dead else gr.detach().cpu().clone().reshape(ref_p[n + "/" + k].shape)
        opt_g.step(); opt_d.step()
        step.optimizer_step(lr=1e-3, lr_disc=2e-4)
    torch.cuda.synchronize()
    worst = sorted(((float((v.detach().cpu() - ref_p[n + "/" + k].detach()).abs().max() / ref_p[n + "/" + k].detach().abs().max().clamp_min(1e-12)), n + "/" + k)
                    for n, m in mods for k, v in m.state_dict().items()), reverse=True)
    print("C5 AdamW: worst max-abs error relative to the tensor's max after 2 steps:", worst[:3], "of", len(worst))
    assert len(worst) > 800 and worst[0][0] < 1e-5, worst[:4]


def test_c5_properties_at_reference_model_size():
    """Size-independent properties at the reference's model size (latent 192, speaker vector 512, 513 spectrogram bins, B = 4 x 100 symbols x 400
    frames): the flow is a bijection (reverse(forward(z)) = z on the unmasked frames), the alignment the acoustic path searches is a monotonic
    path (one symbol per frame, non-decreasing, every symbol visited, durations summing to the frame count), and the decoder's backward is
    linear in the cotangent (gradients for 2 r = 2 x gradients for r)."""
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    B, Tt, Ty = 4, 100, 400
    gen = torch.Generator().manual_seed(3)
    ac = AcousticTrainPath(256, 31, pitch=True)                                                # the reference's layer counts (10 text-encoder / 16 posterior-encoder layers)
    x_lens = torch.tensor([100, 77, 60, 31]); y_lens = torch.tensor([400, 333, 251, 140])
    tokens = (torch.randint(1, 256, (B, Tt), generator=gen) * (torch.arange(Tt)[None, :] < x_lens[:, None])).cuda()
    y = (torch.rand(B, 513, Ty, generator=gen) * (torch.arange(Ty)[None, None, :] < y_lens[:, None, None])).cuda()
    dvec = torch.randn(B, 512, generator=gen).cuda(); lids = torch.randint(0, 31, (B,), generator=gen).cuda()
    pitch = ((torch.rand(B, 1, Ty, generator=gen) * 3 - 1.2).clamp_min(0) * (torch.arange(Ty)[None, None, :] < y_lens[:, None, None])).cuda()
    o = ac(tokens, x_lens.cuda(), y, y_lens.cuda(), dvec, lids, pitch_padded=pitch)
    assert all(torch.isfinite(o[k]).all() for k in ("loss_kl", "loss_duration", "loss_pitch", "z_p"))
    attn = o["attn"].cpu()                                                                      # (B, Tt, Ty)
    for b in range(B):
        a = attn[b, :x_lens[b], :y_lens[b]]
        assert bool((a.sum(0) == 1).all()) and bool((a.sum(1) >= 1).all())                      # one symbol per frame, every symbol used
        idx = a.argmax(0)
        assert bool((idx[1:] - idx[:-1] >= 0).all()) and bool((idx[1:] - idx[:-1] <= 1).all()) and int(idx[0]) == 0 and int(idx[-1]) == int(x_lens[b]) - 1
        assert float(attn[b].sum()) == float(y_lens[b])
    g = torch.nn.functional.normalize(dvec).unsqueeze(-1)
    with torch.no_grad():
        zf = ac.flow(o["z"].detach(), o["y_mask"], g=g)
        back = ac.flow(zf, o["y_mask"], g=g, reverse=True)
    assert _rel(back * o["y_mask"], o["z"].detach() * o["y_mask"]) < 1e-4
    dec = VitsDecoder(192, 512)
    sd = {}
    for k, (off, numel, shape) in dec.table.items():
        sd[k] = torch.randn(shape, generator=gen) * 0.03
    for k in list(sd):
        if k.endswith("weight_g"):
            v = sd[k[:-1] + "v"]
            sd[k] = v.reshape(v.size(0), -1).norm(dim=1).reshape(sd[k].shape)
    dec.load_state_dict(sd)
    z = torch.randn(B, 192, 32, generator=gen).cuda()
    r = torch.randn(B, 1, 8192, generator=gen).cuda()
    res = []
    for scale in (1.0, 2.0):
        dec.zero_grad()
        zz = z.clone().requires_grad_(True)
        (dec(zz, g) * (r * scale)).sum().backward()
        res.append((zz.grad.clone(), dec.grad.clone()))
    assert _rel(res[1][0], 2 * res[0][0]) < 1e-5 and _rel(res[1][1], 2 * res[0][1]) < 1e-5


@pytest.mark.parametrize("tag", ["te", "pp"])
def test_rel_transformer_dropout_against_oracle_with_the_same_masks(golden_dir, tag):
    """Train-mode RelativePositionTransformer (dropout 0.1 at the reference's four sites per layer: inside the attention kernels, in the conv_o /
    conv_1 GEMM epilogues, on the feed-forward output) against the CPU oracle given THE SAME masks — oracle.xvapitch.HashDrop in the HIP
    path's element order — on the reference modules' weights and inputs of tests/golden/xvapitch_dropout.npz (whose own vectors pin the
    oracle's sites to the reference's train mode, tests/test_xvapitch_cpu.py): output, input gradient and every parameter gradient at 1e-3;
    eval mode returns the dropout-free output; a second seed gives different masks."""
    from oracle import xvapitch as oxv
    from xva_trainer_amd.xvapitch.transformer import RelativePositionTransformer
    g = np.load(os.path.join(golden_dir, "xvapitch_dropout.npz"))
    B, Cc, Co, Fh, H, L, K, W, T = (int(v) for v in g[tag + "_cfg"])
    p = float(g[tag + "_p"][0])
    lens = torch.from_numpy(g["lens"])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    sd = {k[len(tag) + 4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith(tag + "_sd/")}
    tr = RelativePositionTransformer(Cc, Co, Cc, Fh, H, L, kernel_size=K, dropout_p=p, rel_attn_window_size=W, layer_norm_type="2", dropout_site_base=1000)
    tr.load_state_dict(sd)
    seed = 987654321012345
    tr.set_dropout_seed(seed)
    tr.zero_grad()
    x = torch.from_numpy(g[tag + "_x"]).cuda().requires_grad_(True)
    r = torch.from_numpy(g[tag + "_r"])
    y = tr(x, x_mask.cuda())
    (y * r.cuda()).sum().backward()
    torch.cuda.synchronize()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = torch.from_numpy(g[tag + "_x"]).requires_grad_(True)
    yo = oxv.rel_transformer(leaves, xo, x_mask, H, L, K, W, drop=oxv.HashDrop(p, seed, layout="hip"), site0=1000)
    (yo * r).sum().backward()
    assert _rel(y, yo.detach()) < 1e-3 and _rel(x.grad, xo.grad) < 1e-3
    grads = tr.grads()
    worst = sorted(((_rel(grads[n], v.grad), n) for n, v in leaves.items() if v.grad is not None and not n.endswith("conv_k.bias") and float(v.grad.abs().max()) > 0),
                   reverse=True)
    print("transformer dropout (%s): worst gradients %s" % (tag, worst[:3]))
    assert len(worst) >= 18 * L - 8 and worst[0][0] < 1e-3, worst[:4]
    y_eval_ref = oxv.rel_transformer(sd, torch.from_numpy(g[tag + "_x"]), x_mask, H, L, K, W)
    assert _rel(y, y_eval_ref) > 1e-2                                   # dropout really ran
    tr.eval()
    assert _rel(tr(x.detach(), x_mask.cuda()), y_eval_ref) < 1e-3
    tr.train(); tr.set_dropout_seed(seed + 1)
    assert _rel(tr(x.detach(), x_mask.cuda()), yo.detach()) > 1e-2       # another seed, other masks


def test_duration_predictor_dropout_against_oracle_with_the_same_masks(golden_dir):
    """StochasticDurationPredictor in train mode, dropout 0.5 in `convs` / `post_convs` (python/xvapitch/sdp.py:90,227,237), against the oracle
    given the same keyed-hash masks in the HIP path's (B, T, C) element order: the NLL and every gradient."""
    from oracle import xvapitch as oxv
    from xva_trainer_amd.xvapitch.sdp import StochasticDurationPredictor
    g = np.load(os.path.join(golden_dir, "xvapitch_dropout.npz"))
    B, Cin, Hh, Cg, Cl, Ts = (int(v) for v in g["sdp_cfg"])
    lens2 = torch.from_numpy(g["sdp_lens"])
    m_ = (torch.arange(Ts)[None, :] < lens2[:, None]).float().unsqueeze(1)
    sd = {k[7:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sdp_sd/")}
    t = lambda k: torch.from_numpy(g[k])
    dp = StochasticDurationPredictor(Cin, Hh, 3, 0.5, 4, cond_channels=Cg, language_emb_dim=Cl, dropout_site_base=3000)
    assert set(dp.state_dict()) == set(sd)
    dp.load_state_dict({k: v.cuda() for k, v in sd.items()})
    seed = 424242424242
    dp.set_dropout_seed(seed)
    x = t("sdp_x").cuda().requires_grad_(True)
    nll = dp(x, m_.cuda(), t("sdp_dr").cuda(), g=t("sdp_g").cuda(), lang_emb=t("sdp_le").cuda(), noise=t("sdp_noise").cuda())
    nll.sum().backward()
    torch.cuda.synchronize()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xo = t("sdp_x").requires_grad_(True)
    nllo = oxv.sdp_forward(leaves, xo, m_, t("sdp_dr"), t("sdp_noise"), Hh, 3, 4, g=t("sdp_g"), lang_emb=t("sdp_le"), drop=oxv.HashDrop(0.5, seed, layout="hip"), site0=3000)
    nllo.sum().backward()
    assert _rel(nll, nllo.detach()) < 1e-3 and _rel(x.grad, xo.grad) < 2e-3
    worst = sorted(((_rel(v.grad, leaves[k].grad), k) for k, v in dp.p.items() if leaves[k].grad is not None and float(leaves[k].grad.abs().max()) > 0), reverse=True)
    print("sdp dropout: worst gradients", worst[:3])
    assert len(worst) > 100 and worst[0][0] < 5e-3, worst[:4]
    nll_eval = oxv.sdp_forward(sd, t("sdp_x"), m_, t("sdp_dr"), t("sdp_noise"), Hh, 3, 4, g=t("sdp_g"), lang_emb=t("sdp_le"))
    assert _rel(nll, nll_eval) > 1e-3
    dp.eval()
    assert _rel(dp(x.detach(), m_.cuda(), t("sdp_dr").cuda(), g=t("sdp_g").cuda(), lang_emb=t("sdp_le").cuda(), noise=t("sdp_noise").cuda()), nll_eval) < 1e-3


@pytest.mark.parametrize("T", [1, 3, 7])
def test_vits_decoder_short_latents(T):
    """Fewer than 8 latent frames (< 2048 samples) through the waveform decoder — what inference on a short utterance needs — against the oracle:
    waveform and parameter gradients."""
    from oracle import hifigan as ohg
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    sd = ohg.init_vits_decoder_sd(5, 192, 512)
    dec = VitsDecoder(192, 512)
    dec.load_state_dict(sd)
    gen = torch.Generator().manual_seed(6 + T)
    z = torch.randn(2, 192, T, generator=gen); gv = torch.randn(2, 512, 1, generator=gen); r = torch.randn(2, 1, T * 256, generator=gen)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    yo = ohg.vits_decoder(leaves, z, gv)
    (yo * r).sum().backward()
    dec.zero_grad()
    y = dec(z.cuda(), gv.cuda())
    assert tuple(y.shape) == (2, 1, T * 256)
    (y * r.cuda()).sum().backward()
    assert _rel(y, yo.detach()) < 1e-4
    worst = sorted(((_rel(v, leaves[k].grad), k) for k, v in dec.grads().items()), reverse=True)
    assert worst[0][0] < 2e-2 and worst[len(worst) // 2][0] < 1e-3, worst[:4]


@pytest.mark.parametrize("seg", [256, 768, 1792])
def test_vits_discriminator_short_segments(seg):
    """Segments under 2048 samples through VitsDiscriminator (five period discriminators + the scale discriminator): the three losses and the
    waveform gradient of the G pass against the oracle."""
    from oracle import hifigan as ohg
    from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
    sd = ohg.init_vits_disc_sd(3)
    D = VitsDiscriminator()
    D.load_state_dict(sd)
    gen = torch.Generator().manual_seed(seg)
    y = (torch.rand(2, 1, seg, generator=gen) * 1.6 - 0.8); yh = (0.3 * torch.randn(2, 1, seg, generator=gen)).clamp(-1, 1)
    D.zero_grad()
    loss_disc = float(D.d_pass(y.cuda(), yh.cuda()))
    loss_gen, loss_feat, d_wav = D.g_pass(y.cuda(), yh.cuda())
    torch.cuda.synchronize()
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    rs, fr, gs, fg = ohg.vits_disc(leaves, y, yh)
    ld = ohg.discriminator_loss(rs, gs)
    ld.backward()
    yhg = yh.clone().requires_grad_(True)
    rs, fr, gs, fg = ohg.vits_disc(sd, y, yhg)
    lg, lf = ohg.generator_loss(gs), ohg.feature_loss(fr, fg)
    (lg + lf).backward()
    rel = lambda a, b: abs(float(a) - float(b)) / abs(float(b))
    assert rel(loss_disc, ld) < 1e-4 and rel(loss_gen, lg) < 1e-4 and rel(loss_feat, lf) < 1e-4
    assert _rel(d_wav.reshape(yhg.grad.shape), yhg.grad) < 2e-3
    worst = sorted(((_rel(v, leaves[k].grad), k) for k, v in D.grads().items()), reverse=True)
    assert worst[0][0] < 1e-2 and worst[len(worst) // 2][0] < 1e-3, worst[:4]


def test_rq_spline_inverse_against_reference_golden(golden_dir):
    """xva_rq_spline_inv (the duration predictor's sampling direction) against piecewise_rational_quadratic_transform(inverse=True,
    tails="linear") of the reference (util.py:203-350) on values across both tails, the bin edges and the interior: 1e-5; and the forward kernel
    undoes it."""
    import ctypes as C
    from xva_trainer_amd import _lib
    from xva_trainer_amd.xvapitch import sdp
    g = np.load(os.path.join(golden_dir, "xvapitch_infer.npz"))
    y, h = torch.from_numpy(g["spline/y"]).cuda(), torch.from_numpy(g["spline/h"]).cuda().contiguous()
    x = torch.empty_like(y)
    _lib.check(_lib.lib.xva_rq_spline_inv(_lib.ptr(y), _lib.ptr(h), _lib.ptr(x), y.numel(), 10, float(g["spline/wh_scale"]), float(g["spline/bound"]),
                                          _lib.stream_ptr()), "xva_rq_spline_inv")
    assert torch.allclose(x.cpu(), torch.from_numpy(g["spline/x"]), rtol=1e-5, atol=1e-5), float((x.cpu() - torch.from_numpy(g["spline/x"])).abs().max())
    back, _ = sdp.RqSpline.apply(x, h, 10, float(g["spline/wh_scale"]), float(g["spline/bound"]))
    assert torch.allclose(back, y, atol=1e-4)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_infer_against_reference_infer_golden(golden_dir, case):
    """AcousticTrainPath.infer + VitsDecoder against the reference's own `xVAPitch.infer` (model.py:417-599; recorded by
    oracle/gen_golden_xvapitch_infer.py with the duration predictor's N(0, 1) draw): 19 / 7 / 1 symbols at pacing 2.2 / 3.7 / 1 — the ceil
    durations bit-exact (and through durs_only), the latent before the decoder and the waveform at 1e-3."""
    from oracle import hifigan as ohg
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    g = np.load(os.path.join(golden_dir, "xvapitch_infer.npz"))
    c = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    ac = AcousticTrainPath(c["vocab"], c["langs"], latent_size=c["latent"], embedded_language_dim=c["lang_dim"], d_vector_dim=c["dvec"],
                           hidden_channels_ffn=c["ffn"], num_heads=c["heads"], text_layers=c["te_layers"], posterior_layers=c["pe_layers"],
                           flow_layers=c["flow_layers"], num_flows=c["num_flows"], spec_bins=c["spec_bins"], pitch=True, dropout_p=0.1, sdp_dropout_p=0.5)
    ac.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
    dec = VitsDecoder(c["latent"], c["dvec"])
    dec.load_state_dict(ohg.init_vits_decoder_sd(int(g["dec_seed"]), c["latent"], c["dvec"]))
    pre = "c%d/" % case
    t = lambda k: torch.from_numpy(g[pre + k]).cuda()
    kw = dict(pacing=float(g[pre + "pacing"]), noise=t("noise"))
    w = ac.infer(t("tokens"), t("dvec"), t("lid"), dec, durs_only=True, **kw)
    assert torch.equal(w.cpu(), torch.from_numpy(g[pre + "w_ceil"]))
    wav = ac.infer(t("tokens"), t("dvec"), t("lid"), dec, **kw)
    ref = torch.from_numpy(g[pre + "wav"])
    assert wav.shape == ref.shape
    assert _rel(ac.last_infer["z"], torch.from_numpy(g[pre + "z"])) < 1e-3
    assert _rel(wav, ref) < 1e-3, _rel(wav, ref)
    assert ac.training                                                              # infer leaves the training flag as it found it


def test_c5_benchmarked_schedule_at_full_size_equals_one_stream_and_the_four_clip_batch():
    """the schedule bench.py times — B = 16 x 100 symbols x 400 frames at the reference's model size, throughput mode, five streams,
    the discriminator pass inside the generator pass (eager_disc), late_join — checked at ITS size:
      (i) the same batch on ONE stream (XVA_C5_STREAMS=0, XVA_C5_BRANCH_STREAM=0, XVA_C5_LATE_JOIN=0, engine lanes off, the discriminator pass after the
          backward pass, as python/xvapitch/model.py:272-384 orders it) gives the same losses and the same gradients: a missing event wait or a tensor
          freed under a side stream shows up here and nowhere at B = 4;
      (ii) the batch is 4 copies each of 4 clips with the random draws replicated (dropout off): every loss of the reference is a masked mean (numerator and normaliser
          both scale by 4), so every loss and every parameter gradient must equal the 4-clip batch's — tile shapes, split-K factors and launch grids
          differ between the two, so the bound is bf16 summation-order noise (the bound tests/test_fullsize_gpu.py uses for HiFi-GAN)."""
    from xva_trainer_amd import _lib
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
    from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
    from xva_trainer_amd.xvapitch.train_step import XVAPitchStep
    dev = torch.device("cuda")
    VOCAB, LANGS, SEG, Tt, Ty = 256, 31, 32, 100, 400
    gen = torch.Generator().manual_seed(5)
    ac = AcousticTrainPath(VOCAB, LANGS, pitch=True, compute="bf16", device=dev, dropout_p=0.1, sdp_dropout_p=0.5)
    dec, D = VitsDecoder(192, 512, compute="bf16", device=dev), VitsDiscriminator(compute="bf16", device=dev)
    for eng in (dec, D):
        sd = {k: torch.randn(shape, generator=gen) * 0.02 for k, (off, numel, shape) in eng.table.items()}
        for k in list(sd):
            if k.endswith("weight_g"):
                v = sd[k[:-1] + "v"]
                sd[k] = v.reshape(v.size(0), -1).norm(dim=1).reshape(sd[k].shape)
        eng.load_state_dict(sd)
    step = XVAPitchStep(GeneratorPass(ac, dec, SEG), D)
    x4 = torch.tensor([100, 77, 60, 51]); y4 = torch.tensor([400, 333, 251, 140])
    tok4 = torch.randint(1, VOCAB, (4, Tt), generator=gen) * (torch.arange(Tt)[None, :] < x4[:, None])
    wl4 = (y4 - 1) * 256 + torch.randint(0, 256, (4,), generator=gen)
    wav4 = torch.rand(4, (Ty - 1) * 256 + 255, generator=gen) * 0.1 - 0.05
    wav4 = wav4 * (torch.arange(wav4.size(1))[None, :] < wl4[:, None])
    dv4, li4 = torch.randn(4, 512, generator=gen), torch.randint(0, LANGS, (4,), generator=gen)
    pit4 = (torch.rand(4, 1, Ty, generator=gen) * 3 - 1.2).clamp_min(0) * (torch.arange(Ty)[None, None, :] < y4[:, None, None])
    eps4, noi4 = torch.randn(4, 192, Ty, generator=gen), torch.randn(4, 2, Tt, generator=gen)
    ids4 = (torch.rand(4, generator=gen) * (y4 - SEG + 1)).long()
    rep = torch.arange(16) % 4                               # clip 0 first (the longest): the padded sizes are the same in both batches
    LOSSES = ("loss", "loss_kl", "loss_duration", "loss_pitch", "loss_mel", "loss_gen", "loss_feat")

    def run(idx, eager, train=True):
        c = lambda t: t[idx].contiguous().to(dev)
        ac.train(train)
        ac.set_dropout_seed(20240905)                        # every run draws the masks of the same "iteration" (the call count restarts)
        step.gen.zero_grad(); D.zero_grad()
        y, yl, wav = step.gen.batch_from_wav(c(wav4), c(wl4))
        o = step.generator_pass(c(tok4), c(x4), y, yl, wav, c(dv4), c(li4), pitch_padded=c(pit4), eps=c(eps4), noise=c(noi4), slice_ids=c(ids4), eager_disc=eager)
        o["loss"].backward()
        ld = step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
        torch.cuda.synchronize()
        from xva_trainer_amd.xvapitch import ops
        ops.raise_deferred()
        res = {k: float(o[k]) for k in LOSSES}
        res["loss_disc"] = float(ld)
        g = {"ac/" + k: v.detach().float().clone() for k, v in ac.grads().items() if v is not None}
        g["dec"] = dec.grad.detach().float().clone(); g["disc"] = D.grad.detach().float().clone()
        return res, g, o["model_outputs"].detach().float().clone()

    def compare(a, b, ltol, gtol, what):
        (la, ga, oa), (lb, gb, ob) = a, b
        print("C5 full size, %s: relative loss differences %s" % (what, {k: "%.1e" % (abs(la[k] - lb[k]) / abs(lb[k])) for k in la}))
        for k in la:
            assert abs(la[k] - lb[k]) <= ltol * abs(lb[k]), (what, k, la[k], lb[k])
        assert set(ga) == set(gb) and len(ga) > 400
        # (the keys' bias shifts every logit of a softmax row alike: its gradient is zero up to rounding noise — as the reference-golden cases, skip it)
        worst = sorted(((float((ga[k].double() - gb[k].double()).norm() / gb[k].double().norm().clamp_min(1e-30)), k) for k in ga
                        if float(gb[k].abs().max()) > 0 and not k.endswith("conv_k.bias")), reverse=True)
        print("C5 full size, %s: worst gradient tensors (relative L2) %s of %d" % (what, worst[:3], len(worst)))
        va, vb = torch.cat([ga[k].double().flatten() for k in sorted(ga)]), torch.cat([gb[k].double().flatten() for k in sorted(gb)])
        whole = float((va - vb).norm() / vb.norm())
        print("   all %d gradient elements as one vector: relative L2 %.2e" % (va.numel(), whole))
        if gtol is not None:                                 # the same kernels on the same shapes: element-level agreement
            assert worst[0][0] < gtol and whole < gtol, (what, worst[:5], whole)
        else:
            # different launch shapes: bf16 summation-order noise.  It is direction-preserving (cosine, norm) per tensor; the relative L2 of a tensor whose
            # gradient is a sum of near-cancelling terms (the flow's WaveNet weight_g / weight_v: 3 - 4 %) is not a useful bound
            for r, k in worst:
                xa, xb = ga[k].double().flatten(), gb[k].double().flatten()
                if xa.numel() >= 64:
                    cos, ratio = float(xa @ xb / (xa.norm() * xb.norm())), float(xa.norm() / xb.norm())
                    assert cos > 0.999 and abs(ratio - 1) < 5e-3, (what, k, r, cos, ratio)
            assert whole < 1e-2, (what, whole)
        return oa, ob

    full = run(rep, eager=True)                              # the benchmarked schedule
    assert full[2].shape[0] == 16 and all(np.isfinite(v) for v in full[0].values())
    # (i) one stream, reference order of the two passes
    env = {"XVA_C5_STREAMS": "0", "XVA_C5_BRANCH_STREAM": "0", "XVA_C5_LATE_JOIN": "0"}
    old = {k: os.environ.get(k) for k in env}
    old_lanes = _lib.lib.xva_hg_set_streams(1)
    os.environ.update(env)
    try:
        serial = run(rep, eager=False)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _lib.lib.xva_hg_set_streams(old_lanes)
    oa, ob = compare(full, serial, 5e-4, 2e-4, "five streams / eager_disc / late_join vs one stream")
    assert _rel(oa, ob) < 1e-6
    # (ii) 4 copies of 4 clips = the 4-clip batch (dropout off: a mask is a function of the element's position in the batch)
    full_eval = run(rep, eager=True, train=False)
    four = run(torch.arange(4), eager=True, train=False)
    ac.train(True)
    oa, ob = compare(full_eval, four, 2e-3, None, "16 = 4 x 4 clips vs the 4-clip batch")
    for b in range(4):
        assert _rel(oa[b], ob[b]) < 1e-2 and _rel(oa[b + 12], ob[b]) < 1e-2
