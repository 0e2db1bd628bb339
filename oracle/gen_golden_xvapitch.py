"""Golden vectors for the xVAPitch-only blocks, recorded by RUNNING THE REFERENCE's own modules (python/xvapitch/{wavenet,model,util,
losses}.py imported from /root/reference with the stubs of oracle/ref_import.py).  Build container only:

    python oracle/gen_golden_xvapitch.py

Writes tests/golden/xvapitch_blocks.npz: WN (with conditioning) and ResidualCouplingBlock (mean_only, forward + reverse) state_dicts, inputs,
outputs and every gradient of a fixed scalar loss; maximum_path inputs / path; segment and kl_loss inputs / outputs / gradients.  Asserts
oracle/xvapitch.py equal to the reference first.  Data only — no reference source is copied."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import, xvapitch as oxv  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def import_ref():
    ref_import._install_stubs()
    for name in ("unidecode", "inflect", "soundfile", "pysbd", "gruut", "TTS"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    from python.xvapitch import wavenet, util
    try:
        from python.xvapitch import model
        RCB = model.ResidualCouplingBlock
    except Exception as e:                                  # model.py pulls the whole text front end: fall back to exec'ing only the class it defines
        print("note: python.xvapitch.model does not import here (%s): loading ResidualCouplingBlock from its source lines" % type(e).__name__)
        src = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "model.py")).read()
        a = src.index("class ResidualCouplingBlock(nn.Module):")
        b = src.index("class DiscriminatorS(torch.nn.Module):")
        ns = {"torch": torch, "nn": torch.nn, "WN": wavenet.WN}
        exec(compile(src[a:b], "model.py:ResidualCouplingBlock", "exec"), ns)          # executed in memory only; nothing of it is stored
        RCB = ns["ResidualCouplingBlock"]
    src = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "model.py")).read()
    a = src.index("class PosteriorEncoder(nn.Module):")
    b = src.index("class ResidualCouplingBlock(nn.Module):")
    ns3 = {"torch": torch, "nn": torch.nn, "WN": wavenet.WN, "sequence_mask": util.sequence_mask}
    exec(compile(src[a:b], "model.py:PosteriorEncoder", "exec"), ns3)
    global PosteriorEncoderRef
    PosteriorEncoderRef = ns3["PosteriorEncoder"]
    # losses.py imports the whole model (text front end, espeak, ...) at module level: take VitsGeneratorLoss.kl_loss — a pure staticmethod —
    # from its source lines, again in memory only
    import textwrap
    lsrc = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "losses.py")).read()
    a = lsrc.index("    def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):")
    b = lsrc.index("    @staticmethod", a)
    ns2 = {"torch": torch}
    exec(compile(textwrap.dedent(lsrc[a:b]), "losses.py:kl_loss", "exec"), ns2)
    return wavenet, util, ns2["kl_loss"], RCB


def grads_of(module, loss):
    for p in module.parameters():
        p.grad = None
    loss.backward()
    return {n: p.grad.detach().clone() for n, p in module.named_parameters()}


def main():
    wavenet, util, ref_kl_loss, RCB = import_ref()
    torch.manual_seed(11)
    out = {}
    B, H, T, CIN, L, K = 2, 32, 50, 16, 3, 5
    lens = torch.tensor([50, 37])
    x_mask = (torch.arange(T)[None, :] < lens[:, None]).float().unsqueeze(1)
    # ---- WN with conditioning
    wn = wavenet.WN(H, H, K, 1, L, c_in_channels=CIN)
    for p in wn.parameters():
        p.data += 0.05 * torch.randn_like(p)
    x = torch.randn(B, H, T, requires_grad=True)
    g = torch.randn(B, CIN, 1, requires_grad=True)
    r = torch.randn(B, H, T)
    y = wn(x * 1.0, x_mask, g=g)
    sd = {k: v.detach().clone() for k, v in wn.state_dict().items()}
    yo = oxv.wn(sd, x, x_mask, g, hidden=H, kernel_size=K, dilation_rate=1, num_layers=L)
    assert torch.allclose(y, yo, rtol=1e-5, atol=1e-6), (y - yo).abs().max()
    pg = grads_of(wn, (y * r).sum())
    out.update({"wn_cfg": np.array([B, H, T, CIN, L, K]), "wn_lens": lens.numpy(), "wn_x": x.detach().numpy(), "wn_g": g.detach().numpy(), "wn_r": r.numpy(),
                "wn_y": y.detach().numpy(), "wn_dx": x.grad.numpy(), "wn_dg": g.grad.numpy()})
    for k, v in sd.items():
        out["wn_sd/" + k] = v.numpy()
    for k, v in pg.items():
        out["wn_grad/" + k] = v.numpy()
    # ---- ResidualCouplingBlock (mean_only), forward and reverse
    CH = 32
    blk = RCB(CH, H, K, 1, 2, cond_channels=0, mean_only=True)
    for p in blk.parameters():
        p.data += 0.05 * torch.randn_like(p)
    blk.post.weight.data = 0.1 * torch.randn_like(blk.post.weight)          # the reference zero-initialises `post`: give it values so the gradient is exercised
    xc = torch.randn(B, CH, T, requires_grad=True)
    rc = torch.randn(B, CH, T)
    yc, logdet = blk(xc * 1.0, x_mask)
    sdc = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    yco = oxv.coupling(sdc, xc, x_mask, hidden=H, kernel_size=K, dilation_rate=1, num_layers=2)
    assert torch.allclose(yc, yco, rtol=1e-5, atol=1e-6) and float(logdet.abs().max()) == 0.0
    pgc = grads_of(blk, (yc * rc).sum())
    with torch.no_grad():
        yrev = blk(xc.detach(), x_mask, reverse=True)
        assert torch.allclose(yrev, oxv.coupling(sdc, xc.detach(), x_mask, reverse=True, hidden=H, kernel_size=K, dilation_rate=1, num_layers=2), rtol=1e-5, atol=1e-6)
    out.update({"cp_cfg": np.array([B, CH, H, T, 2, K]), "cp_x": xc.detach().numpy(), "cp_r": rc.numpy(), "cp_y": yc.detach().numpy(), "cp_dx": xc.grad.numpy(),
                "cp_yrev": yrev.numpy()})
    for k, v in sdc.items():
        out["cp_sd/" + k] = v.numpy()
    for k, v in pgc.items():
        out["cp_grad/" + k] = v.numpy()
    # ---- PosteriorEncoder (linear-spectrogram bins -> z, m, logs): conditioned WN inside, the N(0, 1) draw pinned by the seed
    CSP, CO = 40, 12
    pe = PosteriorEncoderRef(CSP, CO, H, K, 1, 3, cond_channels=CIN)
    for p in pe.parameters():
        p.data += 0.05 * torch.randn_like(p)
    xp = torch.rand(B, CSP, T, requires_grad=True)
    gp = torch.randn(B, CIN, 1, requires_grad=True)
    torch.manual_seed(77)
    z, mean, logs, pm = pe(xp * 1.0, lens, g=gp)
    torch.manual_seed(77)
    eps = torch.randn(B, CO, T)
    sdp = {k: v.detach().clone() for k, v in pe.state_dict().items()}
    zo, mo, lo2, _ = oxv.posterior_encoder(sdp, xp, lens, gp, eps, CO, hidden=H, kernel_size=K, dilation_rate=1, num_layers=3)
    assert torch.allclose(z, zo, rtol=1e-5, atol=1e-6) and torch.allclose(mean, mo, rtol=1e-5, atol=1e-6) and torch.allclose(logs, lo2, rtol=1e-5, atol=1e-6)
    rz, rm, rl = torch.randn(B, CO, T), torch.randn(B, CO, T), torch.randn(B, CO, T)
    pgp = grads_of(pe, (z * rz).sum() + (mean * rm).sum() + 0.5 * (logs * rl).sum())
    out.update({"pe_cfg": np.array([B, CSP, CO, H, T, 3, K, CIN]), "pe_x": xp.detach().numpy(), "pe_g": gp.detach().numpy(), "pe_eps": eps.numpy(),
                "pe_z": z.detach().numpy(), "pe_mean": mean.detach().numpy(), "pe_logs": logs.detach().numpy(), "pe_rz": rz.numpy(), "pe_rm": rm.numpy(),
                "pe_rl": rl.numpy(), "pe_dx": xp.grad.numpy(), "pe_dg": gp.grad.numpy()})
    for k, v in sdp.items():
        out["pe_sd/" + k] = v.numpy()
    for k, v in pgp.items():
        out["pe_grad/" + k] = v.numpy()
    # ---- maximum_path
    b, tx, ty = 4, 17, 40
    xl, yl = torch.tensor([17, 9, 12, 1]), torch.tensor([40, 31, 12, 20])
    mask = ((torch.arange(tx)[None, :, None] < xl[:, None, None]) & (torch.arange(ty)[None, None, :] < yl[:, None, None])).float()
    value = torch.randn(b, tx, ty)
    value[2] = torch.round(value[2] * 2) / 2                                 # ties: the >= rule decides
    path = util.maximum_path(value, mask)
    assert np.array_equal(oxv.maximum_path(value.numpy(), mask.numpy()), path.numpy())
    out.update({"mp_value": value.numpy(), "mp_mask": mask.numpy(), "mp_path": path.numpy()})
    # ---- segment
    xs = torch.randn(3, 6, 30)
    idx = torch.tensor([0, 11, 26])
    seg = util.segment(xs, idx, 4)
    assert torch.equal(seg, oxv.segment(xs, idx, 4))
    out.update({"sg_x": xs.numpy(), "sg_idx": idx.numpy(), "sg_out": seg.numpy()})
    # ---- kl_loss
    t4 = [torch.randn(B, 12, T, requires_grad=True) for _ in range(4)]
    l, kl_sw = ref_kl_loss(t4[0], t4[1], t4[2], t4[3], x_mask)
    lo, _ = oxv.kl_loss(t4[0], t4[1], t4[2], t4[3], x_mask)
    assert abs(l.item() - lo.item()) < 1e-6
    (l * 1.7).backward()
    out.update({"kl_in": np.stack([t.detach().numpy() for t in t4]), "kl_mask": x_mask.numpy(), "kl_loss": np.float32(l.item()), "kl_sw": kl_sw.detach().numpy(),
                "kl_grads": np.stack([t.grad.numpy() for t in t4])})
    np.savez_compressed(os.path.join(OUT, "xvapitch_blocks.npz"), **out)
    print("xvapitch_blocks.npz:", len(out), "arrays;", "WN y", out["wn_y"].shape, "coupling y", out["cp_y"].shape, "kl", float(l))


if __name__ == "__main__":
    main()
