"""Golden vectors of xVAPitch's INFERENCE direction: the reference's own `xVAPitch.infer` (python/xvapitch/model.py:417-599) and
`expand_pitch_energy` (:935-958), compiled from their source lines in memory and bound to the holder of gen_golden_xvapitch_genpass.py — the
reference's TextEncoder, StochasticDurationPredictor (run with reverse=True, sdp.py:311-321: the inverse rational-quadratic splines of
util.py:322-350), pitch predictor, ResidualCouplingBlocks (reverse) and HifiganGenerator — on the switches xVAPitchModel sets
(xva_train.py:1424-1428: --pitch 1, pe_scaling 0.1, energy / ow_flow / expanded_flow 0).

    python oracle/gen_golden_xvapitch_infer.py       -> tests/golden/xvapitch_infer.npz

Records the symbols, speaker vector, language id, the duration predictor's N(0, 1) draw, and the reference's outputs: ceil durations, the waveform,
plus — for three spline-only cases — inputs / raw parameters / outputs of piecewise_rational_quadratic_transform(inverse=True, tails="linear").
Asserts the CPU restatement (oracle.xvapitch.infer / rq_spline_inverse) equal to the reference first.  Data only."""
import importlib
import os
import sys
import textwrap
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_xvapitch_genpass as gg, hifigan as ohg, ref_import, xvapitch as oxv  # noqa: E402


def main():
    bd = gg.build()
    ns, m, c, dec_sd = bd["ns"], bd["m"], bd["c"], bd["dec_sd"]
    util = importlib.import_module("python.xvapitch.util")
    src = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "model.py")).read()

    def cut(a, b):
        return src[src.index(a):src.index(b)]
    ns["generate_path"] = util.generate_path
    exec(compile(textwrap.dedent(cut("    def infer (self,", "    def voice_conversion(self,")), "model.py:infer", "exec"), ns)
    exec(compile(textwrap.dedent(cut("    def expand_pitch_energy (self,", "    # def expand_lang (self,")), "model.py:expand_pitch_energy", "exec"), ns)
    m.infer = types.MethodType(ns["infer"], m)
    m.expand_pitch_energy = types.MethodType(ns["expand_pitch_energy"], m)
    m.args.pitch, m.args.energy, m.args.energy_sp = 1, 0, 0
    m.length_scale, m.inference_noise_scale, m.inference_noise_scale_dp, m.max_inference_len = 1.0, 0.333, 0.333, None      # model.py:68-75
    m.eval()
    res = {"cfg_keys": np.array(sorted(c)), "cfg_vals": np.array([c[k] for k in sorted(c)]), "dec_seed": np.int64(gg.DEC_SEED)}
    sd = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith("waveform_decoder.")}
    for k, v in sd.items():
        res["sd/" + k] = v.numpy()
    dl = {k: v.clone() for k, v in dec_sd.items()}
    for case, (Tt, seed, pacing) in enumerate(((19, 5, 2.2), (7, 6, 3.7), (1, 7, 1.0))):
        g = torch.Generator().manual_seed(seed)
        tokens = torch.randint(1, c["vocab"], (1, Tt), generator=g)
        dvec = torch.randn(c["dvec"], generator=g)
        lid = torch.tensor(int(torch.randint(0, c["langs"], (1,), generator=g)))
        torch.manual_seed(100 + seed)
        with torch.no_grad():
            wav = m.infer(tokens, lid, dvec, pacing=pacing)
            torch.manual_seed(100 + seed)
            w_ceil = m.infer(tokens, lid, dvec, durs_only=True, pacing=pacing)
        torch.manual_seed(100 + seed)
        noise = torch.randn(1, 2, Tt)                                    # sdp.py:313, the first draw of infer
        with torch.no_grad():
            o = oxv.infer(sd, tokens, dvec, lid, noise, c, lambda z, gg_: ohg.vits_decoder(dl, z, gg_), pacing=pacing)
        assert torch.equal(o["w_ceil"], w_ceil), (o["w_ceil"], w_ceil)
        assert wav.shape == o["wav"].shape and torch.allclose(o["wav"], wav, rtol=1e-4, atol=2e-5), float((o["wav"] - wav).abs().max())
        print("case %d: Tt %d -> %d frames, |wav| max %.4f, oracle - reference %.2e; logw min margin to an integer boundary %.3e" % (
            case, Tt, int(w_ceil.sum()), float(wav.abs().max()), float((o["wav"] - wav).abs().max()),
            float(((torch.exp(o["logw"]) * pacing) - torch.round(torch.exp(o["logw"]) * pacing)).abs().min())))
        pre = "c%d/" % case
        res.update({pre + "tokens": tokens.numpy(), pre + "dvec": dvec.numpy(), pre + "lid": lid.numpy(), pre + "noise": noise.numpy(), pre + "pacing": np.float32(pacing),
                    pre + "w_ceil": w_ceil.numpy(), pre + "wav": wav.numpy(), pre + "logw": o["logw"].numpy(), pre + "z": o["z"].numpy()})
    # the inverse spline alone, through the reference function (values across the tails, the bin edges and the interior)
    torch.manual_seed(9)
    K = 10
    y = torch.cat([torch.linspace(-6.5, 6.5, 201), torch.tensor([-5.0, 5.0, 0.0])]).reshape(1, 1, -1)
    n = y.numel()
    h = torch.randn(1, 1, n, 3 * K - 1) * 1.5
    with torch.no_grad():
        x_ref, ld_ref = util.piecewise_rational_quadratic_transform(y, h[..., :K] / 4.0, h[..., K:2 * K] / 4.0, h[..., 2 * K:], inverse=True, tails="linear",
                                                                    tail_bound=5.0)
        x_or = oxv.rq_spline_inverse(y, h[..., :K] / 4.0, h[..., K:2 * K] / 4.0, h[..., 2 * K:], 5.0)
        back, _ = util.piecewise_rational_quadratic_transform(x_ref, h[..., :K] / 4.0, h[..., K:2 * K] / 4.0, h[..., 2 * K:], inverse=False, tails="linear",
                                                              tail_bound=5.0)
    assert torch.allclose(x_or, x_ref, rtol=1e-5, atol=1e-5), float((x_or - x_ref).abs().max())
    assert torch.allclose(back, y, atol=1e-4)
    res.update({"spline/y": y.reshape(-1).numpy(), "spline/h": h.reshape(n, 3 * K - 1).numpy(), "spline/x": x_ref.reshape(-1).numpy(),
                "spline/wh_scale": np.float32(0.25), "spline/bound": np.float32(5.0)})
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_infer.npz")
    np.savez_compressed(path, **res)
    print("xvapitch_infer.npz: %.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
