"""Golden vectors of the xVAPitch acoustic training path — text encoder, posterior encoder, flow, monotonic alignment search, stochastic
duration predictor, prior expansion, KL + duration losses — recorded by RUNNING THE REFERENCE's own `xVAPitch.train_step`
(python/xvapitch/model.py:681-870) in the build container:

    python oracle/gen_golden_xvapitch_acoustic.py

model.py does not import here as a module (its header pulls the whole text front end), so its classes and the train_step /
_set_cond_input methods are compiled from their source lines IN MEMORY (nothing of them is stored) and bound to a small holder object
that carries exactly the attributes train_step reads: the reference's own TextEncoder, PosteriorEncoder, ResidualCouplingBlocks and
StochasticDurationPredictor at a reduced width, nn.Embedding for the language, the default switches of xva_train.py:1098-1120
(--pitch / --energy / --flc / --ow_flow / --mltts_rc 0; detach_dp_input True, model.py:52; dropout 0), and a waveform decoder stand-in
that returns zeros (the decoder / discriminator branch is the HiFi-GAN path and takes no part in the losses recorded here).

Two runs of train_step on the same weights and batch: the argparse defaults (--pitch 0), and --pitch 1 with pe_scaling 0.1 — what the shipped
trainer sets (xva_train.py:1421-1425): pitch_emb subtracted from z_p, average_pitch targets, the pitch predictor and the pitch loss term
(losses.py:224-241, restated here on the reference's outputs because VitsGeneratorLoss needs the discriminator branch).

The script asserts oracle/xvapitch.py:acoustic_losses equal to each run (outputs, losses and every parameter gradient), then writes
tests/golden/xvapitch_acoustic.npz: state_dict, batch, the two N(0, 1) draws, outputs, losses; gradients in full for the first run, for the
second run the pitch tensors in full and every tensor by norm + 256 evenly spaced samples (oracle/golden_util.py).  Data only."""
import importlib
import math
import os
import sys
import textwrap
import types

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import, xvapitch as oxv  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
CFG = {"latent": 32, "lang_dim": 4, "dvec": 16, "heads": 2, "te_layers": 2, "ffn": 48, "pe_layers": 3, "flow_layers": 2, "num_flows": 4, "spec_bins": 41,
       "vocab": 30, "langs": 5}


def load_reference():
    ref_import._install_stubs()
    if ref_import.REF_ROOT not in sys.path:
        sys.path.insert(0, ref_import.REF_ROOT)
    glow = importlib.import_module("python.xvapitch.glow_tts")
    sdp = importlib.import_module("python.xvapitch.sdp")
    wavenet = importlib.import_module("python.xvapitch.wavenet")
    util = importlib.import_module("python.xvapitch.util")
    src = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "model.py")).read()
    ns = {"torch": torch, "nn": torch.nn, "F": F, "math": math, "WN": wavenet.WN, "RelativePositionTransformer": glow.RelativePositionTransformer,
          "sequence_mask": util.sequence_mask, "maximum_path": util.maximum_path, "rand_segments": util.rand_segments, "segment": util.segment,
          "mask_from_lens": lambda lens, max_len=None: torch.arange(max_len)[None, :] < lens[:, None]}

    def cut(a, b):
        return src[src.index(a):src.index(b)]
    exec(compile(cut("class TextEncoder(nn.Module):", "class ResidualCouplingBlocks(nn.Module):"), "model.py:TextEncoder+pitch encoder", "exec"), ns)
    exec(compile(cut("def average_pitch(pitch, durs):", "class GradientReversalFunction(torch.autograd.Function):"), "model.py:average_pitch", "exec"), ns)
    exec(compile(cut("class ResidualCouplingBlocks(nn.Module):", "class DiscriminatorS(torch.nn.Module):"), "model.py:flow+posterior", "exec"), ns)
    methods = textwrap.dedent(cut("    def train_step(self,", "    # Opposite of average_pitch"))
    exec(compile(methods, "model.py:train_step", "exec"), ns)
    lsrc = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "losses.py")).read()
    a = lsrc.index("    def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):")
    exec(compile(textwrap.dedent(lsrc[a:lsrc.index("    @staticmethod", a)]), "losses.py:kl_loss", "exec"), ns)
    ns["StochasticDurationPredictor"] = sdp.StochasticDurationPredictor
    return ns


class Holder(torch.nn.Module):
    """The attributes xVAPitch.train_step reads, built from the reference's classes with the constructor arguments of model.py:79-135 at CFG's width."""

    def __init__(self, ns):
        super().__init__()
        c = CFG
        Cc = c["latent"]
        self.args = types.SimpleNamespace(pitch=0, energy=0, flc=0, ow_flow=0, mltts_rc=0, expanded_flow=0, expanded_flow_dim=0, lang_w=1, detach_dp_input=True,
                                          d_vector_dim=c["dvec"], pe_scaling=0.1)
        self.spec_segment_size = 4
        self.emb_l = torch.nn.Embedding(c["langs"], c["lang_dim"])
        self.text_encoder = ns["TextEncoder"](c["vocab"], Cc, Cc, c["ffn"], c["heads"], c["te_layers"], 3, 0.0, language_emb_dim=c["lang_dim"])
        self.posterior_encoder = ns["PosteriorEncoder"](c["spec_bins"], Cc, Cc, kernel_size=5, dilation_rate=1, num_layers=c["pe_layers"], cond_channels=c["dvec"])
        self.flow = ns["ResidualCouplingBlocks"](Cc, Cc, kernel_size=5, dilation_rate=1, num_layers=c["flow_layers"], cond_channels=c["dvec"], args=self.args)
        self.duration_predictor = ns["StochasticDurationPredictor"](Cc, Cc, 3, 0.0, 4, cond_channels=c["dvec"], language_emb_dim=c["lang_dim"])
        self.waveform_decoder = lambda z_slice, g=None: torch.zeros(z_slice.size(0), 1, z_slice.size(2) * 256)
        # --pitch 1 (what the shipped trainer sets, xva_train.py:1421-1425): model.py:153-176 at CFG's width
        self.pitch_predictor = ns["RelativePositioningPitchEnergyEncoder"](out_channels=1, hidden_channels=Cc + c["lang_dim"], hidden_channels_ffn=c["ffn"],
                                                                          num_heads=c["heads"], num_layers=3, kernel_size=3, dropout_p=0.0,
                                                                          conditioning_emb_dim=c["dvec"])
        self.pitch_emb = torch.nn.Conv1d(1, Cc, kernel_size=3, padding=1)


def main():
    ns = load_reference()
    torch.manual_seed(41)
    m = Holder(ns)
    m.train_step = types.MethodType(ns["train_step"], m)
    m._set_cond_input = types.MethodType(ns["_set_cond_input"], m)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if "gamma" in n or "beta" in n or n.endswith("log_scale") or n.endswith("translation"):
                p += 0.1 * torch.randn_like(p)
            if n.endswith("post.weight") or n.endswith("post.bias") or (".proj." in n and "duration_predictor.flows" in n) or (".proj." in n and "post_flows" in n):
                p += 0.05 * torch.randn_like(p)                         # the reference zero-initialises these: give them values so their gradients are exercised
    c = CFG
    B, Tt, Ty = 3, 19, 50
    x_lens, y_lens = torch.tensor([19, 11, 7]), torch.tensor([50, 37, 21])
    tokens = torch.randint(1, c["vocab"], (B, Tt))
    tokens = tokens * (torch.arange(Tt)[None, :] < x_lens[:, None])
    y = torch.rand(B, c["spec_bins"], Ty) * (torch.arange(Ty)[None, None, :] < y_lens[:, None, None])
    dvec = torch.randn(B, c["dvec"])
    lids = torch.tensor([0, 3, 1])
    zeros_t, zeros_y = torch.zeros(B, 1, Tt), torch.zeros(B, 1, Ty * 256)
    SEED = 97
    pitch = (torch.rand(B, 1, Ty) * 3 - 1.2).clamp_min(0) * (torch.arange(Ty)[None, None, :] < y_lens[:, None, None])      # zeros = unvoiced frames
    y_mask = (torch.arange(Ty)[None, :] < y_lens[:, None]).float()
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    res = {"cfg_keys": np.array(sorted(c)), "cfg_vals": np.array([c[k] for k in sorted(c)]), "tokens": tokens.numpy(), "x_lens": x_lens.numpy(),
           "y": y.numpy(), "y_lens": y_lens.numpy(), "dvec": dvec.numpy(), "lids": lids.numpy(), "pitch": pitch.numpy()}
    for k, v in sd.items():
        res["sd/" + k] = v.numpy()
    from oracle import golden_util
    for tag, use_pitch in (("", 0), ("p_", 1)):
        m.args.pitch = use_pitch
        torch.manual_seed(SEED)
        out = m.train_step(tokens, x_lens, y, y_lens, pitch if use_pitch else zeros_t, zeros_t, zeros_y, aux_input={"d_vectors": dvec, "language_ids": lids})
        torch.manual_seed(SEED)
        eps = torch.randn(B, c["latent"], Ty)                            # PosteriorEncoder's randn_like (model.py:1472), first draw of the step
        noise = torch.randn(B, 2, Tt)                                    # sdp.py:281, second draw
        loss_kl, _ = ns["kl_loss"](out["z_p"], out["logs_q"], out["m_p"], out["logs_p"], y_mask.unsqueeze(1))      # losses.py:213
        loss_dur = torch.sum(out["loss_duration"].float())                                                      # losses.py:220
        loss = loss_kl + loss_dur
        loss_pitch = torch.zeros(())
        if use_pitch:                                                    # losses.py:224-241, its lines restated on the reference's outputs
            lp = F.mse_loss(out["pitch_tgt"], out["pitch_pred"], reduction="none") * out["mask"].unsqueeze(1)
            loss_pitch = lp.sum() / out["mask"].sum() / out["pitch_pred"].shape[0] * 0.1
            loss = loss + loss_pitch
        m.zero_grad()
        loss.backward()
        grads = {n: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for n, p in m.named_parameters()}
        # ---- the restatement must agree with the reference run before anything is recorded
        leaves = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in sd.items()}
        o = oxv.acoustic_losses(leaves, tokens, x_lens, y, y_lens, dvec, lids, eps, noise, c, pitch_padded=pitch if use_pitch else None, pe_scaling=m.args.pe_scaling)
        keys = ("z", "m_q", "logs_q", "z_p", "m_p", "logs_p") + (("pitch_tgt", "pitch_pred") if use_pitch else ())
        for k in keys:
            assert torch.allclose(o[k], out[k], rtol=1e-4, atol=1e-5), (tag, k, float((o[k] - out[k]).abs().max()))
        for a, b in ((o["loss_kl"], loss_kl), (o["loss_duration"], loss_dur)) + (((o["loss_pitch"], loss_pitch),) if use_pitch else ()):
            assert abs(float(a.detach()) - float(b)) < 1e-4 * max(1.0, abs(float(b))), (tag, float(a.detach()), float(b))
        o["loss"].backward()
        worst = 0.0
        for n, gr in grads.items():
            go = leaves[n].grad if leaves[n].grad is not None else torch.zeros_like(gr)
            if float(gr.norm()) < 1e-5 * gr.numel() ** 0.5:               # mathematically zero (conv_k.bias: softmax is shift invariant; unused branch): rounding noise only
                assert float(go.norm()) < 1e-4, (n, float(go.norm()))
                continue
            err = float((go - gr).norm() / gr.norm())
            worst = max(worst, err)
            assert err < 2e-4, (tag, n, err)
        res.update({tag + "eps": eps.numpy(), tag + "noise": noise.numpy(), tag + "loss_kl": np.float32(loss_kl.item()),
                    tag + "loss_duration": np.float32(loss_dur.item()), tag + "loss_pitch": np.float32(float(loss_pitch)),
                    tag + "attn": o["attn"].numpy().astype(np.uint8)})
        for k in keys:
            res[tag + "out/" + k] = out[k].detach().numpy()
        if not use_pitch:
            for k, v in grads.items():
                res["grad/" + k] = v.numpy()
        else:                                                            # second scenario: pitch tensors in full, the rest as norms + evenly spaced samples
            gkeys = sorted(grads)
            flat, off = golden_util.pack_samples(grads, gkeys, 256)
            res.update({"p_grad_keys": np.array(gkeys), "p_grad_samples": flat, "p_grad_offsets": off,
                        "p_grad_norms": np.array([float(grads[k].norm()) for k in gkeys], dtype=np.float32)})
            for k, v in grads.items():
                if k.startswith("pitch_"):
                    res["p_grad/" + k] = v.numpy()
        nz = sum(1 for v in grads.values() if float(v.norm()) >= 1e-5 * v.numel() ** 0.5)
        print("scenario pitch=%d: loss_kl %.5f loss_duration %.5f loss_pitch %.5f; %d / %d gradient tensors non-zero; oracle vs reference worst rel %.2e"
              % (use_pitch, loss_kl.item(), loss_dur.item(), float(loss_pitch), nz, len(grads), worst))
    path = os.path.join(OUT, "xvapitch_acoustic.npz")
    np.savez_compressed(path, **res)
    print("xvapitch_acoustic.npz: %d arrays, %.2f MB" % (len(res), os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    main()
