"""Record the CHECKPOINT LAYOUT of the reference's xVAPitch trainer — state_dict keys / shapes / dtypes of the model (python/xvapitch/model.py:40-215
at the trainer's own switches `--big 1 --pitch 1`, xva_train.py:1098-1132,1421-1425), the parameter order of the two AdamW optimisers
(python/xvapitch/training_util.py:4-57) and the checkpoint dict's keys (xva_train.py:935-952) — by building the reference's own classes here
(build container only; the 130 M-parameter checkpoint itself is not a fixture):

    python oracle/gen_xvapitch_checkpoint_layout.py        -> tests/golden/xvapitch_checkpoint_layout.json

model.py does not import as a module (its header pulls the text front end), so — as in gen_golden_xvapitch_acoustic.py — its classes are compiled from
their source lines in memory and assembled exactly as xVAPitch.__init__ assembles them (same constructor arguments, same attribute order, which fixes
the state_dict / parameters() order)."""
import importlib
import json
import os
import sys
import types
from itertools import chain

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import gen_golden_xvapitch_acoustic as ga, ref_import  # noqa: E402

N_SYMBOLS_KEY = "n_symbols"


def load_reference_disc(hg):
    """VitsDiscriminator (python/xvapitch/model.py:1548-1640), compiled from its source lines in memory like gen_golden_vits_disc.py."""
    src = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "model.py")).read()
    ns = {"torch": torch, "nn": torch.nn, "Conv1d": torch.nn.Conv1d, "DiscriminatorP": hg.DiscriminatorP}
    exec(compile(src[src.index("class DiscriminatorS(torch.nn.Module):"):src.index("def mask_from_lens(lens, max_len= None):")], "model.py:VitsDiscriminator", "exec"), ns)
    return ns["VitsDiscriminator"]


def all_symbols_count():
    """len(ALL_SYMBOLS) (python/xvapitch/text/ipa_to_xvaarpabet.py:103): the text package does not import here (espeak front end), so the module's
    top-level literal assignments are evaluated in order, nothing else."""
    import ast
    src = open(os.path.join(ref_import.REF_ROOT, "python", "xvapitch", "text", "ipa_to_xvaarpabet.py"), encoding="utf8").read()
    ns = {}
    for node in ast.parse(src).body:
        if isinstance(node, (ast.Assign, ast.AugAssign)):
            try:
                exec(compile(ast.Module([node], []), "ipa_to_xvaarpabet.py", "exec"), ns)
            except Exception:
                pass
        if "ALL_SYMBOLS" in ns:
            break
    return len(ns["ALL_SYMBOLS"])


def build(big=1, n_symbols=None, num_languages=31):
    ns = ga.load_reference()
    hg = importlib.import_module("python.xvapitch.hifigan")
    if n_symbols is None:
        n_symbols = all_symbols_count()
    L, Cc = (12, 256) if big else (4, 192)
    args = types.SimpleNamespace(pitch=1, energy=0, flc=0, ow_flow=0, mltts_rc=0, expanded_flow=0, expanded_flow_dim=32, lang_w=1, detach_dp_input=True,
                                 d_vector_dim=512, pe_scaling=0.2, big=big, frozen_vocoder_langs=0, hifi_only=0)

    class M(torch.nn.Module):          # attribute order = xVAPitch.__init__ (model.py:40-215)
        def __init__(self):
            super().__init__()
            self.emb_l = torch.nn.Embedding(num_languages, L)
            self.text_encoder = ns["TextEncoder"](n_symbols, Cc, Cc, 768, 2, 10, 3, 0.1, language_emb_dim=L)
            self.posterior_encoder = ns["PosteriorEncoder"](513, Cc, Cc, kernel_size=5, dilation_rate=1, num_layers=16, cond_channels=512)
            self.flow = ns["ResidualCouplingBlocks"](Cc, Cc, kernel_size=5, dilation_rate=1, num_layers=4, cond_channels=512, args=args)
            self.duration_predictor = ns["StochasticDurationPredictor"](Cc, Cc, 3, 0.5, 4, cond_channels=512, language_emb_dim=L)
            self.waveform_decoder = hg.HifiganGenerator(Cc, 1, "1", [[1, 3, 5], [1, 3, 5], [1, 3, 5]], [3, 7, 11], [16, 16, 4, 4], 512, [8, 8, 2, 2],
                                                        inference_padding=0, cond_channels=512, conv_pre_weight_norm=False, conv_post_weight_norm=False,
                                                        conv_post_bias=False)
            self.disc = load_reference_disc(hg)(use_spectral_norm=False)
            self.pitch_predictor = ns["RelativePositioningPitchEnergyEncoder"](out_channels=1, hidden_channels=Cc + L, hidden_channels_ffn=768, num_heads=2,
                                                                              num_layers=3, kernel_size=3, dropout_p=0.1, conditioning_emb_dim=512)
            self.pitch_emb = torch.nn.Conv1d(1, Cc, kernel_size=3, padding=1)
    return M(), args, n_symbols


def main():
    m, args, n_symbols = build()
    sd = m.state_dict()
    names = {id(p): n for n, p in m.named_parameters()}
    gen_params = chain(m.emb_l.parameters(), m.text_encoder.parameters(), m.duration_predictor.parameters(), m.flow.parameters(),
                       m.posterior_encoder.parameters(), m.waveform_decoder.parameters(), m.pitch_predictor.parameters(), m.pitch_emb.parameters())
    out = {"switches": {"big": 1, "pitch": 1, N_SYMBOLS_KEY: n_symbols, "num_languages": 31, "latent": 256, "lang_dim": 12},
           "checkpoint_keys": ["model", "optimizer", "scaler", "step", "epoch", "lr", "date", "avg_disc_loss_per_epoch", "avg_disc_loss_per_epoch_deltas",
                               "training_stage"],
           "model_extra_keys": ["avg_disc_loss_per_epoch", "avg_disc_loss_per_epoch_deltas"],
           "state_dict": [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()],
           "optimizer0_param_order": [names[id(p)] for p in gen_params],
           "optimizer1_param_order": [names[id(p)] for p in m.disc.parameters()],
           "adamw": {"lr": 0.000175, "lr_disc": 0.0002, "betas": [0.8, 0.99], "eps": 1e-09, "weight_decay": 0.01, "gamma": 0.999875}}
    path = os.path.join(ROOT, "tests", "golden", "xvapitch_checkpoint_layout.json")
    with open(path, "w") as f:
        json.dump(out, f)
    # the registration order of sibling modules / parameters (what fixes parameters() order for ANY layer count): parent pattern -> child -> rank,
    # digits in ModuleList positions replaced by "#"; written as data into the package (xvapitch/param_order.py)
    ranks = {}
    for k in out["optimizer0_param_order"] + out["optimizer1_param_order"]:
        toks = k.split(".")
        for i, t in enumerate(toks):
            if t.isdigit():
                continue
            parent = ".".join("#" if x.isdigit() else x for x in toks[:i])
            d = ranks.setdefault(parent, {})
            d.setdefault(t, len(d))
    with open(os.path.join(ROOT, "xva-trainer_amd", "xvapitch", "param_order.py"), "w") as f:
        f.write('"""Registration order of sibling sub-modules / parameters in the reference xVAPitch model (python/xvapitch/model.py:40-215 and the classes it\n'
                'builds): parent key pattern ("#" = a ModuleList index) -> {child name: rank}.  torch\'s `parameters()` walks a module\'s own parameters, then its\n'
                'children, in this order, which fixes the parameter order of the two AdamW state_dicts inside `xVAPitch_*.pt` checkpoints whatever the layer\n'
                'counts.  DATA generated by oracle/gen_xvapitch_checkpoint_layout.py from the reference\'s own classes; do not edit."""\n')
        f.write("SIBLING_RANK = " + json.dumps(ranks, indent=0).replace("\n", "") + "\n\n\n")
        f.write('def order_key(name):\n    """sort key reproducing parameters() order for a reference state_dict key"""\n'
                '    toks, key = name.split("."), []\n    for i, t in enumerate(toks):\n        if t.isdigit():\n            key.append(int(t))\n'
                '        else:\n            parent = ".".join("#" if x.isdigit() else x for x in toks[:i])\n'
                '            key.append(SIBLING_RANK.get(parent, {}).get(t, 1 << 20))\n    return key\n')
    n = sum(v.numel() for v in sd.values())
    print("wrote %s: %d tensors, %.1f M elements; optimizer0 %d params, optimizer1 %d params; n_symbols %s"
          % (path, len(sd), n / 1e6, len(out["optimizer0_param_order"]), len(out["optimizer1_param_order"]), n_symbols))


if __name__ == "__main__":
    main()
