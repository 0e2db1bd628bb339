"""Record the CHECKPOINT LAYOUTS of the reference — keys, shapes, dtypes, parameter order, optimizer state structure — by building
the reference's own classes (imported from /root/reference, oracle/ref_import.py) and serialising their state_dict() metadata.
Build container only:

    python oracle/gen_checkpoint_layouts.py

Writes tests/golden/fastpitch_state_dict_layout.json (the 185-entry FastPitch state_dict; python/fastpitch1_1/xva_train.py:1001-1016)
and tests/golden/checkpoint_layouts.json (FastPitch full checkpoint incl. the Lamb optimizer state, python/fastpitch1_1/lamb.py:63-100;
HiFi-GAN `g_########` / `do_########` incl. both AdamW states in the reference's parameter order, python/hifigan/xva_train.py:298-300,
570-601).  The real checkpoints are 185 MB / 340 MB: the fixtures hold the layout only (data, not code); tests rebuild files of
that exact layout with seeded values and load them into the trainers (tests/test_checkpoint_compat_*.py).
"""
import itertools
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sd_layout(sd):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in sd.items()]


def optim_layout(opt, named_params):
    """param order by NAME (via identity with the module's named parameters) + the per-param state keys / shapes + param_group keys."""
    names = {id(p): n for n, p in named_params}
    for group in opt.param_groups:
        for p in group["params"]:
            p.grad = torch.zeros_like(p)
    opt.step()
    sd = opt.state_dict()
    order = [names[id(p)] for g in opt.param_groups for p in g["params"]]
    st0 = sd["state"][0]
    return {"param_order": order,
            "state_keys": {k: (list(v.shape) if torch.is_tensor(v) else type(v).__name__) for k, v in st0.items()},
            "state_count": len(sd["state"]),
            "param_group": {k: (v if not isinstance(v, (list, tuple)) or k != "params" else "range(n)") for k, v in sd["param_groups"][0].items()}}


class _H(dict):
    __getattr__ = dict.__getitem__


def main():
    ns = ref_import.import_reference()
    torch.manual_seed(0)
    fp = ns.FastPitch(logger=None)
    layout = sd_layout(fp.state_dict())
    sd = fp.state_dict()
    with open(os.path.join(OUT, "fastpitch_state_dict_layout.json"), "w") as f:
        json.dump({"keys": list(sd.keys()), "shapes": [list(v.shape) for v in sd.values()], "dtypes": [str(v.dtype) for v in sd.values()],
                   "param_order": [n for n, _ in fp.named_parameters()]}, f)
    lamb = ns.Lamb(fp.parameters(), lr=0.1, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    out = {"fastpitch": {"checkpoint_keys": ["epoch", "iteration", "avg_loss_per_epoch", "training_stage", "state_dict", "optimizer"],
                         "state_dict": layout, "optimizer": optim_layout(lamb, list(fp.named_parameters()))}}
    hm = ns.hifigan_models
    h = _H(json.load(open(os.path.join(ref_import.REF_ROOT, "python", "hifigan", "config_v1.json"))))
    h["USE_EMB_CONDITIONING"] = False
    gen, mpd, msd = hm.Generator(h), hm.MultiPeriodDiscriminator(), hm.MultiScaleDiscriminator()
    og = torch.optim.AdamW(gen.parameters(), 2e-4, betas=[0.8, 0.99])
    od = torch.optim.AdamW(itertools.chain(msd.parameters(), mpd.parameters()), 2e-4, betas=[0.8, 0.99])
    torch.optim.lr_scheduler.ExponentialLR(og, gamma=0.999, last_epoch=-1)
    torch.optim.lr_scheduler.ExponentialLR(od, gamma=0.999, last_epoch=-1)
    d_named = [("msd." + n, p) for n, p in msd.named_parameters()] + [("mpd." + n, p) for n, p in mpd.named_parameters()]
    out["hifigan"] = {"g_keys": ["generator"], "do_keys": ["mpd", "msd", "optim_g", "optim_d", "steps", "epoch", "avg_loss_per_epoch", "ckpts_finetuned"],
                      "generator": sd_layout(gen.state_dict()), "mpd": sd_layout(mpd.state_dict()), "msd": sd_layout(msd.state_dict()),
                      "optim_g": optim_layout(og, list(gen.named_parameters())), "optim_d": optim_layout(od, d_named)}
    with open(os.path.join(OUT, "checkpoint_layouts.json"), "w") as f:
        json.dump(out, f)
    print("fastpitch: %d tensors, lamb state %s" % (len(layout), out["fastpitch"]["optimizer"]["state_keys"]))
    print("hifigan: G %d / mpd %d / msd %d tensors; optim_d first params: %s" % (len(out["hifigan"]["generator"]), len(out["hifigan"]["mpd"]),
                                                                              len(out["hifigan"]["msd"]), out["hifigan"]["optim_d"]["param_order"][:3]))
    print("adamw param_group:", out["hifigan"]["optim_g"]["param_group"])


if __name__ == "__main__":
    main()
