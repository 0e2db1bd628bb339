"""CPU: the checkpoint layouts of the package equal the reference's, as recorded from the reference's own classes by
oracle/gen_checkpoint_layouts.py (tests/golden/checkpoint_layouts.json): HiFi-GAN generator / MPD / MSD state_dict keys, shapes and
order, both AdamW parameter orders (optim_d = chain(msd, mpd)), the AdamW / Lamb state_dict structure.  (The FastPitch state_dict layout
itself is checked in tests/test_fastpitch_host_cpu.py.)"""
import json
import os

import torch


def _lay(golden_dir):
    return json.load(open(os.path.join(golden_dir, "checkpoint_layouts.json")))


def test_hifigan_tables_match_reference_state_dicts(golden_dir):
    from xva_trainer_amd.hifigan import engine as HE
    lay = _lay(golden_dir)["hifigan"]
    g = [(n, list(sh)) for n, o, c, sh, k in HE.tensor_table(HE.G)]
    assert g == [(k, sh) for k, sh, dt in lay["generator"]]
    d = HE.tensor_table(HE.D)
    for pre in ("mpd", "msd"):
        mine = {n[len(pre) + 1:]: list(sh) for n, o, c, sh, k in d if n.startswith(pre + ".")}
        ref = {k: sh for k, sh, dt in lay[pre]}
        assert mine == ref, pre
    assert all(dt == "float32" for part in ("generator", "mpd", "msd") for _, _, dt in lay[part])


def test_hifigan_optimizer_orders_and_state_dict_format(golden_dir):
    from xva_trainer_amd.hifigan.step import FlatAdamW, optimizer_orders
    lay = _lay(golden_dir)["hifigan"]
    og, od = optimizer_orders()
    assert [t[0] for t in og] == lay["optim_g"]["param_order"]
    assert [t[0] for t in od] == lay["optim_d"]["param_order"]              # msd first, then mpd
    flat = torch.zeros(max(o + c for _, o, c, _ in og))
    opt = FlatAdamW(flat, flat.numel(), order=og)
    assert opt.state_dict()["state"] == {}                                   # torch: no state before the first step
    opt.step_count = 7
    opt.exp_avg.fill_(0.5)
    sd = opt.state_dict()
    assert len(sd["state"]) == lay["optim_g"]["state_count"] == len(og)
    assert set(sd["state"][0]) == set(lay["optim_g"]["state_keys"]) == {"step", "exp_avg", "exp_avg_sq"}
    assert list(sd["state"][0]["exp_avg"].shape) == lay["optim_g"]["state_keys"]["exp_avg"]
    ref_group = lay["optim_g"]["param_group"]
    assert set(sd["param_groups"][0]) == set(ref_group)
    assert sd["param_groups"][0]["params"] == list(range(len(og)))
    # it is a state_dict torch.optim.AdamW itself accepts over parameters of these shapes
    params = [torch.nn.Parameter(torch.zeros(sh)) for _, _, _, sh in og]
    ref = torch.optim.AdamW(params, 2e-4, betas=[0.8, 0.99])
    ref.load_state_dict(sd)
    assert float(ref.state[params[3]]["step"]) == 7.0 and float(ref.state[params[3]]["exp_avg"].flatten()[0]) == 0.5
    # and the other way round
    opt2 = FlatAdamW(torch.zeros_like(flat), flat.numel(), order=og)
    back = ref.state_dict()
    back["param_groups"][0]["lr"] = 1.5e-4
    opt2.load_state_dict(back)
    assert opt2.step_count == 7 and float(opt2.exp_avg[og[3][1]]) == 0.5 and opt2.param_groups[0]["lr"] == 1.5e-4


def test_fastpitch_lamb_state_layout(golden_dir):
    from xva_trainer_amd.fastpitch import engine as E, params as P
    lay = _lay(golden_dir)["fastpitch"]
    assert P.reference_param_order(E.tensor_table()) == lay["optimizer"]["param_order"]
    assert set(lay["optimizer"]["state_keys"]) == {"step", "exp_avg", "exp_avg_sq", "weight_norm", "adam_norm", "trust_ratio"}
    assert lay["checkpoint_keys"] == ["epoch", "iteration", "avg_loss_per_epoch", "training_stage", "state_dict", "optimizer"]
