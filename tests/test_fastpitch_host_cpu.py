"""CPU: host-side logic of the FastPitch mirror — checkpoint layout, flat<->reference conversion, parameter table,
stage freezing, LR schedule.  (No kernel is launched here.)"""
import json
import os

import torch


def _layout(golden_dir):
    return json.load(open(os.path.join(golden_dir, "fastpitch_state_dict_layout.json")))


def test_state_dict_matches_reference_layout(golden_dir):
    """Keys, order, shapes and dtypes equal FastPitch().state_dict() of the reference (recorded by importing it)."""
    from xva_trainer_amd.fastpitch.model import FastPitch
    lay = _layout(golden_dir)
    m = FastPitch(compute="fp32")
    sd = m.state_dict()
    assert list(sd.keys()) == lay["keys"]
    assert [list(v.shape) for v in sd.values()] == lay["shapes"]
    assert [str(v.dtype) for v in sd.values()] == lay["dtypes"]


def test_state_dict_roundtrip_and_module_prefix(golden_dir):
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch.model import FastPitch
    sd = ofp.init_state_dict(3)
    sd["pitch_mean"] = torch.tensor([211.5]); sd["pitch_std"] = torch.tensor([55.25])
    m = FastPitch(compute="fp32")
    m.load_state_dict({"module." + k: v for k, v in sd.items()})   # DataParallel-style prefix (xva_train.py:1063-1070)
    out = m.state_dict()
    for k, v in sd.items():
        assert torch.equal(out[k], v), k
    # tap-major internal layout: flat[off + (o*3 + k)*Cin + i] == W[o, i, k]
    name, off, n, shape, kind = next(t for t in m._table if t[0] == "decoder.layers.2.pos_ff.CoreNet.0.weight")
    assert kind == 1 and shape == (1536, 384, 3)
    W = sd[name]
    assert m.flat.data[off + (5 * 3 + 2) * 384 + 7].item() == W[5, 7, 2].item()


def test_param_table_and_reference_order(golden_dir):
    from xva_trainer_amd.fastpitch import engine as E, params as P
    lay = _layout(golden_dir)
    table = E.tensor_table()
    assert len(table) == 181
    assert P.reference_param_order(table) == lay["param_order"]
    assert sum(t[2] for t in table) == 46268449 or sum(t[2] for t in table) > 46e6
    for name, off, n, shape, kind in table:
        assert off % 4 == 0
    offs = sorted((t[1], t[1] + t[2]) for t in table)
    assert all(a[1] <= b[0] for a, b in zip(offs, offs[1:]))


def test_trainable_ranges_match_reference_freezing():
    """xva_train.py:589-672: which module groups train in each stage."""
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import engine as E
    table = E.tensor_table()
    keys = [t[0] for t in table]
    for stage in (2, 3, 4):
        rng = E.trainable_ranges(stage)
        mine = {t[0] for t in table if any(b <= t[1] < e for b, e in rng)}
        assert mine == set(ofp.trainable_names(keys, stage)), stage


def test_lamb_state_dict_layout():
    from xva_trainer_amd.fastpitch import engine as E
    table = E.tensor_table()
    assert any(t[0] == "proj.weight" for t in table)
