"""Flat-buffer <-> checkpoint (reference state_dict) conversion for FastPitch.

The engine keeps all 181 parameters in one flat fp32 buffer (16-byte aligned tensors; conv k=3 weights tap-major
[Cout][3][Cin]).  Checkpoints keep the reference's keys, shapes and dtypes (python/fastpitch1_1/xva_train.py:1001-1016):
conversion happens only here, at the save/load boundary.
"""
import math

import torch

BUFFER_KEYS = ("pitch_mean", "pitch_std", "encoder.pos_emb.inv_freq", "decoder.pos_emb.inv_freq")


def reference_param_order(table):
    """Order of FastPitch.named_parameters() in the reference (model.py:181-265 construction order); optimizer
    state_dicts index parameters by this order."""
    names = [t[0] for t in table]
    groups = ["encoder.", "duration_predictor.", "decoder.", "pitch_predictor.", "pitch_emb.", "energy_predictor.", "energy_emb.",
              "proj.", "attention."]
    out = []
    for g in groups:
        out += [n for n in names if n.startswith(g)]
    assert len(out) == len(names)
    return out


def to_flat(sd, table, flat):
    """Copy reference-layout tensors from `sd` into the flat buffer (in place). Missing keys raise KeyError."""
    with torch.no_grad():
        for name, off, n, shape, kind in table:
            t = sd[name]
            if tuple(t.shape) != tuple(shape):
                raise ValueError("%s: checkpoint shape %s != %s" % (name, tuple(t.shape), shape))
            t = t.to(device=flat.device, dtype=torch.float32)
            if kind == 1:
                t = t.permute(0, 2, 1)
            flat[off:off + n].copy_(t.reshape(-1))
    return flat


def from_flat(flat, table, names=None, dtype=None):
    """Reference-layout tensors (fresh storage) keyed by reference names."""
    out = {}
    with torch.no_grad():
        for name, off, n, shape, kind in table:
            if names is not None and name not in names:
                continue
            v = flat[off:off + n]
            if kind == 1:
                v = v.view(shape[0], shape[2], shape[1]).permute(0, 2, 1)
            t = v.reshape(shape).clone()
            out[name] = t.to(dtype) if dtype is not None else t
    return out


def default_init_(flat, table, seed=1234):
    """torch-default-like initialisation (nn.Linear / nn.Conv1d kaiming-uniform bounds, LayerNorm 1/0, Embedding N(0,1) with
    the padding row zero) written straight into the flat buffer."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        flat.zero_()
        for name, off, n, shape, kind in table:
            if name.endswith("word_emb.weight"):
                w = torch.randn(shape, generator=g)
                w[0] = 0
            elif ".norm." in name or "layer_norm" in name:
                w = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            else:
                if name.endswith("bias"):
                    wname = name[:-4] + "weight"
                    wshape = next(t[3] for t in table if t[0] == wname)
                else:
                    wshape = shape
                fan_in = 1
                for s in wshape[1:]:
                    fan_in *= s
                bound = 1.0 / math.sqrt(fan_in)
                w = (torch.rand(shape, generator=g) * 2 - 1) * bound
            if kind == 1:
                w = w.permute(0, 2, 1)
            flat[off:off + n].copy_(w.reshape(-1).to(flat.device))
    return flat
