"""Shared helpers for the FastPitch GPU parity tests."""
import numpy as np
import torch


def load_case(golden_dir, name):
    import os
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    return g, batch


def build_engine(sd, compute="fp32"):
    from xva_trainer_amd.fastpitch import engine as E, params as P
    eng = E.FastPitchEngine("cuda", compute)
    flat = torch.zeros(eng.total, device="cuda")
    P.to_flat(sd, eng.table, flat)
    grads = torch.zeros_like(flat)
    return eng, flat, grads


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def grad_report(eng, grads, ref_grads, tol):
    """Compare per-tensor gradients (reference layout dict) and return a list of (name, relerr) failures."""
    from xva_trainer_amd.fastpitch import params as P
    mine = P.from_flat(grads, eng.table)
    bad = []
    worst = (None, 0.0)
    for k, g in ref_grads.items():
        r = ((mine[k].double().cpu() - g.double()).norm() / g.double().norm().clamp_min(1e-30)).item()
        if r > worst[1]:
            worst = (k, r)
        if not (r < tol):
            bad.append((k, r))
    for name, off, n, shape, kind in eng.table:
        if name not in ref_grads:
            z = mine[name].abs().max().item()
            if z != 0.0:
                bad.append((name + " (should have no grad)", z))
    return bad, worst
