"""Test infrastructure (used by oracle/gen_golden*.py and tests/ only): how whole-gradient evidence is stored in tests/golden/.

Every gradient tensor of a reference run is recorded (a) by its L2 norm and sum, (b) by an evenly spaced SAMPLE of up to
`n` of its elements (all of them when the tensor has <= n elements), so that a wrong element anywhere in any tensor has a
fixed chance of being seen without committing hundreds of MB, and (c) in full for a named list of tensors.
"""
import numpy as np
import torch


def sample_index(numel, n=2048):
    n_s = min(int(numel), int(n))
    return (torch.arange(n_s, dtype=torch.int64) * int(numel)) // max(n_s, 1)


def sample(t, n=2048):
    f = t.detach().reshape(-1)
    return f[sample_index(f.numel(), n).to(f.device)]


def pack_samples(grads, keys, n=2048):
    """-> (flat float32 samples, int64 offsets of len(keys)+1), keys in the given order."""
    parts = [sample(grads[k], n).float().cpu().numpy() for k in keys]
    off = np.zeros(len(keys) + 1, dtype=np.int64)
    off[1:] = np.cumsum([p.size for p in parts])
    return np.concatenate(parts).astype(np.float32), off


def check_samples(mine, keys, flat, off, n=2048):
    """-> list of (relative L2 error of the sampled elements, key), worst first."""
    errs = []
    for i, k in enumerate(keys):
        ref = torch.from_numpy(np.asarray(flat[off[i]:off[i + 1]])).double()
        got = sample(mine[str(k)], n).double().cpu()
        errs.append((float((got - ref).norm() / ref.norm().clamp_min(1e-30)), str(k)))
    return sorted(errs, reverse=True)
