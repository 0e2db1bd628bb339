"""Batch sources for the trainers.

The reference builds batches on the CPU from `metadata.csv` + `wavs/` through text cleaners, CMUdict, pyin pitch extraction
and `.npy` side caches (python/fastpitch1_1/fastpitch/data_function.py, python/hifigan/meldataset.py) — string / file
preprocessing that is outside the accelerated path.  The trainers therefore take any iterable that yields
batches in the reference's collate format; a maintainer plugs the reference's own DataLoader in through `loader_factory`
(INTEGRATION.md), and the synthetic loaders below serve benchmarks and tests (there is no dataset offline).
"""
import numpy as np
import torch

from . import synthetic


def beta_binomial_prior_distribution(phoneme_count, mel_count, scaling=1.0):
    """The (mel_count, phoneme_count) beta-binomial attention prior of one utterance — same call and formula as
    python/fastpitch1_1/fastpitch/data_function.py:84-94 (scipy.stats.betabinom)."""
    from scipy.stats import betabinom
    x = np.arange(0, phoneme_count)
    rows = [betabinom(phoneme_count, scaling * i, scaling * (mel_count + 1 - i)).pmf(x) for i in range(1, mel_count + 1)]
    return torch.tensor(np.array(rows))


def collate_attn_prior(in_lens, mel_lens):
    """TTSCollate's zero-padded (B, max_mel, max_text) stack of the per-item priors (data_function.py:600-609)."""
    out = torch.zeros(len(in_lens), int(max(mel_lens)), int(max(in_lens)))
    for b, (L, M) in enumerate(zip(in_lens, mel_lens)):
        out[b, :int(M), :int(L)] = beta_binomial_prior_distribution(int(L), int(M)).float()
    return out


class SyntheticFastPitchLoader:
    """Yields (x, y, num_frames)-ready dict batches shaped like TTSCollate's output (data_function.py:565-695).
    with_prior adds the beta-binomial `attn_prior` training stage 1 consumes."""

    def __init__(self, batch_size, n_batches=8, t_text=150, t_mel=860, seed=1234, ragged=True, with_prior=False):
        self.batches = [synthetic.fastpitch_batch(batch_size, t_text, t_mel, seed + i, ragged=ragged) for i in range(n_batches)]
        if with_prior:
            for b in self.batches:
                b["attn_prior"] = collate_attn_prior(b["in_lens"].tolist(), b["mel_lens"].tolist())

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        return iter(self.batches)


class SyntheticHifiLoader:
    """Yields (wav (B, segment) fp32 in [-1, 1]) crops like MelDataset.__getitem__ (meldataset.py:340-373); the mels
    are computed on the GPU by the trainer (the reference computes them on the CPU in the dataset)."""

    def __init__(self, batch_size, n_batches=8, segment=8192, seed=4321):
        self.items = []
        for i in range(n_batches):
            wav = np.stack([synthetic.synth_wave(segment, seed + i * batch_size + j) for j in range(batch_size)])
            wav = wav / np.abs(wav).max(axis=1, keepdims=True) * 0.95
            self.items.append(torch.from_numpy(wav.astype(np.float32)))

    def __len__(self):
        return len(self.items)

    def __iter__(self):
        return iter(self.items)
