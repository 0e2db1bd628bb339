"""Synthetic inputs of BASELINE.json's shapes (there is no network for datasets): 22050 Hz clips, FastPitch batches
shaped like TTSCollate's output (python/fastpitch1_1/fastpitch/data_function.py:565-695), HiFi-GAN segments."""
import math

import numpy as np
import torch

N_SYMBOLS = 148
N_MEL = 80


def synth_wave(n_samples, seed, sr=22050):
    """Gliding f0 100->300 Hz + 3 harmonics at -6 dB/octave + noise, clipped and int16-quantised."""
    rng = np.random.RandomState(seed)
    f0 = np.linspace(100.0, 300.0, n_samples)
    phase = 2 * np.pi * np.cumsum(f0) / sr
    sig = np.zeros(n_samples)
    for h in range(1, 5):
        sig += (0.5 / h) * np.sin(h * phase)
    sig = 0.5 * sig / 0.9 + 0.05 * rng.randn(n_samples)
    q = np.round(np.clip(sig, -1.0, 1.0) * 32767.0).astype(np.int16)
    return q.astype(np.float32) / 32768.0


def fastpitch_batch(B, T_text, T_mel, seed, ragged=False, mel=None):
    """dict(text, in_lens, mel_tgt, mel_lens, pitch, energy, durs) on the CPU; durations sum to each item's mel length."""
    g = torch.Generator().manual_seed(seed)
    if ragged and B > 1:
        in_lens = torch.sort(torch.randint(max(2, T_text // 3), T_text + 1, (B,), generator=g), descending=True).values
        in_lens[0] = T_text
    else:
        in_lens = torch.full((B,), T_text, dtype=torch.long)
    text = torch.zeros(B, T_text, dtype=torch.long)
    durs = torch.zeros(B, T_text, dtype=torch.long)
    mel_lens = torch.zeros(B, dtype=torch.long)
    for b in range(B):
        L = int(in_lens[b])
        text[b, :L] = torch.randint(1, N_SYMBOLS, (L,), generator=g)
        tm = T_mel if b == 0 else max(L, int(T_mel * L / T_text))
        extra = torch.multinomial(torch.ones(L), tm - L, replacement=True, generator=g) if tm > L else torch.zeros(0, dtype=torch.long)
        d = torch.ones(L, dtype=torch.long)
        d.scatter_add_(0, extra, torch.ones_like(extra))
        durs[b, :L] = d
        mel_lens[b] = tm
    Tm = int(mel_lens.max())
    if mel is None:
        mel = torch.zeros(B, N_MEL, Tm)
        for b in range(B):
            tm = int(mel_lens[b])
            mel[b, :, :tm] = torch.clamp(torch.randn(N_MEL, tm, generator=g) * 2 - 5, math.log(1e-5), 2.0)
    pitch = torch.zeros(B, 1, Tm)
    for b in range(B):
        tm = int(mel_lens[b])
        p = torch.randn(tm, generator=g)
        p[torch.rand(tm, generator=g) < 0.3] = 0.0
        pitch[b, 0, :tm] = p
    energy = torch.norm(mel.float(), dim=1, p=2)
    for b in range(B):
        energy[b, int(mel_lens[b]):] = 0
    return {"text": text, "in_lens": in_lens, "mel_tgt": mel, "mel_lens": mel_lens, "pitch": pitch, "energy": energy, "durs": durs}
