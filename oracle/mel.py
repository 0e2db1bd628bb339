"""CPU oracle for the mel front end (floating point; torch fp32 on CPU).

Restates:
  M1  TacotronSTFT.mel_spectrogram      python/fastpitch1_1/common/layers.py:121-138
      STFT.__init__/transform           python/fastpitch1_1/common/stft.py:53-114
      dynamic_range_compression         python/fastpitch1_1/common/audio_processing.py:105-111
  M2  mel_spectrogram                   python/hifigan/meldataset.py:217-240
  M3  TorchSTFT.__call__                python/xvapitch/audio.py:138-181
and the un-vendored librosa==0.8.1 `filters.mel` (PARITY UNPINNED: published Slaney
algorithm, no reference vector exists for it).
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---- librosa 0.8.1 filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm='slaney') ----
def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        log_t = f >= min_log_hz
        mels[log_t] = min_log_mel + np.log(f[log_t] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    log_t = m >= min_log_mel
    freqs[log_t] = min_log_hz * np.exp(logstep * (m[log_t] - min_log_mel))
    return freqs


def slaney_mel_filterbank(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    if fmax is None:
        fmax = float(sr) / 2
    n_mels = int(n_mels)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]
    return weights


def peak_normalize(x):
    """librosa.util.normalize(x) default (norm=inf, axis=0) for a 1-D signal (hifigan/meldataset.py:349)."""
    m = np.max(np.abs(x))
    return x / m if m > np.finfo(x.dtype).tiny else x


def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True) (common/stft.py:75)."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def stft_forward_basis(filter_length=1024, win_length=1024):
    """common/stft.py:61-84: rows [real(0..cutoff-1) | imag(0..cutoff-1)] of fft(eye), fp32, times fp32 window."""
    fourier_basis = np.fft.fft(np.eye(filter_length))
    cutoff = filter_length // 2 + 1
    fourier_basis = np.vstack([np.real(fourier_basis[:cutoff, :]), np.imag(fourier_basis[:cutoff, :])])
    forward_basis = torch.FloatTensor(fourier_basis[:, None, :])
    win = hann_periodic(win_length)
    assert win_length == filter_length
    forward_basis *= torch.from_numpy(win).float()
    return forward_basis  # (2*cutoff, 1, filter_length)


def mel_m1(y, sr=22050, n_fft=1024, hop=256, n_mel=80, fmin=0.0, fmax=8000.0):
    """TacotronSTFT.mel_spectrogram. y: (B, N) float32 in [-1, 1] -> (B, n_mel, 1 + N//hop)."""
    assert y.min() >= -1 and y.max() <= 1
    basis = stft_forward_basis(n_fft, n_fft)
    x = y.view(y.size(0), 1, y.size(1))
    x = F.pad(x.unsqueeze(1), (n_fft // 2, n_fft // 2, 0, 0), mode="reflect").squeeze(1)
    ft = F.conv1d(x, basis, stride=hop, padding=0)
    cutoff = n_fft // 2 + 1
    mag = torch.sqrt(ft[:, :cutoff] ** 2 + ft[:, cutoff:] ** 2)
    melb = torch.from_numpy(slaney_mel_filterbank(sr, n_fft, n_mel, fmin, fmax)).float()
    return torch.log(torch.clamp(torch.matmul(melb, mag), min=1e-5))


def mel_m2(y, sr=22050, n_fft=1024, hop=256, win=1024, n_mel=80, fmin=0.0, fmax=8000.0):
    """hifigan mel_spectrogram(center=False). y: (B, N) -> (B, n_mel, N//hop). fmax=None -> sr/2."""
    melb = torch.from_numpy(slaney_mel_filterbank(sr, n_fft, n_mel, fmin, fmax)).float()
    p = int((n_fft - hop) / 2)
    x = F.pad(y.unsqueeze(1), (p, p), mode="reflect").squeeze(1)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    spec = torch.view_as_real(spec)
    spec = torch.sqrt(spec.pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(melb, spec), min=1e-5))


def mel_m3(y, sr=22050, n_fft=1024, hop=256, win=1024, n_mel=80, fmin=0.0, fmax=8000.0):
    """xvapitch TorchSTFT (center=True reflect, use_mel=True, do_amp_to_db=True). y: (B, N) -> (B, n_mel, 1+N//hop)."""
    melb = torch.from_numpy(slaney_mel_filterbank(sr, n_fft, n_mel, fmin, fmax)).float()
    o = torch.stft(y, n_fft, hop, win, torch.hann_window(win), center=True, pad_mode="reflect", normalized=False,
                   onesided=True, return_complex=True)
    o = torch.view_as_real(o)
    S = torch.sqrt(torch.clamp(o[..., 0] ** 2 + o[..., 1] ** 2, min=1e-8))
    return torch.log(torch.clamp(torch.matmul(melb.to(S), S), min=1e-5))


def linear_m3(y, n_fft=1024, hop=256, win=1024):
    """xVAPitch's linear spectrogram, the posterior encoder's input (python/xvapitch/audio.py:138-181 TorchSTFT with use_mel=False: the same centred
    STFT and clamp as mel_m3, no filterbank, no log).  y: (B, N) -> (B, 513, 1 + N // hop)."""
    o = torch.stft(y, n_fft, hop, win, torch.hann_window(win), center=True, pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    o = torch.view_as_real(o)
    return torch.sqrt(torch.clamp(o[..., 0] ** 2 + o[..., 1] ** 2, min=1e-8))


def synth_wave(n_samples, seed, sr=22050):
    """SURVEY.md §8d synthetic clip: gliding f0 100->300 Hz + 3 harmonics (-6 dB/oct) + noise, int16-quantised."""
    rng = np.random.RandomState(seed)
    t = np.arange(n_samples) / sr
    f0 = np.linspace(100.0, 300.0, n_samples)
    phase = 2 * np.pi * np.cumsum(f0) / sr
    sig = np.zeros(n_samples)
    for h in range(1, 5):
        sig += (0.5 / h) * np.sin(h * phase)
    sig = 0.5 * sig / 0.9 + 0.05 * rng.randn(n_samples)
    sig = np.clip(sig, -1.0, 1.0)
    q = np.round(sig * 32767.0).astype(np.int16)
    return (q.astype(np.float32) / 32768.0)
