"""Host-side mirror of the reference's mel front ends, running on libxvahip (HIP, gfx950).

Same names / argument meaning as the reference so it drops into the trainers:
  TacotronSTFT(...).mel_spectrogram(y)     python/fastpitch1_1/common/layers.py:100-138
  mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False)
                                            python/hifigan/meldataset.py:217-240
  TorchSTFTMel(...)                         python/xvapitch/audio.py:138-181 (use_mel, amp_to_db)
The windowed-DFT matrix and the Slaney filterbank are module buffers exactly as in the
reference (STFT.forward_basis, TacotronSTFT.mel_basis); only the arithmetic moved to HIP.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib


def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    if f.ndim:
        sel = f >= min_log_hz
        mels[sel] = min_log_mel + np.log(f[sel] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz, logstep = 1000.0, np.log(6.4) / 27.0
    min_log_mel = min_log_hz / f_sp
    sel = m >= min_log_mel
    freqs[sel] = min_log_hz * np.exp(logstep * (m[sel] - min_log_mel))
    return freqs


def librosa_mel_fn(sr, n_fft, n_mels=128, fmin=0.0, fmax=None):
    """librosa 0.8.1 filters.mel (htk=False, norm='slaney') — the dependency the reference pins in reqs_cpu.txt."""
    if fmax is None:
        fmax = float(sr) / 2
    nb = 1 + n_fft // 2
    weights = np.zeros((int(n_mels), nb), dtype=np.float32)
    fftfreqs = np.linspace(0, float(sr) / 2, nb, endpoint=True)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), int(n_mels) + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    for i in range(int(n_mels)):
        weights[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    weights *= (2.0 / (mel_f[2:int(n_mels) + 2] - mel_f[:int(n_mels)]))[:, None]
    return weights


def _dft_basis(n_fft, window):
    """[real | imag] rows of fft(eye(n_fft)), fp32, times the fp32 window (common/stft.py:61-84)."""
    fb = np.fft.fft(np.eye(n_fft))
    cutoff = n_fft // 2 + 1
    fb = np.vstack([np.real(fb[:cutoff]), np.imag(fb[:cutoff])])
    basis = torch.from_numpy(fb.astype(np.float32))
    return (basis * window.float()[None, :]).contiguous()


def _pad_mel_basis(mel_basis):
    nb = mel_basis.shape[1]
    ldm = (nb + 31) // 32 * 32
    out = torch.zeros(mel_basis.shape[0], ldm, dtype=torch.float32)
    out[:, :nb] = mel_basis
    return out


class _MelEngine(torch.nn.Module):
    def __init__(self, n_fft, hop, n_mel, pad, mag_eps_add, mag_clamp_min, window, mel_basis):
        super().__init__()
        self.cfg = _lib.MelConfig(n_fft, hop, n_mel, pad, mag_eps_add, mag_clamp_min, 1e-5)
        self.register_buffer("forward_basis", _dft_basis(n_fft, window), persistent=False)
        self.register_buffer("mel_basis", torch.as_tensor(mel_basis).float().contiguous(), persistent=False)
        self.register_buffer("_mel_basis_padded", _pad_mel_basis(self.mel_basis), persistent=False)
        self._ws = None

    def num_frames(self, n_samples):
        return int(_lib.lib.xva_mel_num_frames(C.byref(self.cfg), int(n_samples)))

    def forward(self, y):
        _lib.require_cuda(y)
        if self.forward_basis.device != y.device:
            self.to(y.device)
        y = y.float()
        if y.stride(-1) != 1 or y.stride(0) % 4 != 0 or y.data_ptr() % 16 != 0:
            ld = (y.size(1) + 3) // 4 * 4
            buf = torch.empty(y.size(0), ld, device=y.device, dtype=torch.float32)
            buf[:, :y.size(1)] = y
            y_buf, ldy = buf, ld
        else:
            y_buf, ldy = y, y.stride(0)
        B, N = y.shape
        T = self.num_frames(N)
        if T <= 0:
            raise _lib.XvaError("mel: bad clip length %d" % N)
        need = int(_lib.lib.xva_mel_workspace_bytes(C.byref(self.cfg), B, N))
        if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != y.device:
            self._ws = torch.empty((need + 3) // 4, device=y.device, dtype=torch.float32)
        out = torch.empty(B, self.cfg.n_mel, T, device=y.device, dtype=torch.float32)
        rc = _lib.lib.xva_mel_spectrogram(C.byref(self.cfg), _lib.ptr(y_buf), B, N, ldy, _lib.ptr(self.forward_basis),
                                          _lib.ptr(self._mel_basis_padded), _lib.ptr(out), _lib.ptr(self._ws),
                                          self._ws.numel() * 4, _lib.stream_ptr())
        _lib.check(rc, "xva_mel_spectrogram")
        return out


def _engine_linear(self, y):
    _lib.require_cuda(y)
    if self.forward_basis.device != y.device:
        self.to(y.device)
    y = y.float().contiguous()
    B, N = y.shape
    if y.stride(0) % 4 != 0 or y.data_ptr() % 16 != 0:
        ld = (N + 3) // 4 * 4
        buf = torch.zeros(B, ld, device=y.device, dtype=torch.float32)
        buf[:, :N] = y
        y = buf
    T = self.num_frames(N)
    need = int(_lib.lib.xva_mel_workspace_bytes(C.byref(self.cfg), B, N))
    if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != y.device:
        self._ws = torch.empty((need + 3) // 4, device=y.device, dtype=torch.float32)
    out = torch.empty(B, self.cfg.n_fft // 2 + 1, T, device=y.device, dtype=torch.float32)
    _lib.check(_lib.lib.xva_linear_spectrogram(C.byref(self.cfg), _lib.ptr(y), B, N, y.stride(0), _lib.ptr(self.forward_basis), _lib.ptr(out),
                                               _lib.ptr(self._ws), self._ws.numel() * 4, _lib.stream_ptr()), "xva_linear_spectrogram")
    return out


_lib.lib.xva_linear_spectrogram.restype = C.c_int32
_lib.lib.xva_linear_spectrogram.argtypes = [C.POINTER(_lib.MelConfig), C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_int64, C.c_void_p]
_MelEngine.linear = _engine_linear


def _engine_linear_ragged(self, y, lens):
    """y (B, Nmax) float clips, zero-padded; lens (B,) valid samples.  -> ((B, n_fft / 2 + 1, 1 + Nmax // hop) each clip's OWN spectrogram, zeros
    after its 1 + lens // hop frames, n_frames (B,) int32)."""
    _lib.require_cuda(y, lens)
    if self.forward_basis.device != y.device:
        self.to(y.device)
    y = y.float().contiguous()
    B, N = y.shape
    if y.stride(0) % 4 != 0 or y.data_ptr() % 16 != 0:
        ld = (N + 3) // 4 * 4
        buf = torch.zeros(B, ld, device=y.device, dtype=torch.float32)
        buf[:, :N] = y
        y = buf
    lens = lens.to(device=y.device, dtype=torch.int32).contiguous()
    T = self.num_frames(N)
    need = int(_lib.lib.xva_mel_workspace_bytes(C.byref(self.cfg), B, N))
    if self._ws is None or self._ws.numel() * 4 < need or self._ws.device != y.device:
        self._ws = torch.empty((need + 3) // 4, device=y.device, dtype=torch.float32)
    out = torch.empty(B, self.cfg.n_fft // 2 + 1, T, device=y.device, dtype=torch.float32)
    n_frames = torch.empty(B, device=y.device, dtype=torch.int32)
    _lib.check(_lib.lib.xva_linear_spectrogram_ragged(C.byref(self.cfg), _lib.ptr(y), _lib.ptr(lens), B, N, y.stride(0), _lib.ptr(self.forward_basis),
                                                      _lib.ptr(out), _lib.ptr(n_frames), _lib.ptr(self._ws), self._ws.numel() * 4, _lib.stream_ptr()),
               "xva_linear_spectrogram_ragged")
    return out, n_frames


_lib.lib.xva_linear_spectrogram_ragged.restype = C.c_int32
_lib.lib.xva_linear_spectrogram_ragged.argtypes = [C.POINTER(_lib.MelConfig), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                                   C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
_MelEngine.linear_ragged = _engine_linear_ragged


def _hann_periodic(n):
    return torch.from_numpy(0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n))


class TacotronSTFT(torch.nn.Module):
    """Drop-in for python/fastpitch1_1/common/layers.py:100-138 (mel_spectrogram only)."""

    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050,
                 mel_fmin=0.0, mel_fmax=8000.0):
        super().__init__()
        assert win_length == filter_length
        self.n_mel_channels = n_mel_channels
        self.sampling_rate = sampling_rate
        mel_basis = librosa_mel_fn(sampling_rate, filter_length, n_mel_channels, mel_fmin, mel_fmax)
        self.engine = _MelEngine(filter_length, hop_length, n_mel_channels, filter_length // 2, 0.0, 0.0,
                                 _hann_periodic(win_length), mel_basis)

    @property
    def mel_basis(self):
        return self.engine.mel_basis

    def mel_spectrogram(self, y):
        """y: (B, T) in [-1, 1] on the GPU -> (B, n_mel_channels, 1 + T // hop)."""
        assert torch.min(y.data) >= -1
        assert torch.max(y.data) <= 1
        return self.engine(y)


_hifi_engines = {}


def mel_spectrogram(y, n_fft, num_mels, sampling_rate, hop_size, win_size, fmin, fmax, center=False):
    """Drop-in for python/hifigan/meldataset.py:217-240 (forward). y: (B, T) -> (B, num_mels, T // hop_size)."""
    assert not center and win_size == n_fft
    key = (n_fft, num_mels, sampling_rate, hop_size, fmin, fmax, str(y.device))
    eng = _hifi_engines.get(key)
    if eng is None:
        mel_basis = librosa_mel_fn(sampling_rate, n_fft, num_mels, fmin, fmax)
        eng = _MelEngine(n_fft, hop_size, num_mels, int((n_fft - hop_size) / 2), 1e-9, 0.0, torch.hann_window(win_size),
                         mel_basis).to(y.device)
        _hifi_engines[key] = eng
    return eng(y)


class TorchSTFTMel(torch.nn.Module):
    """xvapitch TorchSTFT with use_mel=True, do_amp_to_db=True (python/xvapitch/audio.py:138-181)."""

    def __init__(self, n_fft=1024, hop_length=256, win_length=1024, sample_rate=22050, mel_fmin=0.0, mel_fmax=8000.0,
                 n_mels=80):
        super().__init__()
        assert win_length == n_fft
        mel_basis = librosa_mel_fn(sample_rate, n_fft, n_mels, mel_fmin, mel_fmax)
        self.engine = _MelEngine(n_fft, hop_length, n_mels, n_fft // 2, 0.0, 1e-8, torch.hann_window(win_length), mel_basis)

    def forward(self, x):
        if x.ndim == 3:
            x = x.squeeze(1)
        return self.engine(x)

    def linear(self, x):
        """TorchSTFT(use_mel=False): (B, 513, 1 + N // hop) magnitudes sqrt(clamp(re^2 + im^2, 1e-8)) (audio.py:155-171)."""
        if x.ndim == 3:
            x = x.squeeze(1)
        return self.engine.linear(x)

    def linear_ragged(self, x, lengths):
        """Per-clip linear spectrograms of a zero-padded ragged batch: x (B, Nmax) / (B, 1, Nmax), lengths (B,) samples -> ((B, 513, 1 + Nmax // hop),
        frames (B,)) — what the reference's dataset + collate hand the posterior encoder (python/xvapitch/dataset.py:251,470-475)."""
        if x.ndim == 3:
            x = x.squeeze(1)
        return self.engine.linear_ragged(x, lengths)

    def l1_loss_backward(self, y_hat, mel_tgt, d_wav, scale=45.0, accumulate=True):
        """VitsGeneratorLoss's mel term (python/xvapitch/losses.py:187-193): loss = scale * l1_loss(mel_tgt, self(y_hat)) and
        d_wav (+)= d loss / d y_hat.  Returns (loss 1-elem tensor, mel(y_hat))."""
        if y_hat.ndim == 3:
            y_hat = y_hat.squeeze(1)
        return _l1_loss_backward(self.engine.to(y_hat.device), y_hat, mel_tgt, d_wav, scale, accumulate)


# ---- differentiable mel for the HiFi-GAN generator loss (python/hifigan/xva_train.py:480,504) ----
_lib.lib.xva_mel_backward_workspace_bytes.restype = C.c_int64
_lib.lib.xva_mel_backward_workspace_bytes.argtypes = [C.POINTER(_lib.MelConfig), C.c_int32, C.c_int32]
_lib.lib.xva_mel_l1_loss_backward.restype = C.c_int32
_lib.lib.xva_mel_l1_loss_backward.argtypes = [C.POINTER(_lib.MelConfig), C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p,
                                              C.c_int64, C.c_void_p]
_bwd_ws = {}


def mel_l1_loss_backward(y_hat, y_mel, d_wav, scale=45.0, accumulate=True, n_fft=1024, num_mels=80, sampling_rate=22050, hop_size=256,
                         win_size=1024, fmin=0, fmax=None):
    """loss = scale * F.l1_loss(y_mel, mel_spectrogram(y_hat, ..., fmax)) and d_wav (+)= d loss / d y_hat, fused on the HIP mel
    pipeline.  y_hat, d_wav: (B, N) fp32 contiguous; y_mel: (B, num_mels, N // hop).  Returns (loss 1-elem tensor, mel(y_hat))."""
    _lib.require_cuda(y_hat, y_mel, d_wav)
    key = (n_fft, num_mels, sampling_rate, hop_size, fmin, fmax, str(y_hat.device))
    eng = _hifi_engines.get(key)
    if eng is None:
        mel_basis = librosa_mel_fn(sampling_rate, n_fft, num_mels, fmin, fmax)
        eng = _MelEngine(n_fft, hop_size, num_mels, int((n_fft - hop_size) / 2), 1e-9, 0.0, torch.hann_window(win_size), mel_basis).to(y_hat.device)
        _hifi_engines[key] = eng
    return _l1_loss_backward(eng, y_hat, y_mel, d_wav, scale, accumulate)


def _l1_loss_backward(eng, y_hat, y_mel, d_wav, scale, accumulate):
    num_mels = eng.cfg.n_mel
    B, N = y_hat.shape
    y_hat = y_hat.float().contiguous()
    y_mel = y_mel.float().contiguous()
    need = int(_lib.lib.xva_mel_backward_workspace_bytes(C.byref(eng.cfg), B, N))
    if need < 0:
        raise _lib.XvaError("mel backward: " + _lib.lib.xva_last_error().decode())
    ws = _bwd_ws.get(str(y_hat.device))
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty((need + 3) // 4, device=y_hat.device, dtype=torch.float32)
        _bwd_ws[str(y_hat.device)] = ws
    T = eng.num_frames(N)
    mel_out = torch.empty(B, num_mels, T, device=y_hat.device, dtype=torch.float32)
    loss = torch.zeros(1, device=y_hat.device)
    rc = _lib.lib.xva_mel_l1_loss_backward(C.byref(eng.cfg), _lib.ptr(y_hat), B, N, y_hat.stride(0), _lib.ptr(y_mel), _lib.ptr(eng.forward_basis),
                                           _lib.ptr(eng._mel_basis_padded), float(scale), _lib.ptr(mel_out), _lib.ptr(loss), _lib.ptr(d_wav),
                                           d_wav.stride(0), int(bool(accumulate)), _lib.ptr(ws), ws.numel() * 4, _lib.stream_ptr())
    _lib.check(rc, "xva_mel_l1_loss_backward")
    return loss, mel_out
