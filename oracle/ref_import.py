"""Import the reference's Python modules in THIS container (never on the GPU box).

The reference (read-only at /root/reference) needs `librosa` and `numba`, which are absent
here and cannot be installed.  Following SURVEY.md §8c we pre-seed sys.modules with:
  librosa  — util.pad_center / util.tiny / util.normalize and filters.mel.  filters.mel is
             our restatement of librosa 0.8.1's Slaney filterbank (oracle/mel.py) — this is
             the one piece of the oracle that is parity-UNPINNED by the reference.
  numba    — identity `jit` (plain Python semantics of the MAS loops) and prange=range.
  parselmouth, soundfile, tensorboard — empty placeholders for import-time only.
Used only by oracle/gen_golden.py and tests that are skipped when /root/reference is absent.
"""
import os
import sys
import types

import numpy as np

REF_ROOT = os.environ.get("XVA_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REF_ROOT, "python", "fastpitch1_1"))


def _install_stubs():
    from oracle import mel as omel

    if "librosa" not in sys.modules:
        librosa = types.ModuleType("librosa")
        util = types.ModuleType("librosa.util")
        filters = types.ModuleType("librosa.filters")

        def pad_center(data, size, axis=-1, **kwargs):
            n = data.shape[axis]
            lpad = int((size - n) // 2)
            lengths = [(0, 0)] * data.ndim
            lengths[axis] = (lpad, int(size - n - lpad))
            return np.pad(data, lengths, **kwargs)

        def tiny(x):
            x = np.asarray(x)
            dtype = x.dtype if np.issubdtype(x.dtype, np.floating) else np.float32
            return np.finfo(dtype).tiny

        def normalize(S, **kwargs):
            return omel.peak_normalize(S)

        def mel_fn(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **kwargs):
            return omel.slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax)

        util.pad_center, util.tiny, util.normalize = pad_center, tiny, normalize
        filters.mel = mel_fn
        librosa.util, librosa.filters = util, filters
        librosa.load = None
        sys.modules["librosa"] = librosa
        sys.modules["librosa.util"] = util
        sys.modules["librosa.filters"] = filters

    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")

        def jit(*args, **kwargs):
            if len(args) == 1 and callable(args[0]) and not kwargs:
                return args[0]
            return lambda f: f

        numba.jit = jit
        numba.njit = jit
        numba.prange = range
        sys.modules["numba"] = numba

    for name in ("parselmouth", "soundfile", "tensorboard", "torch.utils.tensorboard"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = types.ModuleType(name)
                if name.endswith("tensorboard"):
                    m.SummaryWriter = object
                sys.modules[name] = m


def import_data_function():
    """The reference's fastpitch/data_function.py (TTSDataset, TTSCollate, batch_to_gpu, beta_binomial_prior_distribution): its text
    front end additionally imports unidecode / inflect, absent from this image; neither is reached by the functions the generators call."""
    import importlib
    _install_stubs()
    for name in ("unidecode", "inflect"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            if name == "unidecode":
                m.unidecode = lambda s: s
            else:
                m.engine = type("engine", (), {"number_to_words": lambda self, *a, **k: ""})
            sys.modules[name] = m
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    return importlib.import_module("python.fastpitch1_1.fastpitch.data_function")


def import_reference():
    """Returns a namespace of the reference classes/functions the oracle is pinned against."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_stubs()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    ns = types.SimpleNamespace()
    from python.fastpitch1_1.fastpitch.model import FastPitch, regulate_len, average_pitch
    from python.fastpitch1_1.fastpitch import loss_function as lf
    from python.fastpitch1_1.common.layers import TacotronSTFT
    from python.fastpitch1_1.lamb import Lamb
    from python.hifigan import models as hmodels
    from python.hifigan import meldataset as hmel
    ns.FastPitch, ns.regulate_len, ns.average_pitch = FastPitch, regulate_len, average_pitch
    ns.loss_function = lf
    ns.TacotronSTFT = TacotronSTFT
    ns.Lamb = Lamb
    ns.hifigan_models = hmodels
    ns.hifigan_meldataset = hmel

    # FastPitchLoss hard-codes torch.device('cuda:N') (loss_function.py:92-129): proxy torch in that module so
    # device() resolves to CPU.
    import torch

    class _TorchProxy:
        def __getattr__(self, k):
            return getattr(torch, k)

        @staticmethod
        def device(*a, **k):
            return torch.device("cpu")

    lf.torch = _TorchProxy()
    return ns
