"""Kernel times of the mel front end on FastPitch's bench batch (32 clips x 219 904 samples -> 27 520 frames), per front-end mode
(xva_mel_set_dft: 0 fused kernel, 2 four-launch FFT pipeline, 1 dense DFT).  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel table;
alone it prints the wall time per call of a back-to-back loop (host issue included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib
from xva_trainer_amd.mel import TacotronSTFT
_lib.lib.xva_mel_set_dft.restype = int
st = TacotronSTFT().cuda()
wav = torch.rand(32, 219904, device="cuda") * 1.6 - 0.8
for mode in (0, 2, 1):
    _lib.lib.xva_mel_set_dft(mode)
    for _ in range(3): st.mel_spectrogram(wav)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): st.mel_spectrogram(wav)
    torch.cuda.synchronize()
    print("mode %d: %.1f us per call (wall, 20 calls back to back)" % (mode, (time.perf_counter() - t0) / 20 * 1e6))
_lib.lib.xva_mel_set_dft(0)
