"""xva-trainer_amd — MI355X-native FastPitch1.1 + HiFi-GAN training hot path.

Host side is Python on PyTorch-ROCm (device memory, streams, torch.distributed only); every
numeric op of the path runs in libxvahip.so (hand-written HIP for gfx950, C ABI declared in
include/xva_hip.h).  There is deliberately NO CPU or eager-PyTorch fallback: importing
`_lib` raises if the shared library is missing.
"""
__version__ = "0.1.0"
