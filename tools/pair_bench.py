"""Generator forward at BASELINE configs[2] (B = 64 x 8192 samples) with the ResBlock pairs of the 32 / 64-channel stages as two launches (mode 0) or one
(conv_pair.hip; modes 1 / 2 / 3: see hifigan_engine.hip gen_forward).    python tools/pair_bench.py <mode> [iterations]
Prints the HIP-event time per forward; under rocprofv3 (tools/pair_ab.sh) gives the per-kernel times and the PMC traffic of the same calls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xva_trainer_amd import _lib
from xva_trainer_amd.hifigan.step import HifiganStep

mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
st = HifiganStep("cuda:0", "bf16")
bench.init_hifigan_weights(st)
x, y, y_mel = bench.hifigan_inputs(64, 0, "cuda:0")
_lib.lib.xva_hg_set_pair_mode(mode)
for _ in range(3):
    st.eng.generator_forward(st.flat_g, x)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    st.eng.generator_forward(st.flat_g, x)
e1.record(); torch.cuda.synchronize()
print("XVA_HG_PAIR=%d: generator forward %.3f ms (B = 64, %d calls)" % (mode, e0.elapsed_time(e1) / N, N))
