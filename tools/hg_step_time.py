"""HiFi-GAN train_step time at BASELINE configs[2] (B = 64 x 8192, bf16) through HifiganStep.train_step — the trainer's own call (token reuse, side-stream zeroing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xva_trainer_amd.hifigan.step import HifiganStep
st = HifiganStep("cuda:0", "bf16")
bench.init_hifigan_weights(st)
x, y, y_mel = bench.hifigan_inputs(64, 0, "cuda:0")
for _ in range(4):
    st.train_step(x, y, y_mel)
torch.cuda.synchronize()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t0 = time.perf_counter()
for _ in range(N):
    st.train_step(x, y, y_mel)
torch.cuda.synchronize()
print("HifiganStep.train_step: %.3f ms per iteration (%d iterations)" % ((time.perf_counter() - t0) / N * 1e3, N))
