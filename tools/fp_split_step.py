"""FastPitch full train step in the fp32 mode with split-bf16 products (the configuration that meets north_star's 1e-3 on outputs and losses), B = 32 x 150 x 860:
ms per step; run under `rocprofv3 --kernel-trace --stats` for the per-kernel table.  XVA_FP_FFN_PLANES=0 gives the round-4 form (every product splits while staging)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib, synthetic
from xva_trainer_amd.fastpitch import engine as E, params as P
from xva_trainer_amd.fastpitch.lamb import Lamb
dev = torch.device("cuda")
_lib.lib.xva_gemm_set_fp32_products(1)
if os.environ.get("XVA_SERIAL"): _lib.lib.xva_fp_set_streams(1)
eng = E.FastPitchEngine(dev, "fp32", p_dropout=0.1, seed=1234)
flat = torch.zeros(eng.total, device=dev); P.default_init_(flat, eng.table, seed=1234)
grads = torch.zeros_like(flat)
opt = Lamb(flat, eng.table, lr=0.1, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
active = {t[0] for t in eng.table if any(b <= t[1] < e for b, e in E.trainable_ranges(3))}
batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(32, 150, 860, 1234), dev)
def step():
    grads.zero_(); eng.fwd_loss_bwd(flat, grads, batch, 3, grad_scale=1.0); opt.step(grads, active, max_grad_norm=1000.0)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(os.environ.get("XVA_STEPS", "8"))
for _ in range(n): step()
torch.cuda.synchronize()
print("split-products step: %.2f ms (FFN planes %s)" % ((time.perf_counter() - t0) / n * 1e3, os.environ.get("XVA_FP_FFN_PLANES", "1")))
