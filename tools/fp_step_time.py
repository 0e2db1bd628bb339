"""FastPitch train step time (mode = argv[2]: bf16 default, f16, fp32) at BASELINE configs[1] (B = 32 x 150 x 860, stage 3, dropout 0.1, LAMB): ms per step (the bench's headline loop without its other legs)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import synthetic
from xva_trainer_amd.fastpitch import engine as E, params as P
from xva_trainer_amd.fastpitch.lamb import Lamb
dev = torch.device("cuda")
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
eng = E.FastPitchEngine(dev, mode, p_dropout=0.1, seed=1234)
flat = torch.zeros(eng.total, device=dev); P.default_init_(flat, eng.table, seed=1234)
grads = torch.zeros_like(flat)
opt = Lamb(flat, eng.table, lr=0.1, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
active = {t[0] for t in eng.table if any(b <= t[1] < e for b, e in E.trainable_ranges(3))}
batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(32, 150, 860, 1234), dev)
it = [50000]; nskip = torch.zeros((), device=dev)
def step():
    it[0] += 1; opt.param_groups[0]["lr"] = 0.1 / it[0] ** 0.5        # the bench's schedule (fine-tune start, xva_train.py:1252-1261)
    grads.zero_(); eng.fwd_loss_bwd(flat, grads, batch, 3, grad_scale=1.0); opt.step(grads, active, max_grad_norm=1000.0, inv_scale=eng.grad_inv_scale); nskip.add_(opt.skipped)
for _ in range(5): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for _ in range(n): step()
torch.cuda.synchronize()
print("FastPitch %s step: %.3f ms (%d steps); loss %.5f grad norm %.4f loss scale %g skipped %d" % (mode, (time.perf_counter() - t0) / n * 1e3, n, eng.slot("LOSSES", (8,)).cpu()[0].item(), opt.grad_norm.item(), eng.loss_scale, int(nskip.item())))
