"""Time of one clip + LAMB step over FastPitch's parameters (44.8 M active in stage 3): python tools/lamb_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd.fastpitch import engine as E, params as P
from xva_trainer_amd.fastpitch.lamb import Lamb

eng = E.FastPitchEngine("cuda:0", "bf16")
flat = torch.zeros(eng.total, device="cuda"); P.default_init_(flat, eng.table, seed=1)
grads = torch.randn_like(flat) * 1e-3
opt = Lamb(flat, eng.table, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
ranges = E.trainable_ranges(3)
active = {t[0] for t in eng.table if any(b <= t[1] < e for b, e in ranges)}
for _ in range(3):
    opt.step(grads, active, max_grad_norm=1000.0)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    opt.step(grads, active, max_grad_norm=1000.0)
e1.record(); torch.cuda.synchronize()

print("clip + LAMB step: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
