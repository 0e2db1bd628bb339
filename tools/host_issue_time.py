"""Is the FastPitch / HiFi-GAN step bound by the host issuing launches?  Times how long the CPU takes to ISSUE n steps (no sync) against
the time until the GPU has finished them.  python tools/host_issue_time.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench


class A: pass


a = A(); a.compute = "bf16"; a.seed = 1234
dev = torch.device("cuda:0")
from xva_trainer_amd.fastpitch import engine as E, params as P
from xva_trainer_amd.fastpitch.lamb import Lamb
from oracle import fastpitch as ofp   # tools only: random state_dict
eng = E.FastPitchEngine(dev, "bf16", p_dropout=0.1, seed=1)
flat = torch.zeros(eng.total, device=dev)
P.to_flat(ofp.init_state_dict(1), eng.table, flat)
grads = torch.zeros_like(flat)
batch = E.DeviceBatch.from_dict(ofp.synth_batch(32, 150, 860, 2), dev)
opt = Lamb(flat, eng.table, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
active = set(ofp.trainable_names([n for n, *_ in eng.table], 3))


def step():
    grads.zero_()
    eng.fwd_loss_bwd(flat, grads, batch, 3)
    opt.step(grads, active, max_grad_norm=1000.0)


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("FastPitch: host issue %.2f ms/step, GPU done after %.2f ms/step" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
