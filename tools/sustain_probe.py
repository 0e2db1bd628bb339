import sys, os
sys.path.insert(0, "/root/repo")
import torch
from xva_trainer_amd import _lib as L
dt = torch.bfloat16
R = 32 * 862
h = torch.randn(R + 2, 1536, device="cuda").to(dt); W2 = torch.randn(384, 4608, device="cuda").to(dt); o2 = torch.zeros(R, 384, device="cuda", dtype=dt)
x = torch.randn(R + 2, 384, device="cuda").to(dt); W1 = torch.randn(1536, 1152, device="cuda").to(dt); o1 = torch.zeros(R, 1536, device="cuda", dtype=dt)
c2 = lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536)
c1 = lambda: L.gemm(x[1:], W1, o1, R, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=-384)
def run(fn, iters):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for fn, name in ((c2, "conv2 fwd 384x128"), (c1, "conv1 fwd 256x256")):
    fn(); fn()
    print(name, "x20: %.1f us  x200: %.1f  x2000: %.1f  x20 again: %.1f" % (run(fn, 20), run(fn, 200), run(fn, 2000), run(fn, 20)))
# alternate the two (each evicts part of the other's operands from L2, not from the Infinity Cache)
def both(): c1(); c2()
print("alternating pair x500: %.1f us per pair" % run(both, 500))
