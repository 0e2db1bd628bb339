"""Stand-alone FastPitch decoder GEMMs with the caches flushed before every launch (what they cost inside the step) and warm."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L
dt = torch.bfloat16
R = 32 * 862
x = torch.randn(R + 2, 384, device="cuda").to(dt); h = torch.randn(R + 2, 1536, device="cuda").to(dt)
W1 = torch.randn(1536, 1152, device="cuda").to(dt); W2 = torch.randn(384, 4608, device="cuda").to(dt)
o1 = torch.zeros(R, 1536, device="cuda", dtype=dt); o2 = torch.zeros(R, 384, device="cuda", dtype=dt)
big = torch.zeros(512 << 20, device="cuda", dtype=torch.uint8)
def bench(fn, cold, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        if cold: big.add_(1)     # flush L2 and the Infinity Cache (512 MB read + written)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    return sum(ts) / len(ts) * 1e3
c1 = lambda: L.gemm(x[1:], W1, o1, R, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=-384)
c2 = lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536)
print("conv1 fwd (256x256): cold %.1f us warm %.1f us   conv2 fwd (384x128): cold %.1f us warm %.1f us" % (bench(c1, True), bench(c1, False), bench(c2, True), bench(c2, False)))
