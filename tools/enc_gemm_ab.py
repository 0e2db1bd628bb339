"""A/B of the tile choice on FastPitch's ENCODER feed-forward GEMMs (4 864 rows = 32 x 152 tokens): python tools/enc_gemm_ab.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L
def bench(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
dt = torch.bfloat16
R = 32 * 152
x = torch.randn(R, 384, device="cuda").to(dt); h = torch.randn(R, 1536, device="cuda").to(dt)
W1 = torch.randn(1536, 1152, device="cuda").to(dt); W2 = torch.randn(384, 4608, device="cuda").to(dt)
W1b = torch.randn(1152, 1536, device="cuda").to(dt); W2b = torch.randn(4608, 384, device="cuda").to(dt)
x3 = torch.randn(R, 1152, device="cuda").to(dt); h3 = torch.randn(R, 4608, device="cuda").to(dt)
o1536 = torch.zeros(R, 1536, device="cuda", dtype=dt); o384 = torch.zeros(R, 384, device="cuda", dtype=dt)
cases = [
 ("conv1 fwd NT 4864x1536x1152", 2 * R * 1536 * 1152, lambda: L.gemm(x3, W1, o1536, R, 1536, 1152, 1152, 1152, 1536, compute=1)),
 ("conv2 fwd NT 4864x384x4608", 2 * R * 384 * 4608, lambda: L.gemm(h3, W2, o384, R, 384, 4608, 4608, 4608, 384, compute=1)),
 ("conv2 bwd NN 4864x1536x1152", 2 * R * 1536 * 1152, lambda: L.gemm(x3, W1b, o1536, R, 1536, 1152, 1152, 1536, 1536, layout=L.GEMM_NN, compute=1)),
 ("conv1 bwd NN 4864x384x4608", 2 * R * 384 * 4608, lambda: L.gemm(h3, W2b, o384, R, 384, 4608, 4608, 384, 384, layout=L.GEMM_NN, compute=1)),
]
modes = [-1, 1, 2, 3, 4, 5, 6, 7, 8]
print("%-32s" % "shape" + "".join("  mode %2d us " % m for m in modes))
for name, fl, fn in cases:
    row = "%-32s" % name
    for m in modes:
        L.lib.xva_gemm_set_mainloop(m)
        try:
            row += "  %9.1f " % (bench(fn) * 1e3)
        except Exception as e:
            row += "  %9s " % "err"
    print(row, flush=True)
L.lib.xva_gemm_set_mainloop(-1)
