import sys, os
sys.path.insert(0, "/root/repo")
import torch
from xva_trainer_amd import _lib as L
def bench(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
dt = torch.bfloat16
R = 32 * 152
x = torch.randn(R + 2, 384, device="cuda").to(dt); h = torch.randn(R + 2, 1536, device="cuda").to(dt)
W1 = torch.randn(1536, 1152, device="cuda").to(dt); W2 = torch.randn(384, 4608, device="cuda").to(dt)
o1 = torch.zeros(R, 1536, device="cuda", dtype=dt); o2 = torch.zeros(R, 384, device="cuda", dtype=dt)
cases = [
 ("enc conv2 fwd NT 4864x384x4608", lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536), 2*R*384*4608),
 ("enc conv1 bwd NN 4864x384x4608", lambda: L.gemm(h[1:], W1, o2, R, 384, 4608, 1536, 1152, 384, layout=L.GEMM_NN, compute=1, seglen=1536, seg0=2*384, segstride=-384, a_offset=-1536), 2*R*384*4608),
 ("enc conv1 fwd NT 4864x1536x1152", lambda: L.gemm(x[1:], W1, o1, R, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=-384), 2*R*1536*1152),
 ("enc conv2 bwd NN 4864x1536x1152", lambda: L.gemm(x[1:], W2, o1, R, 1536, 1152, 384, 4608, 1536, layout=L.GEMM_NN, compute=1, seglen=384, seg0=2*1536, segstride=-1536, a_offset=-384), 2*R*1536*1152),
]
modes = [-1, 1, 2, 3, 4]
print("%-34s" % "shape" + "".join("  mode %2d us/TF  " % m for m in modes))
for name, fn, fl in cases:
    row = "%-34s" % name
    for m in modes:
        L.lib.xva_gemm_set_mainloop(m); ms = bench(fn); row += "  %6.1f/%6.1f " % (ms*1e3, fl/ms/1e9)
    print(row, flush=True)
