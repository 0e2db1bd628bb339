"""A/B of the tile choice on FastPitch's short-reduction GEMMs (attention projections): python tools/small_gemm_ab.py"""
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
R = 32 * 862
x = torch.randn(R, 384, device="cuda").to(dt); a64 = torch.randn(R, 64, device="cuda").to(dt); q = torch.randn(R, 192, device="cuda").to(dt)
Wo = torch.randn(384, 64, device="cuda").to(dt); Wq = torch.randn(192, 384, device="cuda").to(dt)
o384 = torch.zeros(R, 384, device="cuda", dtype=dt); o192 = torch.zeros(R, 192, device="cuda", dtype=dt); o64 = torch.zeros(R, 64, device="cuda", dtype=dt)
bq = torch.randn(192, device="cuda"); lens = torch.full((32,), 860, device="cuda", dtype=torch.int32)
dWq = torch.zeros(192, 384, device="cuda"); dWo = torch.zeros(384, 64, device="cuda"); ws = torch.zeros(64 * 384 * 192, device="cuda")
mk = dict(mask_mode=L.MASK_LEN, lens=lens, Tp=862, mask_pad=1, mask_len=860)
cases = [
 ("o_net fwd NT 27584x384x64 +drop+R", lambda: L.gemm(a64, Wo, o384, R, 384, 64, 64, 64, 384, compute=1, R=x, ldr=384, drop_p=0.1, drop_seed=3, drop_stream=1, **mk)),
 ("qkv fwd NT 27584x192x384 +bias", lambda: L.gemm(x, Wq, o192, R, 192, 384, 384, 384, 192, compute=1, bias=bq)),
 ("qkv bwd-data NN 27584x384x192", lambda: L.gemm(q, Wq, o384, R, 384, 192, 192, 384, 384, layout=L.GEMM_NN, compute=1, **mk)),
 ("o_net bwd-data NN 27584x64x384", lambda: L.gemm(x, Wo, o64, R, 64, 384, 384, 64, 64, layout=L.GEMM_NN, compute=1)),
 ("qkv dW TN 192x384x27584", lambda: L.gemm(q, x, dWq, 192, 384, R, 192, 384, 384, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws)),
 ("o_net dW TN 384x64x27584", lambda: L.gemm(x, a64, dWo, 384, 64, R, 384, 64, 64, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws)),
]
modes = [-1, 1, 2, 3, 4, 5, 8, 0]
print("%-36s" % "shape" + "".join("  mode %2d us " % m for m in modes))
for name, fn in cases:
    row = "%-36s" % name
    for m in modes:
        L.lib.xva_gemm_set_mainloop(m); row += "  %9.1f " % (bench(fn) * 1e3)
    print(row, flush=True)
