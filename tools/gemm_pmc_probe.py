"""Launch the FastPitch decoder conv products on the 256x256 tile with both K loops (for rocprofv3 --pmc: the two K loops are
different kernels, the three layouts different instantiations).  python tools/gemm_pmc_probe.py [launches]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dt = torch.bfloat16
R = 32 * 862
x = torch.randn(R + 2, 384, device="cuda").to(dt); h = torch.randn(R + 2, 1536, device="cuda").to(dt)
W1 = torch.randn(1536, 1152, device="cuda").to(dt); W2 = torch.randn(384, 4608, device="cuda").to(dt)
o1 = torch.zeros(R, 1536, device="cuda", dtype=dt)
dW1 = torch.zeros(1536, 1152, device="cuda"); ws = torch.zeros(8 * 1536 * 1152, device="cuda")
cases = [
    lambda: L.gemm(x[1:], W1, o1, R, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=-384),                                             # conv1 fwd NT
    lambda: L.gemm(x[1:], W2, o1, R, 1536, 1152, 384, 4608, 1536, layout=L.GEMM_NN, compute=1, seglen=384, seg0=2 * 1536, segstride=-1536, a_offset=-384),  # conv2 bwd-data NN
    lambda: L.gemm(h[1:], x, dW1, 1536, 1152, R, 1536, 384, 1152, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws),     # conv1 dW TN
]
L.lib.xva_gemm_set_mainloop(2)
for kl in (0, 1):
    L.lib.xva_gemm_set_kloop(kl)
    for fn in cases:
        for _ in range(n):
            fn()
torch.cuda.synchronize()
