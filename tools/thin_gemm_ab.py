"""The K <= 192 products of the FastPitch attention block (qkv / o_net, forward and backward-data) under each direct-to-LDS tile: they are bandwidth-shaped
(27 584 rows x 384 columns of fp32 or 16-bit residual + output against a 64 - 192-deep reduction), so the epilogue and the tile's launch geometry are all there is."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L
R = 32 * 862
big = torch.zeros(512 << 20, device="cuda", dtype=torch.uint8)
L.lib.xva_gemm_set_mainloop.restype = int


def bench(fn, cold=True, iters=10):
    ts = []
    for i in range(iters + 2):
        if cold: big.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    return sum(ts) / len(ts)


def case(name, layout, M, N, K, dt, r32, c32):
    A = torch.randn(M, K, device="cuda").to(dt)
    B = (torch.randn(N, K, device="cuda") if layout == L.GEMM_NT else torch.randn(K, N, device="cuda")).to(dt)
    Rr = torch.randn(M, N, device="cuda").to(torch.float32 if r32 else dt)
    Cc = torch.zeros(M, N, device="cuda", dtype=torch.float32 if c32 else dt)
    ldb = K if layout == L.GEMM_NT else N
    mb = (A.numel() * 2 + B.numel() * 2 + Rr.numel() * Rr.element_size() + Cc.numel() * Cc.element_size()) / 1e6
    out = []
    for ml, nm in ((-1, "auto"), (4, "64x64"), (3, "128x64"), (1, "128x128"), (2, "256x256")):
        old = L.lib.xva_gemm_set_mainloop(ml)
        try:
            fn = lambda: L.gemm(A, B, Cc, M, N, K, K, ldb, N, layout=layout, compute=1, R=Rr, ldr=N)
            try:
                fn(); out.append("%s %.1f" % (nm, bench(fn)))
            except L.XvaError:
                out.append("%s -" % nm)
        finally:
            L.lib.xva_gemm_set_mainloop(old)
    print("%-34s %6.1f MB (%.1f us at 5 TB/s) | cold us: %s" % (name, mb, mb / 5.0, "  ".join(out)))


h, b = torch.float16, torch.bfloat16
case("f16 qkv bwd-data NN K=192 r32 c32", L.GEMM_NN, R, 384, 192, h, True, True)
case("f16 o_net bwd-data NN N=64 K=384", L.GEMM_NN, R, 64, 384, h, False, False)
case("f16 qkv fwd NT N=192 K=384", L.GEMM_NT, R, 192, 384, h, False, False)
case("bf16 qkv bwd-data NN K=192", L.GEMM_NN, R, 384, 192, b, False, False)
case("bf16 qkv fwd NT N=192 K=384", L.GEMM_NT, R, 192, 384, b, False, False)
case("bf16 o_net bwd-data NN N=64 K=384", L.GEMM_NN, R, 64, 384, b, False, False)
