"""Micro-benchmark of xva_gemm on the FastPitch decoder shapes (run on the GPU box).  Prints TFLOP/s per shape."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L

def bench(name, fn, flops, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print("%-28s %8.1f us  %7.1f TFLOP/s" % (name, ms * 1e3, flops / ms / 1e9), flush=True)

def main():
    compute = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dt = torch.bfloat16 if (len(sys.argv) > 2 and sys.argv[2] == "bf16") else torch.float32
    es = 2 if dt == torch.bfloat16 else 4
    print("compute", compute, "storage", dt, flush=True)
    R = 32 * 862
    x = torch.randn(R + 2, 384, device="cuda").to(dt); h = torch.randn(R + 2, 1536, device="cuda").to(dt)
    W1 = torch.randn(1536, 1152, device="cuda").to(dt); W2 = torch.randn(384, 4608, device="cuda").to(dt)
    o1 = torch.zeros(R, 1536, device="cuda", dtype=dt); o2 = torch.zeros(R, 384, device="cuda", dtype=dt)
    dW1 = torch.zeros(1536, 1152, device="cuda"); dW2 = torch.zeros(384, 4608, device="cuda")
    A = torch.randn(8192, 4096, device="cuda").to(dt); B = torch.randn(8192, 4096, device="cuda").to(dt); Cm = torch.zeros(8192, 8192, device="cuda", dtype=dt)
    bench("square NT 8192x8192x4096", lambda: L.gemm(A, B, Cm, 8192, 8192, 4096, 4096, 4096, 8192, compute=compute), 2 * 8192 * 8192 * 4096)
    bench("conv1 fwd NT", lambda: L.gemm(x[1:], W1, o1, R, 1536, 1152, 384, 1152, 1536, compute=compute, a_offset=-384), 2 * R * 1536 * 1152)
    bench("conv2 fwd NT", lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=compute, a_offset=-1536), 2 * R * 384 * 4608)
    bench("conv2 bwd-data NN (->1536)", lambda: L.gemm(x[1:], W2, o1, R, 1536, 1152, 384, 4608, 1536, layout=L.GEMM_NN, compute=compute, seglen=384, seg0=2*1536, segstride=-1536, a_offset=-384), 2 * R * 1536 * 1152)
    bench("conv1 bwd-data NN (->384)", lambda: L.gemm(h[1:], W1, o2, R, 384, 4608, 1536, 1152, 384, layout=L.GEMM_NN, compute=compute, seglen=1536, seg0=2*384, segstride=-384, a_offset=-1536), 2 * R * 384 * 4608)
    import ctypes as C
    def dw(dy, cout, xx, cin, out, sk):
        L.gemm(dy[1:], xx, out, cout, 3 * cin, R, cout, cin, 3 * cin, layout=L.GEMM_TN, compute=compute, accumulate=True, splitk=sk)
    for sk in (1, 7, 16):
        bench("conv1 dW TN sk=%d" % sk, lambda: dw(h, 1536, x, 384, dW1, sk), 2 * R * 1536 * 1152)
    for sk in (7, 21):
        bench("conv2 dW TN sk=%d" % sk, lambda: dw(x, 384, h, 1536, dW2, sk), 2 * R * 384 * 4608)
    qkv = torch.randn(32, 862, 192, device="cuda").to(dt); S = torch.zeros(32, 862, 864, device="cuda", dtype=dt); av = torch.zeros(32, 862, 64, device="cuda", dtype=dt)
    bench("attn QK^T batched NT", lambda: L.gemm(qkv, qkv[..., 64:], S, 862, 862, 64, 192, 192, 864, compute=compute, batch=32, sA=862*192, sB=862*192, sC=862*864), 2 * 32 * 862 * 862 * 64)
    bench("attn PV batched NN", lambda: L.gemm(S, qkv[..., 128:], av, 862, 64, 862, 864, 192, 64, layout=L.GEMM_NN, compute=compute, batch=32, sA=862*864, sB=862*192, sC=862*64), 2 * 32 * 862 * 862 * 64)

main()
