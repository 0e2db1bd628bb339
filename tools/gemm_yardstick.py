"""Same-box yardstick for the direct-to-LDS GEMM main loop: the vendor's bf16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) against xva_gemm on the
shapes that dominate the FastPitch step, on N(0, 1) data and on zeros (the matrix pipes' power draw depends on operand toggling), warm (operands possibly in the
256 MB Infinity Cache from the previous launch) and after a 256 MB eviction write.  A TOOL: nothing in the package or in a timed region imports torch.matmul.

usage: python tools/gemm_yardstick.py [iters]   -> table on stdout (committed as profiles/r06_gemm_yardstick.txt)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
evict = torch.empty(256 << 20, dtype=torch.uint8, device=dev)

def timed(fn, cold):
    ts = []
    for i in range(iters + 3):
        if cold:
            evict.fill_(i & 255)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        if i >= 3:
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]          # median, us (includes ~5 us of event cost on both sides)

SHAPES = [("conv1 fwd      NT", "NT", 27584, 1536, 1152), ("conv2 fwd      NT", "NT", 27584, 384, 4608), ("conv2 bwd-data NT", "NT", 27584, 1536, 1152),
          ("conv1 wgrad    TN", "TN", 1536, 1152, 27584), ("conv2 wgrad    TN", "TN", 384, 4608, 27584), ("conv1 bwd-data NN", "NN", 27584, 384, 4608),
          ("square         NT", "NT", 8192, 8192, 4096),
          ("hifigan p11 fwd NT", "NT", 16896, 1024, 5120), ("hifigan p11 bwd NN", "NN", 16896, 1024, 5120), ("hifigan p7 fwd  NT", "NT", 15232, 1024, 5120)]
print("%-20s %7s %6s %6s | %-6s %-5s | %10s %10s | %10s %10s | %s" % ("product", "M", "N", "K", "data", "cache", "vendor us", "TFLOP/s", "xva us", "TFLOP/s", "xva / vendor time"))
for name, lay, M, N, K in SHAPES:
    for data in ("N(0,1)", "zeros"):
        mk = (lambda *s: torch.randn(*s, device=dev).bfloat16()) if data == "N(0,1)" else (lambda *s: torch.zeros(*s, device=dev, dtype=torch.bfloat16))
        if lay == "NT":
            A, B = mk(M, K), mk(N, K); ven = lambda: torch.matmul(A, B.t()); lda, ldb = K, K
        elif lay == "NN":
            A, B = mk(M, K), mk(K, N); ven = lambda: torch.matmul(A, B); lda, ldb = K, N
        else:
            A, B = mk(K, M), mk(K, N); ven = lambda: torch.matmul(A.t(), B); lda, ldb = M, N
        Cx = torch.zeros(M, N, device=dev, dtype=torch.float32 if lay == "TN" else torch.bfloat16)
        ws = torch.empty(64 << 20, device=dev, dtype=torch.uint8)
        layout = {"NT": L.GEMM_NT, "NN": L.GEMM_NN, "TN": L.GEMM_TN}[lay]
        if lay == "TN":      # the weight-gradient form: fp32 accumulate through split-K slabs, as the engine calls it
            C0 = torch.zeros(M, N, device=dev)
            xva = lambda: L.gemm(A, B, C0, M, N, K, lda, ldb, N, layout=layout, compute=1, accumulate=True, splitk=0, sk_ws=ws)
            venf = lambda: torch.matmul(A.t(), B)
        else:
            xva = lambda: L.gemm(A, B, Cx, M, N, K, lda, ldb, N, layout=layout, compute=1)
            venf = ven
        if data == "N(0,1)":      # same product
            ref = venf().float(); xva(); got = (C0 if lay == "TN" else Cx).float()
            err = ((got - ref).abs().max() / ref.abs().max()).item()
            assert err < 2e-2, (name, err)
        for cold in (False, True):
            tv, tx = timed(venf, cold), timed(xva, cold)
            fl = 2.0 * M * N * K
            print("%-20s %7d %6d %6d | %-6s %-5s | %10.1f %10.0f | %10.1f %10.0f | %.3f" % (name, M, N, K, data, "cold" if cold else "warm", tv, fl / tv / 1e6, tx, fl / tx / 1e6, tx / tv))
        del A, B
