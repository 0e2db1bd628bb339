"""A/B of the direct-to-LDS GEMM tiles on the FastPitch decoder shapes: python tools/gemm_tile_ab.py [mode ...]
(modes of xva_gemm_set_mainloop: -1 auto, 2 = 256x256, 1 = 128x128, 7 = 384x128 for NT / NN)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


def main():
    # "2:0" = mainloop mode 2 with K loop 0 (xva_gemm_set_kloop); a bare "2" keeps the default K loop (1)
    modes = [tuple(int(v) for v in (a.split(":") + ["1"])[:2]) for a in sys.argv[1:]] or [(2, 0), (2, 1), (7, 1)]
    dt = torch.bfloat16
    R = 32 * 862
    x = torch.randn(R + 2, 384, device="cuda").to(dt); h = torch.randn(R + 2, 1536, device="cuda").to(dt)
    W1 = torch.randn(1536, 1152, device="cuda").to(dt); W2 = torch.randn(384, 4608, device="cuda").to(dt)
    o1 = torch.zeros(R, 1536, device="cuda", dtype=dt); o2 = torch.zeros(R, 384, device="cuda", dtype=dt)
    dW1 = torch.zeros(1536, 1152, device="cuda"); dW2 = torch.zeros(384, 4608, device="cuda")
    ws = torch.zeros(8 * 1536 * 1152, device="cuda")
    b1 = torch.randn(1536, device="cuda"); b2 = torch.randn(384, device="cuda"); r2 = torch.randn(R, 384, device="cuda").to(dt)
    lens = torch.full((32,), 860, device="cuda", dtype=torch.int32)
    A = torch.randn(8192, 4096, device="cuda").to(dt); B = torch.randn(8192, 4096, device="cuda").to(dt); Cm = torch.zeros(8192, 8192, device="cuda", dtype=dt)
    A0 = torch.zeros_like(A); B0 = torch.zeros_like(B)
    cases = [
        ("square NT 8192^2x4096", lambda: L.gemm(A, B, Cm, 8192, 8192, 4096, 4096, 4096, 8192, compute=1), 2 * 8192 * 8192 * 4096),
        ("square NT zeros", lambda: L.gemm(A0, B0, Cm, 8192, 8192, 4096, 4096, 4096, 8192, compute=1), 2 * 8192 * 8192 * 4096),
        ("square NN", lambda: L.gemm(A, B, Cm, 8192, 8192, 4096, 4096, 8192, 8192, layout=L.GEMM_NN, compute=1), 2 * 8192 * 8192 * 4096),
        ("square TN", lambda: L.gemm(A, B, Cm, 8192, 8192, 4096, 8192, 8192, 8192, layout=L.GEMM_TN, compute=1), 2 * 8192 * 8192 * 4096),
        ("conv1 fwd NT", lambda: L.gemm(x[1:], W1, o1, R, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=-384), 2 * R * 1536 * 1152),
        ("conv1 fwd +bias+relu", lambda: L.gemm(x[1:], W1, o1, R, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=-384, bias=b1, relu=True), 2 * R * 1536 * 1152),
        ("conv1 fwd +bias+relu+mask", lambda: L.gemm(x[1:], W1, o1, R, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=-384, bias=b1, relu=True,
                                                  mask_mode=L.MASK_LEN, lens=lens, Tp=862, mask_pad=1, mask_len=860), 2 * R * 1536 * 1152),
        ("conv2 fwd +bias+R+mask", lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536, bias=b2, R=r2, ldr=384,
                                               mask_mode=L.MASK_LEN, lens=lens, Tp=862, mask_pad=1, mask_len=860), 2 * R * 384 * 4608),
        ("conv2 fwd +bias+drop+R+mask", lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536, bias=b2, R=r2, ldr=384,
                                                    mask_mode=L.MASK_LEN, lens=lens, Tp=862, mask_pad=1, mask_len=860, drop_p=0.1, drop_seed=5, drop_stream=3), 2 * R * 384 * 4608),
        ("conv2 bwd-data NN +gate+mask", lambda: L.gemm(x[1:], W2, o1, R, 1536, 1152, 384, 4608, 1536, layout=L.GEMM_NN, compute=1, seglen=384, seg0=2 * 1536, segstride=-1536, a_offset=-384,
                                                     G=h, ldg=1536, mask_mode=L.MASK_LEN, lens=lens, Tp=862, mask_pad=1, mask_len=860), 2 * R * 1536 * 1152),
        ("conv2 fwd NT", lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536), 2 * R * 384 * 4608),
        ("conv2 bwd-data NN", lambda: L.gemm(x[1:], W2, o1, R, 1536, 1152, 384, 4608, 1536, layout=L.GEMM_NN, compute=1, seglen=384, seg0=2 * 1536, segstride=-1536, a_offset=-384), 2 * R * 1536 * 1152),
        ("conv1 bwd-data NN", lambda: L.gemm(h[1:], W1, o2, R, 384, 4608, 1536, 1152, 384, layout=L.GEMM_NN, compute=1, seglen=1536, seg0=2 * 384, segstride=-384, a_offset=-1536), 2 * R * 384 * 4608),
        ("conv1 dW TN auto-sk", lambda: L.gemm(h[1:], x, dW1, 1536, 1152, R, 1536, 384, 1152, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws), 2 * R * 1536 * 1152),
        ("conv2 dW TN auto-sk", lambda: L.gemm(x[1:], h, dW2, 384, 4608, R, 384, 1536, 4608, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws), 2 * R * 384 * 4608),
    ]
    print("%-24s" % "shape" + "".join("  mode %2d:%d us / TF " % m for m in modes))
    for name, fn, fl in cases:
        row = "%-24s" % name
        for m in modes:
            L.lib.xva_gemm_set_mainloop(m[0]); L.lib.xva_gemm_set_kloop(m[1])
            ms = bench(fn)
            row += "  %8.1f / %6.1f " % (ms * 1e3, fl / ms / 1e9)
        print(row, flush=True)
    L.lib.xva_gemm_set_mainloop(-1); L.lib.xva_gemm_set_kloop(1)


main()
