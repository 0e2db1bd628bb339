"""Micro-benchmark of the HiFi-GAN generator resblock convolutions (stages with 128 / 64 / 32 channels, B = 64) through xva_gemm:
forward NT (lrelu on the input or not), backward-data NN (lrelu gate), weight gradient TN (lrelu on the input or not).
python tools/conv_res_bench.py [mainloop-mode]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L


def bench(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    if len(sys.argv) > 1: L.lib.xva_gemm_set_mainloop(int(sys.argv[1]))
    nseq, PAD = 64, 32
    ws = torch.zeros(96 << 18, device="cuda")
    print("%-22s %10s %10s %10s %10s %10s   (us ; TFLOP/s)" % ("shape", "fwd lrelu", "fwd plain", "bwd-data", "dW lrelu", "dW plain"))
    shapes = ((256, 256), (128, 2048), (64, 4096), (32, 8192))
    if os.environ.get("XVA_BENCH_C"):
        shapes = tuple(sh for sh in shapes if sh[0] == int(os.environ["XVA_BENCH_C"]))
    for C, T in shapes:
        Hp = T + 2 * PAD
        rows = nseq * Hp
        xs = torch.zeros(rows + 2 * PAD + 64, C, device="cuda", dtype=torch.bfloat16)
        xs[PAD:PAD + rows].view(nseq, Hp, C)[:, PAD:PAD + T] = torch.randn(nseq, T, C, device="cuda").bfloat16()
        dys = torch.zeros_like(xs)
        dys[PAD:PAD + rows].view(nseq, Hp, C)[:, PAD:PAD + T] = torch.randn(nseq, T, C, device="cuda").bfloat16()
        R = torch.randn(rows, C, device="cuda").bfloat16()
        y = torch.zeros(rows, C, device="cuda", dtype=torch.bfloat16)
        dx = torch.zeros(rows, C, device="cuda", dtype=torch.bfloat16)
        bias = torch.randn(C, device="cuda")
        for k, d in ((3, 1), (7, 1), (11, 1), (11, 5), (3, 5)):
            P_ = d * (k - 1) // 2
            Wt = (torch.randn(C, k * C, device="cuda") * 0.05).bfloat16()
            dW = torch.zeros(C, k * C, device="cuda")
            fl = 2.0 * rows * C * k * C

            def fwd(lrelu):
                L.gemm(xs, Wt, y, rows, C, k * C, C, k * C, C, layout=L.GEMM_NT, compute=1, bias=bias, R=R, ldr=C, a_lrelu=lrelu,
                       a_offset=(PAD - P_) * C, a_seglen=C, a_segadj=d * C - C, mask_mode=L.MASK_PAD, Tp=Hp, mask_pad=PAD, mask_len=T)

            def bwd():
                L.gemm(dys, Wt, dx, rows, C, k * C, C, k * C, C, layout=L.GEMM_NN, compute=1, a_offset=(PAD + P_) * C, a_seglen=C, a_segadj=-d * C - C,
                       seglen=C, seg0=0, segstride=C, G=xs[PAD:], ldg=C, gate_slope=0.1, mask_mode=L.MASK_PAD, Tp=Hp, mask_pad=PAD, mask_len=T)

            def dw(lrelu):
                if C > 64:
                    L.gemm(dys, xs, dW, C, k * C, rows, C, C, k * C, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws,
                           a_offset=PAD * C, b_offset=(PAD - P_) * C, seglen=C, seg0=0, segstride=d * C - C, b_lrelu=lrelu)
                else:
                    L.gemm(xs, dys, dW, k * C, C, rows, C, C, k * C, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws,
                           a_offset=(PAD - P_) * C, b_offset=PAD * C, a_seglen=C, a_segadj=d * C - C, a_lrelu=lrelu, c_trans=1)

            t = [bench(lambda: fwd(0.1)), bench(lambda: fwd(None)), bench(bwd), bench(lambda: dw(0.1)), bench(lambda: dw(None))]
            print("C=%3d k=%2d d=%d         " % (C, k, d) + " ".join("%5.0f;%4.0f" % (x * 1e3, fl / x / 1e9) for x in t), flush=True)


main()
