"""Stand-alone timing of the resident-operand weight-gradient kernel (csrc/wgrad_res.h) on HiFi-GAN's shapes at B = 64: the kernel
alone (slab reduction skipped), kernel + reduction, and the general TN tiles, with COLD operands (a ring of buffer sets larger than L2 +
Infinity Cache), over the plan's tuning knobs (rows per chunk, stage size, workgroups)."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L

lib = L.lib
lib.xva_gemm_set_wgrad.restype = int
SHAPES = {  # name: (items, T_in, Cin, Cout, G, k, s, d)
    "gen_c32_k11": (64, 8192, 32, 32, 1, 11, 1, 5), "gen_c32_k3": (64, 8192, 32, 32, 1, 3, 1, 1),
    "gen_c64_k11": (64, 4096, 64, 64, 1, 11, 1, 5), "gen_c64_k7": (64, 4096, 64, 64, 1, 7, 1, 3),
    "gen_c128_k11": (64, 2048, 128, 128, 1, 11, 1, 5), "gen_c128_k3": (64, 2048, 128, 128, 1, 3, 1, 1),
    "msd_conv1": (64, 8192, 128, 128, 4, 41, 2, 1), "msd_conv2": (64, 4096, 128, 256, 16, 41, 2, 1),
    "msd_conv3": (64, 2048, 256, 512, 16, 41, 4, 1), "msd_conv4": (64, 512, 512, 1024, 16, 41, 4, 1),
    "msd_conv5": (64, 128, 1024, 1024, 16, 41, 1, 1),
}
NSETS = 6


def setup(items, T_in, Cin, Cout, G, k, s, d):
    P = (k * d - d) // 2
    T_out = (T_in + 2 * P - d * (k - 1) - 1) // s + 1
    PAD = 32
    Hp = PAD + T_in + PAD
    sets = []
    for i in range(NSETS):
        xbuf = torch.zeros(items * Hp + 2 * PAD, Cin, device="cuda", dtype=torch.bfloat16)
        xbuf[PAD:PAD + items * Hp].view(items, Hp, Cin)[:, PAD:PAD + T_in] = torch.randn(items, T_in, Cin, device="cuda").bfloat16()
        dy = torch.randn(items, T_out, Cout, device="cuda").bfloat16()
        sets.append((xbuf, dy))
    Cig, Cog = Cin // G, Cout // G
    dW = torch.zeros(G, Cog, k * Cig, device="cuda")
    ws = torch.zeros(96 << 18, device="cuda")
    kw = dict(layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws, seglen=Cig, seg0=0, segstride=d * Cin - Cig, batch2=G, sA2=Cog, sB2=Cig,
              sC2=Cog * k * Cig, a_rowpitch=Cin, kb_len=T_out, kb_sA=T_out * Cout, kb_sB=Hp * Cin)

    def run(i):
        xbuf, dy = sets[i % NSETS]
        L.gemm(dy, xbuf, dW, Cog, k * Cig, items * T_out, Cout, s * Cin, k * Cig, b_offset=(PAD + PAD - P) * Cin, **kw)
    flops = 2.0 * items * T_out * Cout * k * Cig
    byts = 2.0 * (items * T_out * Cout + items * T_in * Cin)
    return run, flops, byts


def timeit(run, n=12):
    for i in range(3): run(i)
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n): run(i)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


names = sys.argv[1].split(",") if len(sys.argv) > 1 and sys.argv[1] != "all" else list(SHAPES)
sweep = len(sys.argv) > 2 and sys.argv[2] == "sweep"
for name in names:
    run, flops, byts = setup(*SHAPES[name])
    lib.xva_gemm_wgrad_tune(0, 0, 0)
    lib.xva_gemm_set_wgrad(0); t_old = timeit(run)
    lib.xva_gemm_set_wgrad(1); t_new = timeit(run)
    lib.xva_gemm_set_wgrad(2); t_k = timeit(run)
    ideal = max(flops / 2.5e15, byts / 8e12) * 1e6
    print("%-13s general %7.1f us | resident %7.1f us (kernel alone %7.1f) | ideal %5.1f us | %6.1f TFLOP/s %5.2f TB/s" % (name, t_old, t_new, t_k, ideal, flops / t_new * 1e-6, byts / t_new * 1e-6), flush=True)
    if len(sys.argv) > 2 and sys.argv[2] == "ablate":
        for ab, what in ((1, "DMA only (no LDS reads / MFMAs)"), (2, "compute only (no DMA in the loop)"), (3, "neither")):
            lib.xva_gemm_wgrad_tune(0, 0, ab << 16)
            lib.xva_gemm_set_wgrad(2); tk = timeit(run, 8)
            print("    %-36s kernel %7.1f us" % (what, tk), flush=True)
    if sweep:
        for r, niw, wgs in itertools.product((64, 128, 192, 256), (2, 4, 6), (256, 512)):
            lib.xva_gemm_wgrad_tune(r, niw, wgs)
            lib.xva_gemm_set_wgrad(2); tk = timeit(run, 8)
            lib.xva_gemm_set_wgrad(1); tn = timeit(run, 8)
            print("    R=%3d stage=%2d KiB wgs=%3d: kernel %7.1f  with reduce %7.1f" % (r, niw * 8, wgs, tk, tn), flush=True)
    lib.xva_gemm_wgrad_tune(0, 0, 0)
lib.xva_gemm_set_wgrad(1)
