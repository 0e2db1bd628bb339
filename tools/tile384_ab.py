"""A/B of the 384x128 tile's K loops (xva_gemm_set_kloop384: 0 lock-step, 1 staggered wave groups) on the FastPitch shapes that take it:
python tools/tile384_ab.py"""
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
    dt = torch.bfloat16
    lens = torch.full((32,), 860, device="cuda", dtype=torch.int32)
    print("%-44s %s" % ("shape", "lock-step us / TF    staggered us / TF"))
    for R in (32 * 862, 32 * 152):
        x = torch.randn(R + 2, 384, device="cuda").to(dt); h = torch.randn(R + 2, 1536, device="cuda").to(dt)
        W1 = torch.randn(1536, 1152, device="cuda").to(dt); W2 = torch.randn(384, 4608, device="cuda").to(dt)
        o2 = torch.zeros(R, 384, device="cuda", dtype=dt)
        b2 = torch.randn(384, device="cuda"); r2 = torch.randn(R, 384, device="cuda").to(dt)
        Tp = R // 32
        cases = [
            ("conv2 fwd NT %dx384x4608" % R, lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536)),
            ("conv2 fwd +bias+drop+R+mask", lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536, bias=b2, R=r2, ldr=384,
                                                        mask_mode=L.MASK_LEN, lens=lens, Tp=Tp, mask_pad=1, mask_len=Tp - 2, drop_p=0.1, drop_seed=5, drop_stream=3)),
            ("conv1 bwd-data NN %dx384x4608" % R, lambda: L.gemm(h[1:], W1, o2, R, 384, 4608, 1536, 1152, 384, layout=L.GEMM_NN, compute=1, seglen=1536, seg0=2 * 384,
                                                                segstride=-384, a_offset=-1536)),
        ]
        fl = 2 * R * 384 * 4608
        for name, fn in cases:
            row = "%-44s" % name
            outs = []
            for m in (0, 1):
                L.lib.xva_gemm_set_mainloop(7); L.lib.xva_gemm_set_kloop384(m)
                o2.zero_(); fn(); outs.append(o2.float().clone())
                ms = bench(fn)
                row += "  %8.1f / %6.1f " % (ms * 1e3, fl / ms / 1e9)
            d = (outs[0] - outs[1]).abs().max().item() / outs[0].abs().max().item()
            print(row + "  rel diff %.2g" % d, flush=True)
    L.lib.xva_gemm_set_mainloop(-1); L.lib.xva_gemm_set_kloop384(1)


main()
