"""Where the HiFi-GAN discriminator phases spend their time: each phase (D forward, D-step backward, G-step backward) timed with the
stream lanes ON for all eight discriminators, the period discriminators alone, the scale discriminators alone, and every discriminator
alone (xva_hg_set_disc_mask: a profiling knob — masked-out networks are skipped, so only the times mean anything)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from xva_trainer_amd import _lib
from xva_trainer_amd.hifigan.step import HifiganStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
st = HifiganStep("cuda:0", "bf16")
bench.init_hifigan_weights(st)
x, y, y_mel = bench.hifigan_inputs(B, 0, "cuda:0")
eng = st.eng
yg = eng.generator_forward(st.flat_g, x)

def timed(fn, n=6):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

masks = [("all", 0xff), ("mpd", 0x1f), ("msd", 0xe0)] + [("d%d" % i, 1 << i) for i in range(8)]
print("%-5s %9s %9s %9s" % ("mask", "fwd", "bwd_d", "bwd_g"))
for name, m in masks:
    _lib.lib.xva_hg_set_disc_mask(m)
    eng.disc_forward(st.flat_d, y, yg)
    f = timed(lambda: eng.disc_forward(st.flat_d, y, yg))
    bd = timed(lambda: eng.disc_backward_d(st.flat_d, st.grads_d))
    bg = timed(lambda: eng.disc_backward_g(st.flat_d))
    print("%-5s %9.3f %9.3f %9.3f" % (name, f, bd, bg))
_lib.lib.xva_hg_set_disc_mask(0xff)
