"""fp16-operand FastPitch mode at the benchmark size: is one forward + backward finite, and how does the gradient depend on the loss scale?
argv: nothing.  Prints, per loss scale, finiteness of outputs / gradients, the gradient norm, and the relative distance to the fp32 engine's gradient."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import synthetic
from xva_trainer_amd.fastpitch import engine as E, params as P
dev = torch.device("cuda")
batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(32, 150, 860, 1234), dev)
def run(mode, scale=None, drop=0.0):
    eng = E.FastPitchEngine(dev, mode, p_dropout=drop, seed=1234)
    if scale is not None: eng.set_loss_scale(scale)
    flat = torch.zeros(eng.total, device=dev); P.default_init_(flat, eng.table, seed=1234)
    grads = torch.zeros_like(flat)
    losses = eng.fwd_loss_bwd(flat, grads, batch, 3).cpu()
    o = eng.outputs(batch, 3)
    g = eng.unscaled(grads).clone()
    return eng, losses, {k: v.float().clone() for k, v in o.items() if v.is_floating_point()}, g
from xva_trainer_amd import _lib
_lib.lib.xva_gemm_set_fp32_products(1)
e32, l32, o32, g32 = run("fp32")
print("fp32 (split products): loss %.6f grad norm %.5f" % (l32[0], g32.norm()))
for sc in (None, 2.0 ** 21, 2.0 ** 17, 2.0 ** 13, 2.0 ** 9, 1.0):
    eng, l, o, g = run("f16", sc)
    fin_o = all(bool(torch.isfinite(v).all()) for v in o.values())
    nbad = int((~torch.isfinite(g)).sum())
    rel = float((torch.nan_to_num(g) - g32).norm() / g32.norm())
    mel = float((o["mel_out"] - o32["mel_out"]).abs().max() / o32["mel_out"].abs().max())
    print("f16 scale %-10g: loss %.6f outputs finite %s mel_rel %.2e | grads non-finite %d, norm %.5f, rel to fp32 %.3e" % (eng.loss_scale, l[0], fin_o, mel, nbad, torch.nan_to_num(g).norm(), rel))
    if nbad:
        tbl = P.from_flat(g, eng.table)
        print("   non-finite tensors:", [k for k, v in tbl.items() if not bool(torch.isfinite(v).all())][:12])
