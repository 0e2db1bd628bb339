"""HiFi-GAN generator: which storage / operand format keeps the WAVEFORM within 1e-3 of the fp32 reference?  (HiFi-GAN half.)
CPU only, on the oracle's generator graph (oracle/hifigan.py, pinned to the reference Generator), fp64 yardstick, 8192-sample segments.
  bf16 / f16        : every stored activation (residual stream x, its LeakyReLU copy, the ResBlock intermediate) and every effective weight rounded (the engine's throughput mode / its fp16 twin)
  bf16_r32 / f16_r32: the residual stream x and the ResBlock sums fp32; rounded only as convolution OPERANDS (activated copy, intermediate, weights)
Two weight sets: the seeded initialisation (N(0, 0.01) ResBlock / upsampling weights as the reference's init_weights) and one with larger gains (weight_g x 1.25: the 78-layer stack amplifies geometrically, x 3 saturates the tanh)."""
import sys, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from oracle import hifigan as oh
torch.set_num_threads(8)

def gen(sd, x, q, r):
    """q: operand / intermediate rounding; r: residual-stream rounding"""
    w = lambda pre: q(oh.wn_weight(sd, pre))
    x = r(F.conv1d(x, oh.wn_weight(sd, "conv_pre."), sd["conv_pre.bias"], padding=3))        # the 80-channel boundary conv runs on fp32-stored mel in the engine
    nk = len(oh.RES_KERNELS)
    for i, (u, k) in enumerate(zip(oh.UPSAMPLE_RATES, oh.UPSAMPLE_KERNELS)):
        x = r(F.conv_transpose1d(q(F.leaky_relu(x, oh.LRELU_SLOPE)), w("ups.%d." % i), sd["ups.%d.bias" % i], stride=u, padding=(k - u) // 2))
        xs = None
        for j in range(nk):
            pre, kk, dil = "resblocks.%d." % (i * nk + j), oh.RES_KERNELS[j], oh.RES_DILATIONS[j]
            y = x
            for m in range(3):
                xt = q(F.leaky_relu(F.conv1d(q(F.leaky_relu(y, oh.LRELU_SLOPE)), w("%sconvs1.%d." % (pre, m)), sd["%sconvs1.%d.bias" % (pre, m)], padding=oh.get_padding(kk, dil[m]), dilation=dil[m]), oh.LRELU_SLOPE))
                y = r(F.conv1d(xt, w("%sconvs2.%d." % (pre, m)), sd["%sconvs2.%d.bias" % (pre, m)], padding=oh.get_padding(kk, 1)) + y)
            xs = y if xs is None else xs + y
        x = r(xs / nk)
    x = q(F.leaky_relu(x))
    return torch.tanh(F.conv1d(x, w("conv_post."), sd["conv_post.bias"], padding=3))

ident = lambda t: t
rd = lambda dt: (lambda t: t.to(dt).to(t.dtype))
MODES = [("fp32", ident, ident), ("bf16", rd(torch.bfloat16), rd(torch.bfloat16)), ("f16", rd(torch.float16), rd(torch.float16)),
         ("bf16_r32", rd(torch.bfloat16), ident), ("f16_r32", rd(torch.float16), ident)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd0 = oh.init_generator_sd(1234)
x_mel, y_wav = oh.synth_batch(B, 5)[:2]
for name, sd in (("seeded init", sd0), ("larger gains (weight_g x 1.25)", {k: (v * 1.25 if k.endswith("weight_g") and "conv_p" not in k else v) for k, v in sd0.items()})):
    with torch.no_grad():
        ref = gen({k: v.double() for k, v in sd.items()}, x_mel.double(), ident, ident)
        print("== %s: waveform rms %.4f max %.3f" % (name, ref.pow(2).mean().sqrt(), ref.abs().max()))
        for m, q, r in MODES:
            y = gen(sd, x_mel, q, r)
            d = y.double() - ref
            print("%-10s wave rel-L2 %.2e  max-abs/max %.2e" % (m, d.norm() / ref.norm(), d.abs().max() / ref.abs().max()))
