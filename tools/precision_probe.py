"""Which reduced-precision storage / operand format brings FastPitch's OUTPUTS within 1e-3 of the fp32 reference?  

CPU only, on the oracle's restatement of the reference graph (oracle/fastpitch.py, itself pinned to runs of the reference classes), at the
benchmark's depth and length: 6 + 6 transformer layers, T_text 150, T_mel 860.  The yardstick is the SAME graph in fp64; each candidate
puts its roundings where the HIP engine would (oracle/fastpitch.py: make_storage):

  bf16          activations and their gradients stored bf16, bf16 weight shadow               (the throughput mode today)
  f16           the same in IEEE half (11 mantissa bits; the reference's own GPU width under autocast: fastpitch1_1/xva_train.py:350,787)
  bf16_r32      residual stream / LayerNorm I/O fp32, bf16 only as MFMA operand and for qkv / attention output / FFN intermediate
  f16_r32       the same with fp16 operands
  2pass_act     activations as hi + lo bf16 pairs, weights ONE bf16 plane (two MFMA passes)   = operand error of the weights alone
  planes        both operands as hi + lo bf16 pairs, three passes                             (the `fastpitch_split` mode today)

Two weight sets: the seeded default initialisation, and a "trained-looking" scale (projection / FFN weights x 2.5, LayerNorm gains U(0.5, 1.5),
biases N(0, 0.1)) whose pre-LayerNorm sums and attention logits are several times larger.

Also reported for the 16-bit formats: the gradient picture in fp16 — the share of stored activation-gradient elements that would flush to
zero (< 2^-24) or go subnormal (< 2^-14) at loss scale 1 and at the scale that puts the largest element at 2^14, i.e. whether a GradScaler-like
loss scale is needed (the reference runs one: xva_train.py:856-859).

usage: python tools/precision_probe.py [B Tt Tm]   -> table on stdout (committed as profiles/r06_precision_probe.txt)
"""
import sys, time, math, torch
sys.path.insert(0, '/root/repo')
from oracle import fastpitch as fo

torch.set_num_threads(8)


class _Split2(torch.autograd.Function):
    """x -> hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits, value and gradient."""
    @staticmethod
    def forward(ctx, x):
        hi = x.to(torch.bfloat16).to(x.dtype)
        return hi + (x - hi).to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        hi = g.to(torch.bfloat16).to(g.dtype)
        return hi + (g - hi).to(torch.bfloat16).to(g.dtype)


def _mk(name, act, wq):
    ident = staticmethod(lambda x: x)
    return type(name, (), dict(name=name, flash=True, resid32=False, s=staticmethod(act), q=staticmethod(wq), gq=staticmethod(act),
                               r=staticmethod(act), o=ident))


TwoPassAct = _mk("2pass_act", _Split2.apply, lambda x: fo._Round.apply(x, True, False, torch.bfloat16))
Planes = _mk("planes", _Split2.apply, lambda x: _Split2.apply(x))

MODES = [("fp32", None), ("bf16", "bf16"), ("f16", "f16"), ("bf16_r32", "bf16_r32"), ("f16_r32", "f16_r32"),
         ("2pass_act", TwoPassAct), ("planes", Planes)]


def trained_looking(sd, seed=7):
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in sd.items():
        if k.endswith("layer_norm.weight") or k.endswith("norm.weight"):
            out[k] = (0.5 + torch.rand(v.shape, generator=g)).to(v.dtype)
        elif k.endswith(".bias"):
            out[k] = (0.1 * torch.randn(v.shape, generator=g)).to(v.dtype)
        elif v.dim() >= 2 and "word_emb" not in k:
            out[k] = v * 2.5
        else:
            out[k] = v.clone()
    return out


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


def run(sd, batch, storage, grads=False):
    work = {k: v.clone().requires_grad_(grads and v.is_floating_point() and v.dim() >= 1 and "inv_freq" not in k and "pitch_" + "mean" not in k
                                         and "pitch_std" not in k) for k, v in sd.items()}
    taps = {} if grads else None
    if grads:
        out = fo.forward(work, batch, 3, taps=taps, storage=storage)
        for t in taps.values():
            if t.requires_grad:
                t.retain_grad()
        l = fo.loss(out, batch, 3)[0]
        l.backward()
        return out, l, work, taps
    with torch.no_grad():
        out = fo.forward(work, batch, 3, storage=storage)
        l = fo.loss(out, batch, 3)[0]
    return out, l, None, None


def main():
    B, Tt, Tm = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (2, 150, 860)
    sd0 = fo.init_state_dict(1)
    batch = fo.synth_batch(B, Tt, Tm, 3)
    print("FastPitch stage 3, B = %d, T_text = %d, T_mel = %d, %d + %d layers; error = rel-L2 against the same graph in fp64" % (B, Tt, Tm, fo.N_LAYERS, fo.N_LAYERS))
    for wname, sd in (("seeded default init", sd0), ("trained-looking scale", trained_looking(sd0))):
        sd64 = {k: v.double() for k, v in sd.items()}
        b64 = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
        ref, lref, w64, taps64 = run(sd64, b64, None, grads=True)
        gkeys = ["decoder.layers.5.pos_ff.CoreNet.0.weight", "decoder.layers.0.dec_attn.qkv_net.weight", "encoder.layers.0.pos_ff.CoreNet.2.weight",
                 "encoder.word_emb.weight", "pitch_predictor.layers.0.conv.weight"]
        print("\n== weights: %s   (mel_out rms %.3f, max |.| %.2f; loss %.6f)" % (wname, float(ref[0].pow(2).mean().sqrt()), float(ref[0].abs().max()), float(lref)))
        print("%-10s %9s %9s %9s %9s | %s | %s" % ("mode", "mel", "pitch", "energy", "loss", "gradient rel-L2 (5 tensors: dec5.ffn1, dec0.qkv, enc0.ffn2, word_emb, pitch.conv0)", "s"))
        for name, st in MODES:
            t0 = time.time()
            out, l, w, taps = run(sd, batch, st, grads=True)
            e = [rel(out[0], ref[0]), rel(out[4], ref[4]), rel(out[6], ref[6]), abs(float(l) - float(lref)) / abs(float(lref))]
            ge = [rel(w[k].grad, w64[k].grad) for k in gkeys]
            print("%-10s %9.2e %9.2e %9.2e %9.2e | %s | %.0f" % (name, e[0], e[1], e[2], e[3], " ".join("%9.2e" % x for x in ge), time.time() - t0))
            sys.stdout.flush()
        # the fp16 gradient range, from the fp64 run's activation gradients (what an fp16 buffer would have to hold)
        allg = torch.cat([t.grad.flatten().abs() for t in taps64.values() if t.grad is not None])
        nz = allg[allg > 0]
        mx = float(nz.max())
        for scale_name, scale in (("loss scale 1", 1.0), ("loss scale 2^%d (max element -> 2^14)" % int(14 - math.ceil(math.log2(mx))), 2.0 ** (14 - math.ceil(math.log2(mx))))):
            v = nz * scale
            print("  stored activation gradients (%d tensors, %d non-zero el.): |g| max %.2e median %.2e ; %s: %.2f %% below 2^-24 (flush), %.2f %% below 2^-14 (subnormal)"
                  % (len(taps64), nz.numel(), mx, float(nz.median()), scale_name, 100 * float((v < 2.0 ** -24).double().mean()), 100 * float((v < 2.0 ** -14).double().mean())))


if __name__ == "__main__":
    main()
