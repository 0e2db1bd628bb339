import sys, time, torch
sys.path.insert(0, '/root/repo')
import torch.nn.functional as F
from oracle import fastpitch as fo

class Mixed:
    name = "mixed"
    q = staticmethod(lambda x: fo._Round.apply(x, True, False))
    gq = staticmethod(lambda x: fo._Round.apply(x, False, True))
    s = staticmethod(lambda x: fo._Round.apply(x, True, True))

# patched layer functions: residual stream exact, operands rounded
def _mha(sd, pre, inp, key_pad_mask, drop=None, site=0, st=None, resid32=True):
    op = st.q(inp) if resid32 else inp
    qkv = st.s(F.linear(op, st.q(sd[pre + "qkv_net.weight"]), sd[pre + "qkv_net.bias"]))
    q, k, v = torch.chunk(qkv, 3, dim=2)
    score = torch.bmm(q, k.transpose(1, 2)) * (1 / (fo.D_HEAD ** 0.5))
    score = score.masked_fill(key_pad_mask.unsqueeze(1), -float("inf"))
    vec = st.s(fo._flash_pv(score, v, None, st))
    out = F.linear(vec, st.q(sd[pre + "o_net.weight"]))
    s = inp + out
    if not resid32: s = st.s(s)
    y = F.layer_norm(s, (fo.D_MODEL,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"])
    return y if resid32 else st.s(y)

def _conv_ff(sd, pre, inp, drop=None, site=0, st=None, resid32=True):
    core = (st.q(inp) if resid32 else inp).transpose(1, 2)
    core = F.conv1d(core, st.q(sd[pre + "CoreNet.0.weight"]), sd[pre + "CoreNet.0.bias"], padding=1)
    core = st.s(F.relu(core))
    core = F.conv1d(core, st.q(sd[pre + "CoreNet.2.weight"]), sd[pre + "CoreNet.2.bias"], padding=1)
    core = core.transpose(1, 2)
    s = inp + core
    if not resid32: s = st.s(s)
    y = F.layer_norm(s, (fo.D_MODEL,), sd[pre + "layer_norm.weight"], sd[pre + "layer_norm.bias"])
    return y if resid32 else st.s(y)

def run(sd, batch, mode, in32=True, out32=True):
    st = Mixed
    resid32 = mode == "mixed"
    def fft(pre, dec_inp, seq_lens=None, embed=False):
        if embed:
            inp = F.embedding(dec_inp, sd[pre + "word_emb.weight"], padding_idx=0); mask = (dec_inp != 0).unsqueeze(2)
        else:
            inp = dec_inp; mask = fo.mask_from_lens(seq_lens, inp.size(1)).unsqueeze(2)
        pos = fo.positional_embedding(inp.size(1), fo.D_MODEL, inp.dtype) * mask
        out = inp + pos
        if not (resid32 and in32): out = st.s(out)
        for i in range(fo.N_LAYERS):
            lp = "%slayers.%d." % (pre, i)
            out = _mha(sd, lp + "dec_attn.", out, ~mask.squeeze(2), st=st, resid32=resid32) * mask
            out = _conv_ff(sd, lp + "pos_ff.", out, st=st, resid32=resid32) * mask
        return out, mask
    text, mel_lens = batch["text"], batch["mel_lens"]
    enc_out, enc_mask = fft("encoder.", text, embed=True)
    dur_tgt = batch["durs"]
    pitch_tgt = fo.average_pitch(batch["pitch"], dur_tgt)
    pitch_emb = F.conv1d(pitch_tgt, sd["pitch_emb.weight"], sd["pitch_emb.bias"], padding=1)
    enc_out = enc_out + pitch_emb.transpose(1, 2)
    if not (resid32 and in32): enc_out = st.s(enc_out)
    energy_tgt = torch.log(1.0 + fo.average_pitch(batch["energy"].unsqueeze(1), dur_tgt))
    energy_emb = F.conv1d(energy_tgt, sd["energy_emb.weight"], sd["energy_emb.bias"], padding=1)
    enc_out = enc_out + energy_emb.transpose(1, 2)
    if not (resid32 and in32): enc_out = st.s(enc_out)
    lr, dec_lens = fo.regulate_len(dur_tgt, enc_out, 1.0, int(mel_lens.max()))
    dec_out, dec_mask = fft("decoder.", lr, seq_lens=dec_lens)
    mel = F.linear(st.q(dec_out) if resid32 else dec_out, st.q(sd["proj.weight"]), sd["proj.bias"])
    if not out32: mel = st.s(mel)
    return mel

torch.manual_seed(0)
B, Tt, Tm = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sd = fo.init_state_dict(1)
batch = fo.synth_batch(B, Tt, Tm, 3)
with torch.no_grad():
    ref = fo.forward({k: v.double() for k, v in sd.items()}, {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}, 3)[0].float()
    r32 = fo.forward(sd, batch, 3)[0]
    rb = fo.forward(sd, batch, 3, storage="bf16")[0]
    def err(a, name):
        d = (a - ref)
        print("%-28s rel-L2 %.2e  max-abs %.2e (ref max %.2f, rms %.3f)" % (name, d.norm() / ref.norm(), d.abs().max(), ref.abs().max(), ref.pow(2).mean().sqrt()))
    err(r32, "fp32"); err(rb, "bf16 storage (engine today)")
    err(run(sd, batch, "bf16", out32=False), "bf16 (my restatement)")
    err(run(sd, batch, "bf16", out32=True), "bf16 + fp32 mel_out")
    err(run(sd, batch, "mixed", in32=False, out32=True), "mixed, bf16 stack inputs")
    err(run(sd, batch, "mixed", in32=True, out32=True), "mixed, fp32 everywhere resid")
