import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from oracle import fastpitch as ofp
from xva_trainer_amd import _lib
from xva_trainer_amd.fastpitch.engine import DeviceBatch
from fp_util import build_engine
sd = ofp.init_state_dict(13); batch = ofp.synth_batch(3, 41, 300, 6)
ws = {}
for name, prod in (("exact", 0), ("planes", 1)):
    _lib.lib.xva_gemm_set_fp32_products(prod)
    eng, flat, grads = build_engine(sd, "fp32")
    b = DeviceBatch.from_dict(batch, "cuda")
    grads.zero_(); eng.fwd_loss_bwd(flat, grads, b, 3); torch.cuda.synchronize()
    ws[name] = eng._ws.clone()
a, r = ws["planes"].view(torch.float32), ws["exact"].view(torch.float32)
n = a.numel() // 4096 * 4096
d = (a[:n] - r[:n]).abs().view(-1, 4096).nan_to_num(1e30).amax(1); m = r[:n].abs().view(-1, 4096).nan_to_num(0).amax(1)
bad = ((d > 1e-3 * m.clamp_min(1e-6)) & (d > 1e-6)).nonzero().flatten().tolist()
runs = []
for i in bad:
    if runs and i == runs[-1][1] + 1: runs[-1][1] = i
    else: runs.append([i, i])
for s0, s1 in runs[:60]: print("bytes [%d, %d): max diff %.3g ref max %.3g" % (s0 * 16384, (s1 + 1) * 16384, float(d[s0:s1 + 1].max()), float(m[s0:s1 + 1].max())))
