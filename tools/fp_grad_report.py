"""Diagnostic: per-tensor relative gradient error of the HIP FastPitch engine vs the CPU oracle (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import fastpitch as ofp
from fp_util import build_engine
from xva_trainer_amd.fastpitch import params as P
from xva_trainer_amd.fastpitch.engine import DeviceBatch

stage = int(sys.argv[1]) if len(sys.argv) > 1 else 3
compute = sys.argv[2] if len(sys.argv) > 2 else "bf16"
if len(sys.argv) > 3 and "," in sys.argv[3]:          # B,Tt,Tm
    B_, Tt_, Tm_ = (int(v) for v in sys.argv[3].split(","))
    sd = ofp.init_state_dict(5)
    batch = ofp.synth_batch(B_, Tt_, Tm_, 21)
elif len(sys.argv) > 3 and sys.argv[3] == "full":       # BASELINE configs[1]'s sequence lengths
    sd = ofp.init_state_dict(5)
    batch = ofp.synth_batch(2, 150, 860, 21)
else:
    sd = ofp.init_state_dict(77)
    batch = ofp.synth_batch(4, 37, 210, 78)
names = ofp.trainable_names(sd.keys(), stage)
leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
work = dict(sd); work.update(leaves)
out_ref = ofp.forward(work, batch, stage)
loss_ref, comps = ofp.loss(out_ref, batch, stage)
loss_ref.backward()
eng, flat, grads = build_engine(sd, compute)
b = DeviceBatch.from_dict(batch, "cuda")
eng.fwd_loss_bwd(flat, grads, b, stage)
mine = P.from_flat(grads, eng.table)
rows = []
for k, v in leaves.items():
    if v.grad is None:
        continue
    g = v.grad.double()
    r = ((mine[k].double().cpu() - g).norm() / g.norm().clamp_min(1e-30)).item()
    rows.append((r, k, g.norm().item()))
rows.sort(reverse=True)
for r, k, n in rows[:40]:
    print("%.5f  %-60s |g|=%.3e" % (r, k, n))
import fp_util
print("mel rel err", fp_util.rel(eng.outputs(b, stage)["mel_out"], out_ref[0]))
