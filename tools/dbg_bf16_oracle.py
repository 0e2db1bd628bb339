import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from oracle import fastpitch as ofp
from fp_util import build_engine, grad_report
from xva_trainer_amd.fastpitch.engine import DeviceBatch
from xva_trainer_amd.fastpitch import params as P

def l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm()).item()
def mx(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max()).item()

def step(sd, batch, stage, storage):
    names = ofp.trainable_names(sd.keys(), stage)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd); work.update(leaves)
    out = ofp.forward(work, batch, stage, storage=storage)
    loss, comps = ofp.loss(out, batch, stage)
    loss.backward()
    return out, loss, {k: v.grad for k, v in leaves.items() if v.grad is not None}

for stage in (3, 2):
    sd = ofp.init_state_dict(77)
    batch = ofp.synth_batch(4, 37, 210, 78)
    eng, flat, grads = build_engine(sd, "bf16")
    b = DeviceBatch.from_dict(batch, "cuda")
    losses = eng.fwd_loss_bwd(flat, grads, b, stage).cpu()
    out = eng.outputs(b, stage)
    mine = P.from_flat(grads, eng.table)
    for storage in ("bf16", "fp32"):
        o, loss, g = step(sd, batch, stage, storage)
        if stage == 2:
            print("stage", stage, storage, "log_dur l2 %.2e max %.2e" % (l2(out["log_dur_pred"], o[3]), mx(out["log_dur_pred"], o[3])), "loss rel %.2e" % (abs(losses[0].item() - loss.item()) / abs(loss.item())))
        else:
            print("stage", stage, storage, "mel l2 %.2e max %.2e | pitch l2 %.2e max %.2e | energy l2 %.2e" % (l2(out["mel_out"].float(), o[0]), mx(out["mel_out"].float(), o[0]),
                  l2(out["pitch_pred"], o[4]), mx(out["pitch_pred"], o[4]), l2(out["energy_pred"], o[6])), "loss rel %.2e" % (abs(losses[0].item() - loss.item()) / abs(loss.item())))
        errs = sorted(((l2(mine[k], g[k]), k) for k in g), reverse=True)
        print("   grads: worst", ["%s %.2e" % (k, e) for e, k in errs[:6]], "median %.2e" % errs[len(errs) // 2][0])
        a = torch.cat([mine[k].double().cpu().flatten() for k in g]); r = torch.cat([g[k].double().flatten() for k in g])
        print("   whole-gradient l2 %.2e  cos-1 %.2e" % (((a - r).norm() / r.norm()).item(), 1 - (a @ r / (a.norm() * r.norm())).item()))
