"""GPU parity of training stage 1 — the aligner (ConvAttention, monotonic alignment search, forward-sum / CTC loss and its backward into
attention.* and encoder.word_emb) against vectors recorded from the REFERENCE classes (tests/golden/fp_stage1_small.npz) and against the
CPU oracle on a larger ragged batch."""
import os

import numpy as np
import pytest
import torch

from fp_util import build_engine, rel

pytestmark = pytest.mark.gpu


def _run(eng, flat, batch, grad_scale=1.0):
    grads = torch.zeros_like(flat)
    loss, durs, soft, logp = eng.align_forward(flat, batch["text"].cuda(), batch["in_lens"], batch["mel_tgt"].cuda(), batch["mel_lens"],
                                               batch["attn_prior"].cuda())
    eng.align_backward(flat, grads, grad_scale)
    torch.cuda.synchronize()
    return loss.cpu(), durs.cpu(), soft.cpu(), logp.cpu(), grads


@pytest.mark.parametrize("compute,tol", [("fp32", 1e-3), ("bf16", 3e-2)])
def test_stage1_against_reference_golden(golden_dir, compute, tol):
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import params as P
    g = np.load(os.path.join(golden_dir, "fp_stage1_small.npz"))
    sd = ofp.init_state_dict(int(g["seed"]))
    batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    batch["attn_prior"] = torch.from_numpy(g["attn_prior"])
    eng, flat, _ = build_engine(sd, compute)
    loss, durs, soft, logp, grads = _run(eng, flat, batch)
    assert abs(loss.item() - float(g["loss"])) < tol * float(g["loss"])
    assert rel(soft, torch.from_numpy(g["attn_soft"])) < tol
    lp_ref = torch.from_numpy(g["attn_logprob"])
    valid = torch.isfinite(lp_ref) & (lp_ref > -30)          # log(0 + 1e-8) on padded prior cells is compared separately
    assert (logp[valid] - lp_ref[valid]).abs().max().item() < (2e-3 if compute == "fp32" else 5e-2)
    if compute == "fp32":
        assert torch.equal(durs.long(), torch.from_numpy(g["attn_hard_dur"]).long())
    else:
        assert torch.equal(durs.sum(1).long(), torch.from_numpy(g["attn_hard_dur"]).sum(1).long())
    mine = P.from_flat(grads, eng.table)
    for k, l2 in zip(g["grad_keys"], g["grad_l2"]):
        k = str(k)
        assert abs(mine[k].double().norm().item() - l2) < (2e-3 if compute == "fp32" else 5e-2) * l2, k
    for name, key in (("attention.key_proj.2.conv.weight", "g_key_proj2_w"), ("attention.query_proj.4.conv.weight", "g_query_proj4_w"),
                      ("encoder.word_emb.weight", "g_word_emb")):
        r = ((mine[name].double().cpu() - torch.from_numpy(g[key]).double()).norm() / torch.from_numpy(g[key]).double().norm()).item()
        assert r < (2e-3 if compute == "fp32" else 5e-2), (name, r)
    # nothing else receives a gradient in stage 1
    for name in mine:
        if not (name.startswith("attention.") or name == "encoder.word_emb.weight"):
            assert mine[name].abs().max().item() == 0.0, name


def test_stage1_against_oracle_ragged_and_grad_scale():
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import params as P
    sd = ofp.init_state_dict(99)
    batch = ofp.synth_batch(4, 41, 230, 100)
    batch["attn_prior"] = ofp.attn_prior_batch(batch["in_lens"], batch["mel_lens"])
    names = [k for k in sd if k.startswith("attention.") or k == "encoder.word_emb.weight"]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd); work.update(leaves)
    dur, soft_ref, hard, logprob = ofp.forward_stage1(work, batch)
    loss_ref = ofp.loss_stage1(logprob, batch)
    (loss_ref * 0.25).backward()
    eng, flat, _ = build_engine(sd, "fp32")
    loss, durs, soft, logp, grads = _run(eng, flat, batch, grad_scale=0.25)
    assert abs(loss.item() - loss_ref.item()) < 1e-3 * loss_ref.item()
    assert rel(soft, soft_ref) < 1e-3
    mism = (durs.long() != dur.long()).sum().item()            # a tie in the DP can flip with the last bit of log(): never more than a token or two
    assert mism <= 2 and torch.equal(durs.sum(1).long(), batch["mel_lens"])
    mine = P.from_flat(grads, eng.table)
    for k, v in leaves.items():
        if v.grad is None or v.grad.abs().max() == 0:
            continue
        r = ((mine[k].double().cpu() - v.grad.double()).norm() / v.grad.double().norm()).item()
        assert r < 2e-3, (k, r)


def test_stage1_through_the_reference_interface(golden_dir):
    """FastPitch(training_stage = 1)(inputs_x) + FastPitchLoss(..., training_stage=1) + backward, as xva_train.py drives them."""
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import params as P
    from xva_trainer_amd.fastpitch.loss_function import FastPitchLoss
    from xva_trainer_amd.fastpitch.model import FastPitch
    g = np.load(os.path.join(golden_dir, "fp_stage1_small.npz"))
    sd = ofp.init_state_dict(int(g["seed"]))
    batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
    prior = torch.from_numpy(g["attn_prior"])
    model = FastPitch(compute="fp32").cuda()
    model.load_state_dict(sd)
    model.training_stage = torch.tensor(1)
    B, Tt = batch["text"].shape
    x = (batch["text"].cuda(), batch["in_lens"].cuda(), batch["mel_tgt"].cuda(), batch["mel_lens"].cuda(), batch["pitch"].cuda(), batch["energy"].cuda(),
         None, prior.cuda(), batch["durs"].cuda(), torch.full((B,), Tt), torch.full((B,), int(batch["mel_lens"].max())), None)
    y = [batch["mel_tgt"].cuda(), batch["in_lens"].cuda(), batch["mel_lens"].cuda(), x[9]]
    crit = FastPitchLoss(dur_predictor_loss_scale=0.1, pitch_predictor_loss_scale=0.1, attn_loss_scale=1.0)
    y_pred = model(x)
    loss, meta, comps = crit(y_pred, y, training_stage=model.training_stage)
    (loss / 4).backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-3 * float(g["loss"]) and comps == [None, None, None, None]
    assert torch.equal(y_pred[10].cpu().long(), torch.from_numpy(g["attn_hard_dur"]).long())
    assert rel(y_pred[8], torch.from_numpy(g["attn_soft"])) < 1e-3
    mine = P.from_flat(model.flat.grad, model._table)
    for k, l2 in zip(g["grad_keys"], g["grad_l2"]):
        assert abs(mine[str(k)].double().norm().item() - l2 / 4) < 2e-3 * l2 / 4, k
