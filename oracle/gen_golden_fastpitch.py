"""FastPitch golden vectors: run the REFERENCE FastPitch / FastPitchLoss / Lamb (imported from /root/reference)
on a seeded state_dict + synthetic batch, and record inputs, outputs, losses, gradient and post-step summaries.

The 46 M-parameter state_dict is not stored: it is regenerated from `seed` by oracle.fastpitch.init_state_dict
(torch CPU generator, deterministic for the pinned torch build); the fixture stores per-tensor checksums so a
generator drift would be detected rather than silently shifting the goldens.
"""
import os

import numpy as np
import torch

from oracle import fastpitch as ofp
from oracle import golden_util as gu

# gradient tensors stored in full (reference layout) next to the evenly spaced samples of ALL of them
FULL_GRADS = [
    "encoder.word_emb.weight", "encoder.layers.0.dec_attn.qkv_net.weight", "encoder.layers.0.dec_attn.o_net.weight",
    "encoder.layers.5.dec_attn.qkv_net.weight", "encoder.layers.3.dec_attn.layer_norm.weight", "encoder.layers.3.pos_ff.layer_norm.bias",
    "decoder.layers.0.dec_attn.qkv_net.weight", "decoder.layers.5.dec_attn.o_net.weight", "decoder.layers.2.pos_ff.CoreNet.0.bias",
    "decoder.layers.2.pos_ff.CoreNet.2.bias", "pitch_emb.weight", "energy_emb.weight", "proj.weight", "proj.bias",
    "duration_predictor.layers.0.conv.bias", "duration_predictor.fc.weight", "pitch_predictor.layers.1.norm.weight", "energy_predictor.fc.weight",
]

CASES = [
    # name, stage, B, T_text, T_mel, seed
    ("fp_stage3_small", 3, 3, 14, 45, 1234),
    ("fp_stage4_small", 4, 2, 9, 30, 1235),
    ("fp_stage2_small", 2, 3, 14, 45, 1236),
]


def _ref_inputs(batch):
    B = batch["text"].size(0)
    max_inp = torch.full((B,), batch["text"].size(1), dtype=torch.long)
    max_mel = torch.full((B,), int(batch["mel_lens"].max()), dtype=torch.long)
    x = (batch["text"], batch["in_lens"], batch["mel_tgt"], batch["mel_lens"], batch["pitch"], batch["energy"], None,
         None, batch["durs"], max_inp, max_mel, None)
    y = [batch["mel_tgt"], batch["in_lens"], batch["mel_lens"], max_inp]
    return x, y


def run_reference(ns, sd, batch, stage, total_iter):
    model = ns.FastPitch()
    model.load_state_dict(sd)
    model.eval()  # dropout off
    model.training_stage = torch.tensor(stage)
    train = set(ofp.trainable_names(sd.keys(), stage))
    for n, p in model.named_parameters():
        p.requires_grad = n in train
    crit = ns.loss_function.FastPitchLoss(dur_predictor_loss_scale=0.1, pitch_predictor_loss_scale=0.1, attn_loss_scale=1.0)
    opt = ns.Lamb(model.parameters(), lr=0.1, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    x, y = _ref_inputs(batch)
    y_pred = model(x)
    loss, meta, comps = crit(y_pred, y, training_stage=stage)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
    gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1000)
    for g in opt.param_groups:
        g["lr"] = ofp.adjust_learning_rate(total_iter)
    opt.step()
    new_sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    return y_pred, float(loss), comps, grads, float(gnorm), new_sd, opt


def generate(ns, out_dir):
    for name, stage, B, Tt, Tm, seed in CASES:
        sd = ofp.init_state_dict(seed)
        batch = ofp.synth_batch(B, Tt, Tm, seed + 1)
        total_iter = 50000
        y_pred, loss, comps, grads, gnorm, new_sd, opt = run_reference(ns, {k: v.clone() for k, v in sd.items()}, batch, stage, total_iter)

        # cross-check the oracle restatement against the live reference before writing anything
        sd2 = {k: v.clone() for k, v in sd.items()}
        o_loss, o_comps, o_grads = ofp.train_step(sd2, batch, stage, {}, total_iter)
        assert abs(o_loss - loss) <= 1e-6 * max(1, abs(loss)), (o_loss, loss)
        assert set(o_grads) == set(grads), set(o_grads) ^ set(grads)
        for k in grads:
            assert torch.allclose(o_grads[k] * 1.0, grads[k], rtol=1e-4, atol=1e-7), k
        for k in new_sd:
            assert torch.allclose(sd2[k], new_sd[k], rtol=1e-5, atol=1e-7), k

        keys = sorted(grads)
        rec = {
            "seed": np.int64(seed), "stage": np.int64(stage), "total_iter": np.int64(total_iter),
            "loss": np.float64(loss), "comps": np.array([float(c) for c in comps], dtype=np.float64),
            "grad_norm": np.float64(gnorm),
            "grad_keys": np.array(keys),
            "grad_l2": np.array([float(grads[k].double().norm()) for k in keys]),
            "grad_sum": np.array([float(grads[k].double().sum()) for k in keys]),
            "param_l2_before": np.array([float(sd[k].double().norm()) for k in keys]),
            "param_l2_after": np.array([float(new_sd[k].double().norm()) for k in keys]),
            "delta_l2": np.array([float((new_sd[k].double() - sd[k].double()).norm()) for k in keys]),
            "sd_checksum": np.array([float(sd[k].double().sum()) for k in sorted(sd)]),
        }
        rec["grad_samples"], rec["grad_sample_off"] = gu.pack_samples(grads, keys)
        full = [k for k in FULL_GRADS if k in grads]
        rec["grad_full_keys"] = np.array(full)
        for i, k in enumerate(full):
            rec["grad_full_%d" % i] = grads[k].numpy()
        for k, v in batch.items():
            rec["in_" + k] = v.numpy()
        if stage == 2:
            rec["log_dur_pred"] = y_pred[3].detach().numpy()
            rec["dur_pred"] = y_pred[2].detach().numpy()
        else:
            rec["mel_out"] = y_pred[0].detach().numpy()
            rec["pitch_pred"] = y_pred[4].detach().numpy()
            rec["pitch_tgt"] = y_pred[5].detach().numpy()
            rec["energy_pred"] = y_pred[6].detach().numpy()
            rec["energy_tgt"] = y_pred[7].detach().numpy()
            # a few raw gradient slices (layout = reference layout)
            rec["g_proj_weight"] = grads["proj.weight"].numpy()
            rec["g_enc0_ffn0_w_slice"] = grads["encoder.layers.0.pos_ff.CoreNet.0.weight"][:8].numpy()
            rec["g_dec5_qkv_w_slice"] = grads["decoder.layers.5.dec_attn.qkv_net.weight"][:8].numpy()
            rec["g_word_emb"] = grads["encoder.word_emb.weight"].numpy()
            rec["new_proj_weight"] = new_sd["proj.weight"].numpy()
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **rec)
        print(name, "loss", loss, "comps", comps, "gnorm", gnorm, "ngrads", len(keys))


def generate_infer(ns, out_dir):
    """Inference path: reference FastPitch.infer in eval() on a seeded state_dict whose duration head is biased
    towards ~4 frames per token (a random-init head predicts ~0 frames)."""
    seed, B, Tt = 4321, 3, 17
    sd = ofp.init_state_dict(seed)
    sd["duration_predictor.fc.bias"] = sd["duration_predictor.fc.bias"] + 1.6
    batch = ofp.synth_batch(B, Tt, 40, seed + 1)
    text = batch["text"]
    model = ns.FastPitch()
    model.load_state_dict(sd)
    model.eval()
    with torch.no_grad():
        mel, dec_lens, dur, pitch, energy = model.infer(text, pace=1.0)
        o_mel, o_dec, o_dur, o_pitch, o_energy = ofp.infer(sd, text, 1.0)
    assert torch.equal(o_dec, dec_lens)
    for a, b in ((o_mel, mel), (o_dur, dur), (o_pitch, pitch), (o_energy, energy)):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    np.savez_compressed(os.path.join(out_dir, "fp_infer_small.npz"), seed=np.int64(seed), dur_bias_shift=np.float64(1.6), text=text.numpy(),
                        mel_out=mel.numpy(), dec_lens=dec_lens.numpy(), dur_pred=dur.numpy(), pitch_pred=pitch.numpy(), energy_pred=energy.numpy())
    print("fp_infer_small: mel", tuple(mel.shape), "dec_lens", dec_lens.tolist())


def generate_stage1(ns, out_dir):
    """Stage 1: reference FastPitch(training_stage = 1) + FastPitchLoss on a seeded batch with the reference's own
    beta-binomial prior; records attention maps, MAS durations, the CTC loss and the gradients it sends into the aligner."""
    import importlib
    seed, B, Tt, Tm = 777, 3, 12, 40
    sd = ofp.init_state_dict(seed)
    batch = ofp.synth_batch(B, Tt, Tm, seed + 1)
    from oracle import ref_import
    prior_fn, prior_src = ref_import.import_data_function().beta_binomial_prior_distribution, "reference"   # data_function.py:84-94
    prior = torch.zeros(B, int(batch["mel_lens"].max()), Tt)
    for b in range(B):
        L, M = int(batch["in_lens"][b]), int(batch["mel_lens"][b])
        prior[b, :M, :L] = prior_fn(L, M).float()
    batch["attn_prior"] = prior
    assert torch.allclose(ofp.attn_prior_batch(batch["in_lens"], batch["mel_lens"]), prior, atol=1e-7)
    model = ns.FastPitch()
    model.load_state_dict(sd)
    model.eval()
    model.training_stage = torch.tensor(1)
    # binarize_attention_parallel (model.py:283-295) ends in `.to(attn.get_device())`, which is -1 on CPU: same two lines around the
    # reference's own b_mas, minus the device move
    import types
    ref_alignment = importlib.import_module("python.fastpitch1_1.fastpitch.alignment")

    def _binarize(self, attn, in_lens, out_lens):
        with torch.no_grad():
            out = ref_alignment.b_mas(attn.data.cpu().numpy(), in_lens.cpu().numpy(), out_lens.cpu().numpy(), width=1)
            return torch.from_numpy(out)
    model.binarize_attention_parallel = types.MethodType(_binarize, model)
    crit = ns.loss_function.FastPitchLoss(dur_predictor_loss_scale=0.1, pitch_predictor_loss_scale=0.1, attn_loss_scale=1.0)
    x = (batch["text"], batch["in_lens"], batch["mel_tgt"], batch["mel_lens"], batch["pitch"], batch["energy"], None, prior, batch["durs"],
         torch.full((B,), Tt, dtype=torch.long), torch.full((B,), int(batch["mel_lens"].max()), dtype=torch.long), None)
    y = [batch["mel_tgt"], batch["in_lens"], batch["mel_lens"], x[9]]
    y_pred = model(x)
    loss, meta, _ = crit(y_pred, y, training_stage=1)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None and p.grad.abs().max() > 0}
    # the oracle restatement against the live reference
    names = [k for k in sd if k.startswith("attention.") or k == "encoder.word_emb.weight"]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    work = dict(sd); work.update(leaves)
    dur, soft, hard, logprob = ofp.forward_stage1(work, batch)
    o_loss = ofp.loss_stage1(logprob, batch)
    o_loss.backward()
    assert abs(float(o_loss) - float(loss)) < 1e-5 * abs(float(loss)), (float(o_loss), float(loss))
    assert torch.allclose(soft, y_pred[8], rtol=1e-5, atol=1e-7) and torch.equal(hard, y_pred[9]) and torch.equal(dur, y_pred[10])
    assert torch.allclose(logprob, y_pred[11], rtol=1e-5, atol=1e-6)
    for k, g in grads.items():
        assert torch.allclose(leaves[k].grad, g, rtol=1e-4, atol=1e-8), k
    keys = sorted(grads)
    rec = {"seed": np.int64(seed), "loss": np.float64(float(loss)), "attn_soft": y_pred[8].detach().numpy(), "attn_hard_dur": y_pred[10].numpy(),
           "attn_logprob": y_pred[11].detach().numpy(), "grad_keys": np.array(keys),
           "grad_l2": np.array([float(grads[k].double().norm()) for k in keys]),
           "g_key_proj2_w": grads["attention.key_proj.2.conv.weight"].numpy(), "g_query_proj4_w": grads["attention.query_proj.4.conv.weight"].numpy(),
           "g_word_emb": grads["encoder.word_emb.weight"].numpy(), "attn_prior": prior.numpy(), "prior_source": np.array(prior_src)}
    for k, v in batch.items():
        if k != "attn_prior":
            rec["in_" + k] = v.numpy()
    np.savez_compressed(os.path.join(out_dir, "fp_stage1_small.npz"), **rec)
    print("fp_stage1_small: loss", float(loss), "durs", y_pred[10].sum(1).tolist(), "grads", keys)
