"""GPU, world size 2, through the reference's entry point: ONE `handleTrainer(models_manager, data, websocket, gpus=[a, b])` call in ONE process
(what server.py does, server.py:171,211-227) trains FastPitch / HiFi-GAN / xVAPitch on two rank workers (xva-trainer_amd/dp_launch.py).

For each trainer, from a synthetic dataset directory in the reference's layout: two (or more) epochs; training.log / checkpoints written once (rank 0
only — a second writer would double the lines); resume from the checkpoint the run left; the ranks' stopping decisions identical (a stage transition
is reached and passed by both); and the parameters after N optimiser steps equal to a SINGLE-process run whose batch is the union of the two ranks'
batches (FastPitch: global loss normalisation; HiFi-GAN and xVAPitch: means over equally sized shards), within 1e-4.

Two launch modes, as in tests/test_dp2_gpu.py: `nccl` (one rank per GPU over RCCL; needs >= 2 GPUs, skipped otherwise) and `gloo` (both ranks on cuda:0 —
the same product code: rank workers, events, side streams, bucket exchanges; only the transport differs).
Replaces the reference's nn.DataParallel (python/fastpitch1_1/xva_train.py:409,451-452,465-466,954-970; python/xvapitch/xva_train.py:427-428)."""
import asyncio
import json
import logging
import os
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _WS:
    def __init__(self):
        self.sent = []

    async def send(self, msg):
        self.sent.append(msg)


def _mm():
    from xva_trainer_amd.models_manager import ModelsManager
    return ModelsManager(logging.getLogger("t"), False, "cuda:0")


def _backends():
    return [pytest.param("nccl", marks=pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs 2 GPUs (RCCL refuses two ranks on one device)")),
            pytest.param("gloo")]


@pytest.fixture()
def dp_env(monkeypatch):
    def use(backend):
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            monkeypatch.delenv(k, raising=False)
        if backend == "gloo":
            monkeypatch.setenv("XVA_DP_BACKEND", "gloo")
            return [0, 0]                       # two ranks share the box's one device
        monkeypatch.delenv("XVA_DP_BACKEND", raising=False)
        return [0, 1]
    return use


def _run(handle, mm, data, ws, gpus, resume=False):
    return asyncio.run(handle(mm, data, ws, gpus, resume))


def _end_group(mm, key):
    g = mm.models_bank.pop(key, None)
    if g is not None and hasattr(g, "close"):
        g.close()


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


# ------------------------------------------------------------------------------------------------ FastPitch
@pytest.mark.timeout(1800)
@pytest.mark.parametrize("backend", _backends())
def test_fastpitch_trainer_world2(tmp_path, dp_env, backend):
    from xva_trainer_amd import data as D
    from xva_trainer_amd.dp_launch import RankGroup
    from xva_trainer_amd.fastpitch import xva_train as T
    gpus = dp_env(backend)
    ds = D.write_synthetic_dataset(str(tmp_path / "in" / "voice_d"), n_items=8, seed=6, min_s=0.5, max_s=1.0)
    opts = {"compute": "fp32", "p_dropout": 0.0}
    base = {"dataset_path": ds, "checkpoint": None, "num_workers": 0, "epochs_per_checkpoint": 1, "force_stage": 3, "trainer_options": opts}
    # ---- two ranks, base batch 1: global batch = int(1 * 3.5 * 2 GPUs * 10 / longest clip) capped to the 8 x 4 repeated items = 16 per rank, one batch
    # an epoch; three optimiser steps = two finished epochs (the roll-over happens when the third batch is asked for)
    out2 = str(tmp_path / "out2")
    mm, ws = _mm(), _WS()
    assert _run(T.handleTrainer, mm, dict(base, output_path=out2, batch_size=1, max_iterations=50003), ws, gpus) is None
    group = mm.models_bank["fastpitch1_1"]
    assert isinstance(group, RankGroup) and group.parked and group.world == 2                 # like the single-process trainer: stopped, resumable
    assert any(m.startswith("Set stage to: 3") for m in ws.sent) and sum(m.startswith("Set stage to") for m in ws.sent) == 1     # rank 0's line, once
    vo = out2 + "/voice_d"
    log = open(vo + "/training.log").read()
    assert "GPUs mult: 2" in log and "Extracting durations from alignments" in log and log.count("New voice") == 1
    steps = re.findall(r"Stage: 3 \| Epoch: \d+ \| iter: .*? -> (\d+) \|", log)
    assert steps == ["50001", "50002", "50003"], steps                                          # one line per optimiser step: rank 0 alone writes
    assert len(os.listdir(ds + "/durs_text")) == 8
    cks = sorted(f for f in os.listdir(vo) if f.startswith("FastPitch_checkpoint_"))
    assert cks == ["FastPitch_checkpoint_2_50001.pt", "FastPitch_checkpoint_3_50002.pt"], cks
    assert not [f for f in os.listdir(vo) if ".tmp." in f]
    # ---- resume in place (server.py "resume"): the same workers continue
    assert _run(T.handleTrainer, mm, dict(base, output_path=out2, batch_size=1, max_iterations=50004), ws, gpus, resume=True) is None
    assert re.findall(r"-> (\d+) \|", open(vo + "/training.log").read())[-1] == "50004"
    _end_group(mm, "fastpitch1_1")
    # ---- single process, base batch 2: the same 32 items per batch (the union of the two ranks' strides of the same shuffled epoch)
    out1 = str(tmp_path / "out1")
    mm1 = _mm()
    assert _run(T.handleTrainer, mm1, dict(base, output_path=out1, batch_size=2, max_iterations=50003), _WS(), [0]) is None
    tr = mm1.models_bank["fastpitch1_1"]
    assert tr.world == 1 and tr.global_batch == 32 and tr.gam == 1
    fresh = T.FastPitch(compute="fp32").state_dict()
    for ck in cks:
        a = torch.load(vo + "/" + ck, weights_only=False)
        b = torch.load(out1 + "/voice_d/" + ck, weights_only=False)
        assert a["iteration"] == b["iteration"] and int(a["training_stage"]) == int(b["training_stage"]) == 3
        worst, moved = 0.0, 0.0
        for k, v in b["state_dict"].items():
            if not v.is_floating_point():
                continue
            worst = max(worst, float((a["state_dict"][k].cpu() - v.cpu()).abs().max()))
            moved = max(moved, float((v.cpu() - fresh[k].cpu()).abs().max()))
        assert worst < 1e-4 and moved > 1e-4, (ck, worst, moved)                                # equal to 1e-4 — and training did move them
        da = torch.cat([(a["state_dict"][k].cpu() - fresh[k].cpu()).flatten() for k in b["state_dict"] if b["state_dict"][k].is_floating_point() and "pitch_" not in k])
        db = torch.cat([(b["state_dict"][k].cpu() - fresh[k].cpu()).flatten() for k in b["state_dict"] if b["state_dict"][k].is_floating_point() and "pitch_" not in k])
        assert _rel(da, db) < 1e-2, (ck, _rel(da, db))                                          # the UPDATES agree, not just the (mostly unchanged) values
    del tr
    mm1.models_bank.clear()
    # ---- a fresh two-rank start on the same output directory is a resume from the newest checkpoint, not a new voice (rank 0 picks the file for both)
    mm3, ws3 = _mm(), _WS()
    assert _run(T.handleTrainer, mm3, dict(base, output_path=out2, batch_size=1, max_iterations=50003), ws3, gpus) is None
    log3 = open(vo + "/training.log").read()
    assert log3.count("New voice") == 1 and "Loading model and optimizer state from %s/FastPitch_checkpoint_4_50003.pt" % vo in log3
    _end_group(mm3, "fastpitch1_1")


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("backend", _backends())
def test_fastpitch_stage_transition_is_taken_by_both_ranks(tmp_path, dp_env, backend):
    """python/fastpitch1_1/xva_train.py:954-970 with two ranks: the epoch losses are averaged over the ranks, so both hit the stopping rule in the same
    epoch; rank 0 re-writes the checkpoint with stage + 1, both ranks' handleTrainer recurse into stage 4 (barrier: the file is complete before anyone
    looks for it), stage 4 finishes the same way and ONE "move to hifi" comes back."""
    from xva_trainer_amd import data as D
    from xva_trainer_amd.fastpitch import xva_train as T
    gpus = dp_env(backend)
    ds = D.write_synthetic_dataset(str(tmp_path / "in" / "voice_e"), n_items=6, seed=8, min_s=0.4, max_s=0.8)
    out = str(tmp_path / "out")
    data = {"dataset_path": ds, "output_path": out, "checkpoint": None, "num_workers": 0, "batch_size": 1, "epochs_per_checkpoint": 1, "force_stage": 3,
            "trainer_options": {"compute": "fp32", "target_delta": 1e9}}
    mm, ws = _mm(), _WS()
    assert _run(T.handleTrainer, mm, data, ws, gpus) == "move to hifi"
    assert mm.models_bank["fastpitch1_1"] == "move to hifi"                                    # python/fastpitch1_1/xva_train.py:160-161
    assert [m for m in ws.sent if m.startswith("Set stage to")] == ["Set stage to: 3 ", "Set stage to: 4 "]
    vo = out + "/voice_e"
    log = open(vo + "/training.log").read()
    assert log.count("Finished training stage 3") == 1 and log.count("Moving to HiFi-GAN") == 1
    done = sorted(f for f in os.listdir(vo) if f.startswith("Stage_"))
    assert len(done) == 2 and done[0].startswith("Stage_3_DONE_") and done[1].startswith("Stage_4_DONE_")
    last = sorted((f for f in os.listdir(vo) if f.startswith("FastPitch_checkpoint_")), key=T.sort_fp)[-1]
    assert int(torch.load(vo + "/" + last, weights_only=False)["training_stage"]) == 5


# ------------------------------------------------------------------------------------------------ HiFi-GAN
def _hifi_pretrained(path):
    """A g_ / do_ pair to fine-tune from (the trainer never trains from scratch): random-init weights of the v1 architecture."""
    from oracle import hifigan as ohg
    from xva_trainer_amd.hifigan.step import HifiganStep
    os.makedirs(path)
    st = HifiganStep("cuda:0", "fp32")
    st.load_state_dicts(ohg.init_generator_sd(1), ohg.init_mpd_sd(2), ohg.init_msd_sd(3))
    sds = st.state_dicts()
    cpu = lambda sd: {k: v.cpu() for k, v in sd.items()}
    torch.save({"generator": cpu(sds["generator"])}, path + "/g_00000000")
    torch.save({"mpd": cpu(sds["mpd"]), "msd": cpu(sds["msd"]), "optim_g": st.optim_g.state_dict(), "optim_d": st.optim_d.state_dict(), "steps": -1, "epoch": -1,
                "avg_loss_per_epoch": [], "ckpts_finetuned": 0}, path + "/do_00000000")
    return cpu(sds["generator"]), cpu(sds["mpd"])


def _close(a, b, init, what):
    """`a` (two ranks) against `b` (one process, the union batch) for a dict of tensors that started at `init`.  AdamW's first steps move EVERY weight by
    about +-lr whatever the size of its gradient, so an element whose gradient is rounding noise (|g| << 1e-6 of its tensor's) may go the other way in
    the two runs — |diff| up to 2 lr x steps on a handful of the 10^7 elements.  Asserted: the parameter vectors agree to 1e-4 (L2), all but 1e-5 of
    the elements agree to 1e-4 absolutely, and the UPDATES (what training changed) agree to 1 %."""
    keys = [k for k in b if b[k].is_floating_point() and k in init]
    va, vb, v0 = (torch.cat([d[k].float().flatten() for k in keys]) for d in (a, b, init))
    assert _rel(va, vb) < 1e-4, (what, _rel(va, vb))
    assert float(((va - vb).abs() > 1e-4).double().mean()) < 1e-5, (what, float((va - vb).abs().max()))
    assert float((vb - v0).abs().max()) > 1e-4, what
    assert _rel(va - v0, vb - v0) < 1e-2, (what, _rel(va - v0, vb - v0))


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("backend", _backends())
def test_hifigan_trainer_world2(tmp_path, dp_env, backend):
    from xva_trainer_amd import data as D
    from xva_trainer_amd.dp_launch import RankGroup
    from xva_trainer_amd.hifigan import xva_train as T
    gpus = dp_env(backend)
    ds = D.write_synthetic_dataset(str(tmp_path / "in" / "voice_h"), n_items=20, seed=9, min_s=0.38, max_s=0.6, with_pitch=False)
    pre = str(tmp_path / "pretrained")
    g0, mpd0 = _hifi_pretrained(pre)
    base = {"dataset_path": ds, "hifigan_checkpoint": pre, "num_workers": 0, "epochs_per_checkpoint": 1, "trainer_options": {"compute": "fp32"}}
    # 20 clips x dm round(1000 / 20) = 50 -> 1000 files; two ranks x batch int(50 * 1.4) = 70 -> 7 iterations an epoch.  Two epochs + 1.
    out2 = str(tmp_path / "out2")
    mm, ws = _mm(), _WS()
    assert _run(T.handleTrainer, mm, dict(base, output_path=out2, batch_size=50, max_iterations=15), ws, gpus) is None
    assert isinstance(mm.models_bank["hifigan"], RankGroup) and ws.sent.count("Set stage to: 5 ") == 1
    vo = out2 + "/voice_h"
    log = open(vo + "/training.log").read()
    assert log.count("Stage 5 |Epoch: 0 | It: 7 | g_00000007") == 1 and log.count("Stage 5 |Epoch: 1 | It: 14 | g_00000014") == 1     # rank 0 alone writes
    assert len(re.findall(r"Stage 5 \| Epoch: \d+ \| It: ", log)) == 15 and "its/s" in log
    assert sorted(os.listdir(vo + "/hifi")) == ["do_00000007", "do_00000014", "g_00000007", "g_00000014"]
    do = torch.load(vo + "/hifi/do_00000014", weights_only=False)
    assert do["steps"] == 14 and do["epoch"] == 1 and abs(do["optim_g"]["param_groups"][0]["lr"] - 2e-4 * 0.999 ** 2) < 1e-12
    # ---- resume in place, then a NEW two-rank start that resumes from the pair rank 0 sees
    assert _run(T.handleTrainer, mm, dict(base, output_path=out2, batch_size=50, max_iterations=16), ws, gpus, resume=True) is None
    assert re.findall(r"It: \d+/\d+ \((\d+)\)", open(vo + "/training.log").read())[-1] == "16"
    _end_group(mm, "hifigan")
    mm2, ws2 = _mm(), _WS()
    assert _run(T.handleTrainer, mm2, dict(base, output_path=out2, batch_size=50, max_iterations=16), ws2, gpus) is None
    log2 = open(vo + "/training.log").read()
    assert "Loading checkpoint from: %s/hifi/g_00000014" % vo in log2 and re.findall(r"It: \d+/\d+ \((\d+)\)", log2)[-1] == "16"
    _end_group(mm2, "hifigan")
    # ---- single process with twice the batch (base 100 -> 140): the same 140 crops per iteration (the crop of an item depends on its place in the
    # global epoch order, not on the rank that draws it)
    out1 = str(tmp_path / "out1")
    mm1 = _mm()
    assert _run(T.handleTrainer, mm1, dict(base, output_path=out1, batch_size=100, max_iterations=8), _WS(), [0]) is None
    tr = mm1.models_bank["hifigan"]
    assert tr.world == 1 and tr.h["batch_size"] == 140 and len(tr.train_loader) == 7
    del tr
    mm1.models_bank.clear()
    _close(torch.load(vo + "/hifi/g_00000007", weights_only=False)["generator"], torch.load(out1 + "/voice_h/hifi/g_00000007", weights_only=False)["generator"], g0, "generator")
    _close(torch.load(vo + "/hifi/do_00000007", weights_only=False)["mpd"], torch.load(out1 + "/voice_h/hifi/do_00000007", weights_only=False)["mpd"], mpd0, "mpd")


# ------------------------------------------------------------------------------------------------ xVAPitch
_XV_SMALL = dict(n_vocab=160, num_languages=31, latent_size=32, embedded_language_dim=4, d_vector_dim=512, hidden_channels_ffn=64, text_layers=2,
                 posterior_layers=3, flow_layers=2, spec_segment_size=8, dropout_p=0.0, sdp_dropout_p=0.0)


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("backend", _backends())
def test_xvapitch_trainer_world2(tmp_path, dp_env, backend):
    """python/xvapitch/xva_train.py's protocol with two ranks: checkpoints every save_step optimiser steps by rank 0 (the other rank waits at the barrier),
    the stage 1 -> 2 transition decided on the rank-averaged discriminator loss, resume, and — with every clip the same length, dropout off and the
    iteration's random draws taken from one stream for the global batch (`world_invariant_noise`) — parameters equal to the single-process run on the
    union batch (per-replica loss normalisation = nn.DataParallel's, which coincides with the global one for equally shaped shards)."""
    from xva_trainer_amd.data import write_synthetic_dataset
    from xva_trainer_amd.dp_launch import RankGroup
    from xva_trainer_amd.xvapitch import xva_train as T
    gpus = dp_env(backend)
    ds = write_synthetic_dataset(str(tmp_path / "in" / "voice_w"), n_items=8, seed=5, min_s=0.9, max_s=0.9, with_se_embs=True, fixed_text="alpha bravo charlie delta.")
    opts = {"compute": "fp32", "allow_random_init": True, "model_kwargs": _XV_SMALL, "world_invariant_noise": True}
    base = {"dataset_path": ds, "checkpoint": None, "num_workers": 0, "lang": "en", "bkp_every_x": 2, "save_step": 3, "trainer_options": opts,
            "priors_path": str(tmp_path / "no_priors")}
    # 8 clips x data_mult 10 = 80 items; two ranks x 4 -> 10 iterations an epoch; gam = ceil(400 / 8) capped... batch_size is PER GPU in this trainer
    out2 = str(tmp_path / "out2")
    mm, ws = _mm(), _WS()
    d2 = dict(base, output_path=out2, batch_size=200, max_iterations=7)
    assert _run(T.handleTrainer, mm, d2, ws, gpus) is None
    assert isinstance(mm.models_bank["xvapitch"], RankGroup) and sum(m.startswith("Set stage to: 1") for m in ws.sent) == 1
    vo = out2 + "/voice_w"
    log = open(vo + "/training.log").read()
    assert "GPUs mult: 2" in log and log.count("New voice") == 1 and "Fine-tune dataset files: 8" in log
    cks = sorted((f for f in os.listdir(vo) if f.startswith("xVAPitch_")), key=T.sort_xvap)
    assert cks == ["xVAPitch_2.pt", "xVAPitch_5.pt"], cks
    assert not [f for f in os.listdir(vo) if ".tmp." in f]
    ck = torch.load(vo + "/xVAPitch_5.pt", weights_only=False)
    assert ck["step"] == 5 and ck["training_stage"] == 1
    # resume in place
    assert _run(T.handleTrainer, mm, dict(d2, max_iterations=8), ws, gpus, resume=True) is None
    _end_group(mm, "xvapitch")
    # a new two-rank start: rank 0's newest checkpoint for both ranks, steps continue
    mm2 = _mm()
    assert _run(T.handleTrainer, mm2, dict(d2, max_iterations=1), _WS(), gpus) is None
    log2 = open(vo + "/training.log").read()
    assert log2.count("New voice") == 1 and "Loading model and optimizer state from %s/xVAPitch_" % vo in log2
    _end_group(mm2, "xvapitch")
    # ---- single process on the union batch
    out1 = str(tmp_path / "out1")
    mm1 = _mm()
    assert _run(T.handleTrainer, mm1, dict(base, output_path=out1, batch_size=400, max_iterations=7), _WS(), [0]) is None
    tr = mm1.models_bank["xvapitch"]
    assert tr.world == 1 and tr.gam == 1
    del tr
    mm1.models_bank.clear()
    a = torch.load(vo + "/xVAPitch_5.pt", weights_only=False)
    b = torch.load(out1 + "/voice_w/xVAPitch_5.pt", weights_only=False)
    init = torch.load(out1 + "/voice_w/xVAPitch_2.pt", weights_only=False)["model"]
    ta = {k: v for k, v in a["model"].items() if torch.is_tensor(v)}
    tb = {k: v for k, v in b["model"].items() if torch.is_tensor(v)}
    _close(ta, tb, {k: v for k, v in init.items() if torch.is_tensor(v)}, "xvapitch")


@pytest.mark.timeout(2400)
@pytest.mark.parametrize("backend", _backends())
def test_xvapitch_stage_transition_is_taken_by_both_ranks(tmp_path, dp_env, backend):
    """xva_train.py:796-851 with two ranks: the checkpoint-time rule runs on the rank-averaged discriminator loss, so stage 1 -> 2 and the end of stage 2
    happen in the same checkpoint interval on both ranks; "Finished training" arrives once."""
    from xva_trainer_amd.data import write_synthetic_dataset
    from xva_trainer_amd.xvapitch import xva_train as T
    gpus = dp_env(backend)
    ds = write_synthetic_dataset(str(tmp_path / "in" / "voice_v"), n_items=6, seed=7, min_s=0.6, max_s=1.0, with_se_embs=True, min_words=3)
    opts = {"compute": "fp32", "allow_random_init": True, "model_kwargs": _XV_SMALL, "target_delta": [1e9, 1e9]}
    data = {"dataset_path": ds, "output_path": str(tmp_path / "out"), "checkpoint": None, "num_workers": 0, "batch_size": 200, "lang": "en", "bkp_every_x": 2,
            "save_step": 2, "trainer_options": opts, "priors_path": str(tmp_path / "no_priors"), "max_iterations": 60}
    mm, ws = _mm(), _WS()
    assert _run(T.handleTrainer, mm, data, ws, gpus) is None
    assert "xvapitch" not in mm.models_bank
    assert [m for m in ws.sent if m.startswith("Set stage to")] == ["Set stage to: 1 ", "Set stage to: 2 "] and ws.sent.count("Finished training\n") == 1
    log = open(data["output_path"] + "/voice_v/training.log").read()
    assert log.count("Finished Stage 1. Moving on..") == 1 and log.count("Finished Stage 2. Stopping training.") == 1
