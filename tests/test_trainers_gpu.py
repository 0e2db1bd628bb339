"""GPU: the trainer protocol mirrors (handleTrainer / FastPitchTrainer / HiFiTrainer / ModelsManager.load_model) drive the HIP engines
end to end: from a reference-layout dataset directory (metadata.csv + wavs/ + pitch/, durations extracted after stage 1), through
stage transitions, checkpoints in the reference's formats (incl. files rebuilt from the layouts recorded from the reference's own
classes), resume with optimizer state, the out-of-memory back-off and the inference wrappers."""
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


def _fp_trainer(mm, ws, out, name, compute="fp32", factory=None):
    mm.sync_init_model("fastpitch1_1", websocket=ws, gpus=[0])
    tr = mm.models_bank["fastpitch1_1"]
    tr.compute = compute
    tr.loader_factory = factory
    tr.init_logs(out + "/" + name)
    return tr


def test_fastpitch_trainer_protocol(tmp_path):
    from xva_trainer_amd.data import SyntheticFastPitchLoader
    mm, ws = _mm(), _WS()
    data = {"dataset_path": str(tmp_path / "in" / "voice_a"), "output_path": str(tmp_path / "out"), "checkpoint": None, "num_workers": 0,
            "batch_size": 4, "epochs_per_checkpoint": 1, "force_stage": 3, "max_iterations": 50003}
    factory = lambda t: SyntheticFastPitchLoader(t.per_rank_batch, n_batches=64, t_text=20, t_mel=90, seed=7)
    tr = _fp_trainer(mm, ws, data["output_path"], "voice_a", factory=factory)
    asyncio.run(tr.start(data, gpus=[0]))
    # stage 3: batch = int(4 * 3.5 * 1 GPU * 10 / max_len) with no wavs -> 14, gam = round(256 / 14) (xva_train.py:387-407)
    assert tr.global_batch == tr.per_rank_batch == 14 and tr.gam == 18 and tr.EPOCH_AVG_SPAN == 6
    assert tr.total_iter == tr.start_iterations + 3 == 50003 and not tr.running                # a new voice starts at start_iterations (:380-385)
    assert any(m.startswith("Set stage to: 3") for m in ws.sent)
    log = open(data["output_path"] + "/voice_a/training.log").read()
    assert "Stage: 3 | Epoch: 1" in log and "frames/s" in log and "Batch size: 14 (Base: 4, Stage mult: 3.5" in log and "New voice" in log
    # checkpoint round trip in the reference's format
    ck_path = data["output_path"] + "/voice_a/FastPitch_checkpoint_1_50003.pt"
    tr.save_checkpoint(force_save=True, total_iter=tr.total_iter, avg_loss_per_epoch=[1.0], fpath=ck_path)
    ck = torch.load(ck_path, weights_only=False)
    assert list(ck) == ["epoch", "iteration", "avg_loss_per_epoch", "training_stage", "state_dict", "optimizer"]
    assert len(ck["state_dict"]) == 185 and ck["state_dict"]["proj.weight"].shape == (80, 384)
    some = next(iter(ck["optimizer"]["state"].values()))
    assert set(some) >= {"step", "exp_avg", "exp_avg_sq", "weight_norm", "adam_norm", "trust_ratio"}
    half = torch.load(data["output_path"] + "/voice_a/voice_a.pt", weights_only=False)
    assert half["proj.weight"].dtype == torch.float16
    assert json.load(open(data["output_path"] + "/voice_a/voice_a.json"))["modelType"] == "FastPitch1.1"
    # resume picks the newest checkpoint of this voice up (not a new voice any more: iteration continues)
    mm2 = _mm()
    tr2 = _fp_trainer(mm2, _WS(), data["output_path"], "voice_a", factory=factory)
    d2 = dict(data); d2["max_iterations"] = 50004
    asyncio.run(tr2.start(d2, gpus=[0]))
    assert tr2.total_iter == 50004 and "New voice" not in "\n".join(tr2.training_log)
    assert torch.equal(tr2.model.state_dict()["pitch_mean"].cpu(), ck["state_dict"]["pitch_mean"].cpu())
    # a real run never falls back to synthetic data silently
    mm3 = _mm()
    tr3 = _fp_trainer(mm3, _WS(), data["output_path"], "voice_z")
    with pytest.raises(FileNotFoundError):
        asyncio.run(tr3.start(dict(data, dataset_path=str(tmp_path / "in" / "voice_z")), gpus=[0]))
    with pytest.raises(NotImplementedError):                                                   # multi-GPU = one process per GPU, never DataParallel
        asyncio.run(_fp_trainer(_mm(), _WS(), data["output_path"], "voice_y").start(dict(data, synthetic_data=True), gpus=[0, 1]))


def test_fastpitch_trainer_from_dataset_directory(tmp_path):
    """metadata.csv + wavs/ + pitch/: stage 1 (aligner) on device-built batches -> durations written in the reference's durs_text/*.npy
    layout -> stages 2 and 3 train from those files."""
    from xva_trainer_amd import data as D
    from xva_trainer_amd.fastpitch.model import FastPitch
    ds = D.write_synthetic_dataset(str(tmp_path / "in" / "voice_c"), n_items=12, seed=4, min_s=0.5, max_s=1.0)
    json.dump({"mean": 180.5, "std": 41.25}, open(ds + "/pitch_stats.json", "w"))
    out = str(tmp_path / "out")
    base = {"dataset_path": ds, "output_path": out, "checkpoint": None, "num_workers": 0, "batch_size": 1, "epochs_per_checkpoint": 1000}
    mm, ws = _mm(), _WS()
    tr = _fp_trainer(mm, ws, out, "voice_c")
    asyncio.run(tr.start(dict(base, max_iterations=50006), gpus=[0]))
    assert int(tr.model.training_stage) == 1 and any(m.startswith("Set stage to: 1") for m in ws.sent)
    assert tr.global_batch == int(1 * 1.5 * 10 / max(tr._dataset_file_lengths())) == 15
    assert len(tr.train_loader) == 3 and tr.gam == 3          # 12 clips x data multiplier 4 = 48 items; round(256 / 15) = 17 capped to one epoch
    assert float(tr.model.pitch_mean[0]) == 180.5 and float(tr.model.pitch_std[0]) == 41.25     # pitch_stats.json -> model buffers (:344-346)
    log = open(out + "/voice_c/training.log").read()
    losses = [float(x) for x in re.findall(r"Stage: 1 .*?loss: ([0-9.]+)", log)]
    assert len(losses) == 6 and losses[-1] < losses[0], losses
    fresh = FastPitch(compute="fp32").state_dict()
    sd = tr.model.state_dict()
    moved = {k for k in sd if not torch.equal(sd[k].cpu(), fresh[k].cpu()) and k not in ("pitch_mean", "pitch_std")}
    assert moved and all(k.startswith("attention.") or k == "encoder.word_emb.weight" for k in moved), moved
    tr.save_checkpoint(force_save=True, total_iter=tr.total_iter, avg_loss_per_epoch=[], fpath=out + "/voice_c/FastPitch_checkpoint_1_%d.pt" % tr.total_iter)
    # stage 2: the trainer extracts the durations first (no durs_text/ yet)
    mm2 = _mm()
    tr2 = _fp_trainer(mm2, _WS(), out, "voice_c")
    asyncio.run(tr2.start(dict(base, force_stage=2, max_iterations=50002), gpus=[0]))     # forcing a later stage restarts at start_iterations (:366-367)
    meta = D.read_metadata(ds)
    enc = D.BasicTextEncoder()
    for name, path, text in meta:
        d = np.load("%s/durs_text/%s.npy" % (ds, name))
        wav, _ = D.read_wav_int16(path)
        assert d.dtype == np.float32 and d.shape == (len(enc.encode(text)),) and int(d.sum()) == 1 + len(wav) // 256 and (d >= 0).all()
    assert "Extracting durations from alignments" in open(out + "/voice_c/training.log").read()
    assert int(tr2.model.training_stage) == 2 and tr2.epochs_per_checkpoint == 3000 and tr2.total_iter == 50002
    assert tr2.per_rank_batch == 48 and "Capping batch size" in open(out + "/voice_c/training.log").read()      # 12 clips x dm 4 < 1 * 12 * 10 / 1.0
    # stage 3 from the same files (pitch cache + durations)
    mm3 = _mm()
    tr3 = _fp_trainer(mm3, _WS(), out, "voice_c", compute="bf16")
    asyncio.run(tr3.start(dict(base, force_stage=3, max_iterations=50002), gpus=[0]))
    assert int(tr3.model.training_stage) == 3
    l3 = [float(x) for x in re.findall(r"Stage: 3 .*?loss: ([0-9.]+)", open(out + "/voice_c/training.log").read())]
    assert len(l3) >= 2 and all(np.isfinite(l3))


def test_fastpitch_trainer_fp16_operand_mode_skips_non_finite_steps_and_backs_the_loss_scale_off(tmp_path, monkeypatch):
    """round 6, compute = "f16" through the trainer: the loss scale comes from the batch geometry, a step whose scaled gradients overflow fp16 is SKIPPED on
    the device (csrc/optim.hip: clip_coef flags it, LAMB leaves weights and moments alone), the skip flag travels with the delayed loss report and the trainer
    halves the scale — torch.cuda.amp.GradScaler's behaviour in the reference (xva_train.py:350,856-859).  Here the first scale is forced absurdly high."""
    from xva_trainer_amd.data import SyntheticFastPitchLoader
    from xva_trainer_amd.fastpitch import engine as E
    real = E.FastPitchEngine._choose_loss_scale
    chosen = []

    def absurd(self, b):
        chosen.append(real(self, b))
        return 2.0 ** 40
    monkeypatch.setattr(E.FastPitchEngine, "_choose_loss_scale", absurd)
    mm, ws = _mm(), _WS()
    n_it = 45
    data = {"dataset_path": str(tmp_path / "in" / "voice_h"), "output_path": str(tmp_path / "out"), "checkpoint": None, "num_workers": 0,
            "batch_size": 4, "epochs_per_checkpoint": 1, "force_stage": 3, "max_iterations": 50000 + n_it}
    factory = lambda t: SyntheticFastPitchLoader(t.per_rank_batch, n_batches=4096, t_text=20, t_mel=90, seed=7)
    tr = _fp_trainer(mm, ws, data["output_path"], "voice_h", compute="f16", factory=factory)
    asyncio.run(tr.start(data, gpus=[0]))
    assert tr.eng.compute == 2 and tr.total_iter == 50000 + n_it
    assert len(chosen) == 1 and 2.0 ** 8 <= chosen[0] <= 2.0 ** 14                 # 14 x 90 x 80 frames-bins ~ 2^16.6, and 2^-4 of it
    log = open(data["output_path"] + "/voice_h/training.log").read()
    skips = len(re.findall(r"non-finite gradient: step skipped", log))
    final = getattr(tr, "_next_loss_scale", None) or tr.eng.loss_scale
    assert skips >= 10 and final == 2.0 ** 40 / 2.0 ** skips                       # one halving per skipped step, nothing else moved the scale
    assert skips < n_it - 3                                                        # ... and the last steps were taken
    assert float(tr.optimizer.skipped) == 0.0
    losses = [float(x) for x in re.findall(r"Stage: 3 .*?loss: ([0-9.]+)", log)]
    assert losses and all(np.isfinite(losses))
    sd = tr.model.state_dict()
    assert all(torch.isfinite(v.float()).all() for v in sd.values())
    from xva_trainer_amd.fastpitch.model import FastPitch
    fresh = FastPitch(compute="fp32").state_dict()
    assert not torch.equal(sd["proj.weight"].cpu(), fresh["proj.weight"].cpu())    # taken steps moved the weights


def test_stage_completion_rewrites_checkpoint_for_the_next_stage(tmp_path):
    """xva_train.py:954-970: patience of 3 on the avg-delta criterion; on completion training_stage + 1 and an EMPTY loss history go into
    the regular FastPitch_checkpoint_* (and Stage_N_DONE_*), so the next trainer starts stage N + 1 with a clean stopping history."""
    from xva_trainer_amd.data import SyntheticFastPitchLoader
    from xva_trainer_amd.fastpitch import xva_train as T
    out = str(tmp_path / "out")
    data = {"dataset_path": str(tmp_path / "in" / "voice_s"), "output_path": out, "checkpoint": None, "num_workers": 0,
            "batch_size": 64, "epochs_per_checkpoint": 1, "force_stage": 3}
    factory = lambda t: SyntheticFastPitchLoader(2, n_batches=2, t_text=12, t_mel=40, seed=11)
    mm, ws = _mm(), _WS()

    async def run():
        # first call: trains stage 3 until the criterion fires, then recurses into stage 4 (bounded by max_iterations)
        real = T.FastPitchTrainer.get_target_delta
        T.FastPitchTrainer.get_target_delta = lambda self, n, stage: 1e9 if stage == 3 else -1e9      # stage 3 converges at once, stage 4 never
        try:
            mm.sync_init_model("fastpitch1_1", websocket=ws, gpus=[0])
            mm.models_bank["fastpitch1_1"].compute = "fp32"
            mm.models_bank["fastpitch1_1"].loader_factory = factory
            orig_sync = mm.sync_init_model

            def sync(key, websocket=None, gpus=[0]):
                orig_sync(key, websocket=websocket, gpus=gpus)
                t = mm.models_bank[key]
                t.compute, t.loader_factory = "fp32", factory
            mm.sync_init_model = sync
            return await T.handleTrainer(mm, dict(data, max_iterations=50030), ws, [0])
        finally:
            T.FastPitchTrainer.get_target_delta = real

    asyncio.run(run())
    log = open(out + "/voice_s/training.log").read()
    assert "Finished training stage 3..." in log
    done = [f for f in os.listdir(out + "/voice_s") if f.startswith("Stage_3_DONE_FastPitch_checkpoint_")]
    assert len(done) == 1
    ck = torch.load(out + "/voice_s/" + done[0], weights_only=False)
    assert int(ck["training_stage"]) == 4 and ck["avg_loss_per_epoch"] == []
    # stage 3 needs len(deltas) >= 2 and three consecutive qualifying epochs: it cannot finish before its 5th epoch
    e3 = [int(x) for x in re.findall(r"Stage: 3 \| Epoch: (\d+) \| iter", log)]
    assert max(e3) >= 4
    # the recursion started stage 4 from the re-written checkpoint (no force_stage) and kept training past one epoch
    assert [m for m in ws.sent if m.startswith("Set stage to:")] == ["Set stage to: 3 ", "Set stage to: 4 "]
    e4 = [int(x) for x in re.findall(r"Stage: 4 \| Epoch: (\d+) \| iter", log)]
    assert e4 and max(e4) - min(e4) >= 2 and "Finished training stage 4" not in log


def test_out_of_memory_backoff(tmp_path):
    """xva_train.py:131-145: an out-of-memory RuntimeError restarts the trainer with the base batch size reduced by 3."""
    from xva_trainer_amd.data import SyntheticFastPitchLoader
    from xva_trainer_amd.fastpitch import xva_train as T
    mm = _mm()
    seen = []
    factory = lambda t: SyntheticFastPitchLoader(2, n_batches=4, t_text=10, t_mel=30, seed=1)
    orig_sync = mm.sync_init_model

    def sync(key, websocket=None, gpus=[0]):
        orig_sync(key, websocket=websocket, gpus=gpus)
        t = mm.models_bank[key]
        t.compute, t.loader_factory = "fp32", factory
        real_init = t.init

        async def init():
            seen.append(t.batch_size)
            if len(seen) < 3:
                raise RuntimeError("HIP out of memory. Tried to allocate 20.00 GiB")
            await real_init()
        t.init = init
    mm.sync_init_model = sync
    data = {"dataset_path": str(tmp_path / "in" / "voice_o"), "output_path": str(tmp_path / "out"), "checkpoint": None, "num_workers": 0,
            "batch_size": 10, "epochs_per_checkpoint": 1, "force_stage": 2, "max_iterations": 50001}
    asyncio.run(T.handleTrainer(mm, data, _WS(), [0]))
    assert seen == [10, 7, 4]
    assert "Reducing base batch size from 10 to 7" in open(data["output_path"] + "/voice_o/training.log").read()


def _rand_like(shape, gen, scale=0.05):
    return torch.randn(tuple(shape), generator=gen) * scale


def test_checkpoints_in_the_reference_layout_load_into_the_trainers(tmp_path, golden_dir):
    """Files rebuilt from the layouts recorded from the reference's own classes (tests/golden/checkpoint_layouts.json, written by
    oracle/gen_checkpoint_layouts.py): a full FastPitch checkpoint with its Lamb state, and a HiFi-GAN g_ / do_ pair whose optimizer
    states come out of torch.optim.AdamW itself in the reference's parameter order."""
    from xva_trainer_amd.data import SyntheticFastPitchLoader, SyntheticHifiLoader
    lay = json.load(open(os.path.join(golden_dir, "checkpoint_layouts.json")))
    gen = torch.Generator().manual_seed(5)
    out = str(tmp_path / "out")
    # ---- FastPitch
    os.makedirs(out + "/voice_r")
    sd = {k: (_rand_like(sh, gen) if "inv_freq" not in k else torch.zeros(sh)) for k, sh, dt in [(a, b, c) for a, b, c in lay["fastpitch"]["state_dict"]]}
    order = lay["fastpitch"]["optimizer"]["param_order"]
    state = {i: {"step": 12, "exp_avg": _rand_like(sd[n].shape, gen), "exp_avg_sq": _rand_like(sd[n].shape, gen).abs(), "weight_norm": torch.tensor(1.0),
                 "adam_norm": torch.tensor(2.0), "trust_ratio": torch.tensor(0.5)} for i, n in enumerate(order)}
    torch.save({"epoch": 7, "iteration": 51234, "avg_loss_per_epoch": [0.5, 0.4], "training_stage": torch.tensor(3), "state_dict": sd,
                "optimizer": {"state": state, "param_groups": [{"lr": 4.4e-4, "betas": (0.9, 0.98), "eps": 1e-9, "weight_decay": 1e-6, "adam": False,
                                                                "params": list(range(len(order)))}]}},
               out + "/voice_r/FastPitch_checkpoint_7_51234.pt")
    data = {"dataset_path": str(tmp_path / "in" / "voice_r"), "output_path": out, "checkpoint": None, "num_workers": 0, "batch_size": 4,
            "epochs_per_checkpoint": 1, "max_iterations": 51235}
    tr = _fp_trainer(_mm(), _WS(), out, "voice_r", factory=lambda t: SyntheticFastPitchLoader(2, n_batches=40, t_text=10, t_mel=30, seed=2))

    asyncio.run(tr.start(dict(data, max_iterations=1), gpus=[0]))      # init() from the checkpoint, then exactly one optimizer step
    assert int(tr.model.training_stage) == 3 and tr.epoch == 8 and tr.avg_loss_per_epoch[:2] == [0.5, 0.4]
    from xva_trainer_amd.fastpitch import params as P
    m = P.from_flat(tr.optimizer.exp_avg, tr.model._table)
    frozen = [n for n in order if n.startswith("duration_predictor.")]    # stage 3 freezes these: parameters and moments stay as loaded
    for n in frozen[:3]:
        assert torch.equal(tr.model.state_dict()[n].cpu(), sd[n]) and torch.equal(m[n].cpu(), state[order.index(n)]["exp_avg"])
    assert tr.optimizer.steps[[t[0] for t in tr.model._table].index("proj.weight")] == 13
    # ---- HiFi-GAN: AdamW states produced by torch.optim.AdamW over the reference parameter order
    hifi = out + "/voice_h/hifi"
    os.makedirs(hifi)
    hl = lay["hifigan"]
    sds = {part: {k: _rand_like(sh, gen) + (1.0 if k.endswith("weight_g") else 0.0) for k, sh, dt in hl[part]} for part in ("generator", "mpd", "msd")}
    for k in list(sds["msd"]):
        if k.endswith("weight_u") or (k.endswith("weight_v") and "discriminators.0." in k):
            sds["msd"][k] = torch.nn.functional.normalize(sds["msd"][k], dim=0)
    opt_sd = {}
    for key, named in (("optim_g", [("", sds["generator"])]), ("optim_d", [("msd.", sds["msd"]), ("mpd.", sds["mpd"])])):
        flat_named = {pre + k: v for pre, d in named for k, v in d.items()}
        params = [torch.nn.Parameter(flat_named[n].clone()) for n in hl[key]["param_order"]]
        opt = torch.optim.AdamW(params, 2e-4, betas=[0.8, 0.99])
        sch = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.999)
        for p in params:
            p.grad = _rand_like(p.shape, gen, 1e-3)
        opt.step(); sch.step(); sch.step()
        opt_sd[key] = opt.state_dict()
    torch.save({"generator": sds["generator"]}, hifi + "/g_00000040")
    torch.save({"mpd": sds["mpd"], "msd": sds["msd"], "optim_g": opt_sd["optim_g"], "optim_d": opt_sd["optim_d"], "steps": 40, "epoch": 2,
                "avg_loss_per_epoch": [], "ckpts_finetuned": 3}, hifi + "/do_00000040")
    mm = _mm()
    mm.sync_init_model("hifigan", websocket=_WS(), gpus=[0])
    th = mm.models_bank["hifigan"]
    th.compute = "fp32"
    th.loader_factory = lambda t: SyntheticHifiLoader(1, n_batches=2)
    th.init_logs(out + "/voice_h")
    hd = {"dataset_path": str(tmp_path / "in" / "voice_h"), "output_path": out, "hifigan_checkpoint": None, "num_workers": 0, "batch_size": 1,
          "epochs_per_checkpoint": 1, "max_iterations": 41}

    async def hg_init():                                              # init() only (no training step: the loaded optimizer state is compared as is)
        th.dataset_input, th.dataset_id, th.dataset_output = hd["dataset_path"], "voice_h", out + "/voice_h"
        th.hifigan_checkpoint, th.workers, th.batch_size, th.epochs_per_checkpoint, th.max_iterations, th.synthetic_data = None, 0, 1, 1, 41, False
        await th.init()
    asyncio.run(hg_init())
    assert th.training_steps == 41 and th.training_epoch == 2 and th.ckpts_finetuned == 3
    got = th.core.state_dicts()
    for part in ("generator", "mpd", "msd"):
        for k, v in sds[part].items():
            assert torch.equal(got[part][k].cpu(), v), (part, k)
    og = th.core.optim_g
    assert og.step_count == 1 and abs(og.param_groups[0]["lr"] - 2e-4 * 0.999 ** 3) < 1e-12      # saved lr (two decays) x the scheduler's initial step
    name, off, n, shape = og.order[5]
    assert torch.equal(og.exp_avg[off:off + n].view(shape).cpu(), opt_sd["optim_g"]["state"][5]["exp_avg"])
    od = th.core.optim_d
    name, off, n, shape = od.order[1]
    assert name == "msd.discriminators.0.convs.0.weight_orig"
    assert torch.equal(od.exp_avg_sq[off:off + n].view(shape).cpu(), opt_sd["optim_d"]["state"][1]["exp_avg_sq"])


def test_hifigan_trainer_protocol_resume_and_inference_wrappers(tmp_path):
    from oracle import hifigan as ohg
    from xva_trainer_amd import data as D
    ds = D.write_synthetic_dataset(str(tmp_path / "in" / "voice_b"), n_items=6, seed=2, min_s=0.3, max_s=0.8, with_pitch=False)
    out = str(tmp_path / "out")
    data = {"dataset_path": ds, "output_path": out, "hifigan_checkpoint": None, "num_workers": 0, "batch_size": 2, "epochs_per_checkpoint": 1}
    mm, ws = _mm(), _WS()
    mm.sync_init_model("hifigan", websocket=ws, gpus=[0])
    tr = mm.models_bank["hifigan"]
    tr.init_logs(out + "/voice_b")
    with pytest.raises(RuntimeError, match="never trains from scratch"):                       # xva_train.py:276-277
        asyncio.run(tr.start(dict(data, max_iterations=1), gpus=[0]))
    # a "pretrained" pair to fine-tune from (random-init weights of the v1 architecture: there are no checkpoints offline)
    pre = str(tmp_path / "pretrained")
    os.makedirs(pre)
    torch.save({"generator": ohg.init_generator_sd(1)}, pre + "/g_00000000")
    from xva_trainer_amd.hifigan.step import HifiganStep
    st = HifiganStep("cuda:0", "fp32")
    st.load_state_dicts(ohg.init_generator_sd(1), ohg.init_mpd_sd(2), ohg.init_msd_sd(3))
    sds = st.state_dicts()
    torch.save({"mpd": {k: v.cpu() for k, v in sds["mpd"].items()}, "msd": {k: v.cpu() for k, v in sds["msd"].items()}, "optim_g": st.optim_g.state_dict(),
                "optim_d": st.optim_d.state_dict(), "steps": -1, "epoch": -1, "avg_loss_per_epoch": [], "ckpts_finetuned": 0}, pre + "/do_00000000")
    del st
    mm, ws = _mm(), _WS()
    mm.sync_init_model("hifigan", websocket=ws, gpus=[0])
    tr = mm.models_bank["hifigan"]
    tr.init_logs(out + "/voice_b")

    async def run(max_it):
        await tr.start(dict(data, hifigan_checkpoint=pre, max_iterations=max_it), gpus=[0])
    # dm = round(1000 / 6) = 167 repetitions of 6 files / batch int(2 * 1.4) = 2 -> 501 iterations per epoch: run one full epoch + 1
    asyncio.run(run(502))
    assert len(tr.train_loader) == 501 and tr.training_steps == 502 and tr.training_epoch == 0
    assert "Set stage to: 5 " in ws.sent
    hifi = out + "/voice_b/hifi"
    files = sorted(os.listdir(hifi))
    assert "g_00000501" in files and "do_00000501" in files
    g = torch.load(hifi + "/g_00000501", weights_only=False)
    assert list(g) == ["generator"] and g["generator"]["conv_pre.weight_v"].shape == (512, 80, 7)
    do = torch.load(hifi + "/do_00000501", weights_only=False)
    assert list(do) == ["mpd", "msd", "optim_g", "optim_d", "steps", "epoch", "avg_loss_per_epoch", "ckpts_finetuned"]
    assert "discriminators.0.convs.0.weight_orig" in do["msd"] and "discriminators.0.convs.0.weight_u" in do["msd"]
    assert len(do["optim_d"]["state"]) == 154 and float(do["optim_d"]["state"][0]["step"]) == 501.0
    assert abs(do["optim_g"]["param_groups"][0]["lr"] - 2e-4 * 0.999) < 1e-12
    log = open(out + "/voice_b/training.log").read()
    assert "Stage 5 | Epoch: 1 | It: 1/501" in log and "Stage 5 |Epoch: 0 | It: 501 | g_00000501 | Mel loss:" in log
    mel_line = float(re.search(r"g_00000501 \| Mel loss: ([0-9.e-]+)", log).group(1))
    assert 0.05 < mel_line < 5.0                                                            # a per-iteration mean, not a mean divided twice
    # resume: parameters, both AdamW states, the step counter and the decayed learning rate come back
    mm2 = _mm()
    mm2.sync_init_model("hifigan", websocket=_WS(), gpus=[0])
    tr2 = mm2.models_bank["hifigan"]
    tr2.init_logs(out + "/voice_b")
    asyncio.run(tr2.start(dict(data, max_iterations=503), gpus=[0]))
    assert tr2.training_steps >= 503 and tr2.core.optim_d.step_count >= 502
    assert abs(tr2.core.optim_g.param_groups[0]["lr"] - 2e-4 * 0.999 * 0.999) < 1e-12
    # ---- inference wrappers (python/models_manager.py:130-150) on the checkpoints the trainers wrote
    from xva_trainer_amd.fastpitch.model import FastPitch
    fp_dir = str(tmp_path / "fp")
    os.makedirs(fp_dir)
    fp_sd = FastPitch(compute="fp32").state_dict()
    fp_sd["duration_predictor.fc.bias"] = torch.tensor([1.5])                              # ~3-4 frames per symbol from a random-init model
    torch.save({k: (v.half() if v.is_floating_point() else v) for k, v in fp_sd.items()}, fp_dir + "/voice.pt")
    json.dump({"version": "2.0", "modelType": "FastPitch1.1"}, open(fp_dir + "/voice.json", "w"))
    mm3 = _mm()
    assert mm3.load_model("infer_hifigan", out + "/voice_b/nope.hg.pt") == "ENOENT"
    mm3.load_model("infer_hifigan", out + "/voice_b/voice_b.hg.pt")
    mm3.load_model("infer_fastpitch1_1", fp_dir + "/voice.pt")
    assert mm3.models("infer_hifigan").ckpt_path.endswith("voice_b.hg.pt") and mm3.models("infer_fastpitch1_1").isReady
    wav_path = str(tmp_path / "preview.wav")
    mm3.models("infer_fastpitch1_1").infer(None, "Hello there, general alpha bravo charlie.", wav_path, None, 0)
    wav, sr = D.read_wav_int16(wav_path)
    assert sr == 22050 and len(wav) % 256 == 0 and len(wav) >= 2048
    mm3.set_device("gpu")
    with pytest.raises(RuntimeError):
        mm3.set_device("cpu")


# ------------------------------------------------------------------------------------------------ xVAPitch (BASELINE configs[4])
_XV_SMALL = dict(n_vocab=160, num_languages=31, latent_size=32, embedded_language_dim=4, d_vector_dim=512, hidden_channels_ffn=64, text_layers=2,
                 posterior_layers=3, flow_layers=2, spec_segment_size=8)


def _xv_trainer(mm, ws, out, name):
    mm.sync_init_model("xvapitch", websocket=ws, gpus=[0])
    tr = mm.models_bank["xvapitch"]
    tr.compute, tr.model_kwargs, tr.allow_random_init = "fp32", dict(_XV_SMALL), True
    tr.init_logs(out + "/" + name)
    return tr


def test_xvapitch_trainer_runs_checkpoints_and_resumes(tmp_path):
    """python/xvapitch/xva_train.py's protocol on a synthetic dataset directory (metadata.csv + wavs/ + pitch/ + se_embs/): epochs roll over, the
    schedulers step per epoch, a checkpoint in the reference's keys is written every save_step optimiser steps, a second trainer resumes from it
    (model, both AdamW states, step count, stage), and handleTrainer drives the same thing through ModelsManager."""
    from xva_trainer_amd.data import write_synthetic_dataset
    from xva_trainer_amd.xvapitch import xva_train as XT
    ds = write_synthetic_dataset(str(tmp_path / "in" / "voice_x"), n_items=6, seed=3, min_s=0.6, max_s=1.2, with_se_embs=True, min_words=3)
    data = {"dataset_path": ds, "output_path": str(tmp_path / "out"), "checkpoint": None, "num_workers": 0, "batch_size": 400, "lang": "en",
            "bkp_every_x": 2, "save_step": 4, "max_iterations": 9}
    mm, ws = _mm(), _WS()
    tr = _xv_trainer(mm, ws, data["output_path"], "voice_x")
    asyncio.run(tr.start(data, gpus=[0]))
    out = data["output_path"] + "/voice_x"
    assert tr.gam == 1 and tr.training_iters == 9 and tr.total_steps_done == 9 and not tr.running
    assert tr.epoch >= 2                                                    # 6 clips x data_mult 10 in batches of 6 (capped): 10 iterations an epoch... at least two roll-overs with the cap
    assert any(m.startswith("Set stage to: 1") for m in ws.sent)
    log = open(out + "/training.log").read()
    assert "New voice" in log and "Stage: 1 | Steps:" in log and "frames/s" in log and "Fine-tune dataset files: 6" in log
    cks = sorted(f for f in os.listdir(out) if f.startswith("xVAPitch_"))
    assert cks == ["xVAPitch_3.pt", "xVAPitch_7.pt"], cks                   # named by the steps done BEFORE the saving step is counted (xva_train.py:853,894)
    ck = torch.load(out + "/xVAPitch_7.pt", weights_only=False)
    assert list(ck) == ["model", "optimizer", "scaler", "step", "epoch", "lr", "date", "avg_disc_loss_per_epoch", "avg_disc_loss_per_epoch_deltas",
                        "training_stage"]
    assert ck["step"] == 7 and ck["training_stage"] == 1 and len(ck["optimizer"]) == 2
    assert ck["lr"] < 0.000175                                              # ExponentialLR stepped at the epoch roll-overs
    assert any(k.startswith("waveform_decoder.") for k in ck["model"]) and any(k.startswith("disc.nets.0.") for k in ck["model"])
    st0 = ck["optimizer"][0]["state"]
    assert len(st0) > 100 and set(next(iter(st0.values()))) == {"step", "exp_avg", "exp_avg_sq"}
    assert torch.load(out + "/voice_x.pt", weights_only=False)["emb_l.weight"].dtype == torch.float16
    meta = json.load(open(out + "/voice_x.json"))
    assert meta["modelType"] == "xVAPitch" and len(meta["games"][0]["base_speaker_emb"]) == 512
    assert json.load(open(out + "/graphs.json"))["stages"]["1"]["loss"]
    # ---- resume: not a new voice, steps continue, weights and moments are the checkpoint's
    tr2 = _xv_trainer(_mm(), _WS(), data["output_path"], "voice_x")
    asyncio.run(tr2.start(dict(data, max_iterations=1), gpus=[0]))
    assert tr2.total_steps_done == 8 and "New voice" not in "\n".join(tr2.training_log[-12:])
    assert tr2.optimizer[0].step_count == ck["optimizer"][0]["state"][0]["step"] + 1
    tr3 = _xv_trainer(_mm(), _WS(), data["output_path"], "voice_x")
    tr3.dataset_input, tr3.dataset_id, tr3.dataset_output, tr3.batch_size, tr3.lang = ds, "voice_x", out, 400, "en"
    tr3.backup_model_every_x_ckpt, tr3.backup_model_counter, tr3.checkpoint, tr3.workers, tr3.learning_rate = 2, 0, None, 0, 0.000175
    tr3.save_step, tr3.max_iterations, tr3.priors_path, tr3.force_stage = 4, None, None, None
    asyncio.run(tr3.init())
    sd = tr3.model_state_dict()
    for k in ("emb_l.weight", "flow.flows.1.enc.in_layers.0.weight_v", "waveform_decoder.ups.1.weight_v", "disc.nets.2.convs.1.weight_g"):
        assert torch.equal(sd[k].cpu(), ck["model"][k]), k
    i_dec = [k for k, _ in tr3.optimizer[0].order()].index("waveform_decoder.ups.1.weight_v")
    m_view = tr3.optimizer[0]._state_views()["waveform_decoder.ups.1.weight_v"][0]
    assert torch.equal(m_view.cpu(), ck["optimizer"][0]["state"][i_dec]["exp_avg"])
    # ---- handleTrainer + ModelsManager (server.py's entry, python/xvapitch/xva_train.py:86-215)
    mm4, ws4 = _mm(), _WS()
    mm4.sync_init_model("xvapitch", websocket=ws4, gpus=[0])
    t4 = mm4.models_bank["xvapitch"]
    t4.compute, t4.model_kwargs, t4.allow_random_init = "fp32", dict(_XV_SMALL), True
    asyncio.run(XT.handleTrainer(mm4, dict(data, max_iterations=1, checkpoint="[base]"), ws4, [0]))
    assert t4.total_steps_done == 8 and t4.ckpt_path.endswith("xVAPitch_7.pt")      # the output directory's newest checkpoint wins over "[base]"
    # ---- the inference wrapper server.py loads for the exported voice (python/models_manager.py:141-143, xva_train.py:1396-1467)
    from xva_trainer_amd.xvapitch.xva_train import xVAPitchModel, LANG_CODES
    assert LANG_CODES.index("en") == 5 and len(LANG_CODES) == 31                   # index into the sorted lang_names keys (dataset.py:123)
    mm5 = _mm()
    assert mm5.load_model("infer_xvapitch", out + "/nope.pt") == "ENOENT"
    mm5.models_bank["infer_xvapitch"] = xVAPitchModel(None, False, "cuda:0", mm5, model_kwargs={k: v for k, v in _XV_SMALL.items() if k != "spec_segment_size"},
                                                      text_to_sequence=lambda text: ([1 + (ord(ch) % 20) for ch in text], None))
    mm5.load_model("infer_xvapitch", out + "/voice_x.pt")
    inf = mm5.models_bank["infer_xvapitch"]
    assert inf.ckpt_path == out + "/voice_x.pt"
    assert torch.equal(inf.model.state_dict()["emb_l.weight"].cpu(), torch.load(out + "/voice_x.pt", weights_only=False)["emb_l.weight"].float())
    assert inf.infer("hello there", str(tmp_path / "o.wav"), meta["games"][0]["base_speaker_emb"]) == ""
    import scipy.io.wavfile
    sr, wav = scipy.io.wavfile.read(str(tmp_path / "o.wav"))
    assert sr == 22050 and wav.dtype == np.int16 and wav.size % 256 == 0 and wav.size >= 11 * 256 and int(np.abs(wav).max()) >= 32000
    durs = inf.model.infer(torch.tensor([[3, 4, 5]], device="cuda"), torch.randn(512, device="cuda"), 5, inf.decoder, durs_only=True)
    assert durs.shape == (1, 1, 3) and bool((durs >= 1).all())


def test_xvapitch_checkpoint_layout_is_the_references():
    """Keys, shapes and the two optimisers' parameter orders of the trainer's model at the reference's own switches (--big 1 --pitch 1) against the
    layout recorded from the reference's classes (tests/golden/xvapitch_checkpoint_layout.json, oracle/gen_xvapitch_checkpoint_layout.py); a
    checkpoint rebuilt in that layout (seeded values) loads into the trainer and comes back out bit for bit."""
    import tempfile
    from xva_trainer_amd.xvapitch import xva_train as XT
    lay = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "xvapitch_checkpoint_layout.json")))
    tr = XT.xVAPitchTrainer(logging.getLogger("t"), False, [0], None)
    tr.compute = "bf16"
    tr.step = tr.init_model(torch.device("cuda:0"))
    ac, dec, disc = tr.step.gen.acoustic, tr.step.gen.decoder, tr.step.disc
    tr.optimizer = [XT.FlatGroupAdamW.for_generator(ac, dec, lr=0.000175), XT.FlatGroupAdamW.for_discriminator(disc, lr=0.0002)]
    mine = tr.model_state_dict()
    ref = {k: tuple(shape) for k, shape, _ in lay["state_dict"]}
    assert set(mine) == set(ref), sorted(set(mine) ^ set(ref))[:10]
    assert all(tuple(mine[k].shape) == ref[k] for k in ref), [k for k in ref if tuple(mine[k].shape) != ref[k]][:5]
    assert [k for k, _ in tr.optimizer[0].order()] == lay["optimizer0_param_order"]
    assert [k for k, _ in tr.optimizer[1].order()] == lay["optimizer1_param_order"]
    gen = torch.Generator().manual_seed(5)
    sd = {k: torch.randn(shape, generator=gen) * 0.01 for k, shape, _ in lay["state_dict"]}
    for k in list(sd):
        if k.endswith("weight_g"):
            sd[k] = sd[k].abs() + 0.5
    opt_sd = []
    for order, lr in ((lay["optimizer0_param_order"], 0.00016), (lay["optimizer1_param_order"], 0.00016)):
        state = {i: {"step": torch.tensor(7.0), "exp_avg": torch.randn(ref[k], generator=gen) * 1e-3, "exp_avg_sq": torch.rand(ref[k], generator=gen) * 1e-6}
                 for i, k in enumerate(order)}
        opt_sd.append({"state": state, "param_groups": [{"lr": lr, "betas": (0.8, 0.99), "eps": 1e-9, "weight_decay": 0.01, "params": list(range(len(order)))}]})
    ck = {"model": dict(sd, avg_disc_loss_per_epoch=[[0.5], []], avg_disc_loss_per_epoch_deltas=[[], []]), "optimizer": opt_sd, "scaler": {}, "step": 1234,
          "epoch": 3, "lr": 0.00016, "date": "x", "avg_disc_loss_per_epoch": [[0.5], []], "avg_disc_loss_per_epoch_deltas": [[], []], "training_stage": 2}
    with tempfile.TemporaryDirectory() as d:
        path = d + "/xVAPitch_1234.pt"
        torch.save(ck, path)
        tr.dataset_output = d
        tr.training_log, tr.training_log_live_line = [], ""
        epoch, steps, adl, _ = tr.load_checkpoint(path)
    assert steps == 1234 and tr.training_stage == 2 and adl == [[0.5], []] and tr.optimizer[0].param_groups[0]["lr"] == 0.00016
    back = tr.model_state_dict()
    assert all(torch.equal(back[k].cpu(), sd[k]) for k in sd)
    osd = tr.optimizer[0].state_dict()
    i = lay["optimizer0_param_order"].index("posterior_encoder.pre.weight")          # held zero-padded to 520 bins, stored at the checkpoint's 513
    assert torch.equal(osd["state"][i]["exp_avg"], opt_sd[0]["state"][i]["exp_avg"]) and tr.optimizer[0].step_count == 7
