"""FastPitchTrainer / handleTrainer — the trainer protocol of python/fastpitch1_1/xva_train.py:57-176,185-1081 on the HIP
engine (training stages 1-4).

Kept from the reference, because the Electron UI / server.py consume them: module-level `async handleTrainer(models_manager,
data, websocket, gpus, resume)` returning None | "move to hifi" (checkpoint resolution incl. "[male]" / "[female]", the
out-of-memory back-off, the stage-to-stage recursion); class `FastPitchTrainer(logger, PROD, gpus, models_manager, websocket)` with
`async start / init / iteration`, `pause`, `finish_epoch`, `save_checkpoint`, `load_checkpoint`, `extract_durations` and the flags
`running / is_init / JUST_FINISHED_STAGE / END_OF_TRAINING`; the `data` dict keys; `training.log` + `graphs.json`; ws strings
"Set stage to: N "; checkpoint files `FastPitch_checkpoint_{epoch}_{iter}.pt` (keep last 2), `Stage_{n}_DONE_...pt`,
`{dataset_id}.pt` (fp16 state_dict) and `{dataset_id}.json`; LAMB(lr 0.1, betas (0.9, 0.98), eps 1e-9, wd 1e-6), clip 1000,
warm-up 1000; the per-stage batch multipliers {1: 1.5, 2: 12, 3: 3.5, 4: 4} x GPUs x 10 / max clip seconds and
gam = max(1, round(256 / batch)); stage freezing; the loss-delta stopping rule (mean of the last EPOCH_AVG_SPAN relative
epoch-loss deltas <= target for 3 consecutive epochs, at least 20 epochs in stage 2); on stage completion the checkpoint is
re-written with training_stage + 1 and an empty loss history before the bare raise.
Changed on purpose: the step is 3 C calls (no autograd graph, no GradScaler: bf16 needs no loss scaling); training.log is
appended, not rewritten, each step; batches are built on the device from the dataset directory (xva-trainer_amd/data.py);
multi-GPU is one process per GPU (dp.GradSync over RCCL) instead of single-process nn.DataParallel: handleTrainer(..., gpus=[0..N-1]) in the
server process spawns and supervises the N rank workers (xva-trainer_amd/dp_launch.py); rank 0 keeps the ws / log / checkpoint duties.
"""
import contextlib
import gc
import json
import os
import time
import traceback
import wave

import numpy as np
import torch

from .. import dp_launch
from ..dp_common import RankMixin, trainer_options
from . import engine as E
from . import params as P
from .lamb import Lamb
from .loss_function import FastPitchLoss  # noqa: F401  (re-exported like the reference module)
from .model import FastPitch

STAGE_BS_MULT = {1: 1.5, 2: 12, 3: 3.5, 4: 4}          # xva_train.py:387-398


def sort_fp(x):
    return int(x.split("FastPitch_checkpoint_")[-1].split(".")[0].split("_")[0])


def adjust_learning_rate(total_iter, opt, learning_rate, warmup_iters=None):
    if warmup_iters == 0:
        scale = 1.0
    elif total_iter > warmup_iters:
        scale = 1.0 / (total_iter ** 0.5)
    else:
        scale = total_iter / (warmup_iters ** 1.5)
    for param_group in opt.param_groups:
        param_group["lr"] = learning_rate * scale


def _atomic_save(obj, path):
    """torch.save under a temporary name, then os.replace: a reader (another rank after a stage change, a resumed run) sees the old file
    or the new one, never a partial write."""
    tmp = "%s.tmp.%d" % (path, os.getpid())
    torch.save(obj, tmp)
    os.replace(tmp, path)


def _is_oom(e):
    s = str(e)
    return "out of memory" in s.lower() or "PYTORCH_CUDA_ALLOC_CONF" in s or "PYTORCH_HIP_ALLOC_CONF" in s


def resolve_checkpoint(trainer, ckpt_fname, dataset_output):
    """xva_train.py:73-102: newest checkpoint in the output directory, else "[male]" / "[female]" pretrained, else the newest in a
    given directory, else the given file."""
    final = None
    if ckpt_fname is None:
        return None
    if os.path.exists(dataset_output):
        ckpts = sorted([c for c in os.listdir(dataset_output) if c.startswith("FastPitch_checkpoint_")], key=sort_fp)
        if ckpts:
            final = "%s/%s" % (dataset_output, ckpts[-1])
    if ckpt_fname == "[male]":
        final = trainer.pretrained_ckpt_male
    elif ckpt_fname == "[female]":
        final = trainer.pretrained_ckpt_female
    else:
        if final is None and os.path.isdir(ckpt_fname):
            ckpts = sorted([c for c in os.listdir(ckpt_fname) if c.startswith("FastPitch_checkpoint_")], key=sort_fp)
            if ckpts:
                final = "%s/%s" % (ckpt_fname, ckpts[-1])
        if final is None:
            final = ckpt_fname
    return final


async def handleTrainer(models_manager, data, websocket, gpus, resume=False):
    """python/fastpitch1_1/xva_train.py:57-176.  gpus=[0, 1, ...] in the server process: dp_launch spawns one rank worker per GPU, each of which
    runs this function with its own GPU (the reference wraps the model in nn.DataParallel instead, :465-466)."""
    if dp_launch.wants_rank_group("fastpitch1_1", models_manager, gpus, resume):
        return await dp_launch.handle_trainer("fastpitch1_1", models_manager, data, websocket, gpus, resume)
    gc.collect()
    torch.cuda.empty_cache()
    if not resume:
        models_manager.sync_init_model("fastpitch1_1", websocket=websocket, gpus=gpus)
        trainer = models_manager.models_bank["fastpitch1_1"]
        dataset_id = data["dataset_path"].split("/")[-1]
        dataset_output = data["output_path"] + "/" + dataset_id
        trainer.init_logs(dataset_output=dataset_output)
        data["checkpoint"] = resolve_checkpoint(trainer, data.get("checkpoint"), dataset_output)
    else:
        trainer = models_manager.models_bank["fastpitch1_1"]
    try:
        return await trainer.start(data, gpus=gpus, resume=resume)
    except KeyboardInterrupt:
        trainer.running = False
        raise
    except RuntimeError as e:
        running = trainer.running
        trainer.running = False
        stage_finished = trainer.stage_finished
        for attr in ("train_loader", "dataloader_iterator", "optimizer", "grads", "sync"):
            if hasattr(trainer, attr):
                try:
                    delattr(trainer, attr)
                except Exception:
                    pass
        gc.collect()
        torch.cuda.empty_cache()
        if _is_oom(e):                                                       # xva_train.py:131-145: retry with the base batch size - 3
            trainer.print_and_log("Out of VRAM")
            if running and int(data["batch_size"]) > 3 and trainer.world == 1:      # a rank worker reports it: the parent restarts the whole group
                trainer.print_and_log("============= Reducing base batch size from %s to %s" % (data["batch_size"], int(data["batch_size"]) - 3),
                                      save_to_file=trainer.dataset_output)
                data["batch_size"] = int(data["batch_size"]) - 3
                models_manager.models_bank.pop("fastpitch1_1", None)
                del trainer
                gc.collect()
                torch.cuda.empty_cache()
                return await handleTrainer(models_manager, data, websocket, gpus)
            models_manager.models_bank.pop("fastpitch1_1", None)
            raise
        if trainer.JUST_FINISHED_STAGE:
            trainer.print_and_log("Finished training stage %d...\n" % stage_finished if stage_finished < 4 else "Moving to HiFi-GAN...\n",
                                  save_to_file=trainer.dataset_output)
            trainer.JUST_FINISHED_STAGE = False
            trainer.is_init = False
            models_manager.models_bank.pop("fastpitch1_1", None)
            del trainer
            gc.collect()
            if stage_finished >= 4:
                models_manager.models_bank["fastpitch1_1"] = "move to hifi"
                return "move to hifi"
            data = dict(data)
            data.pop("force_stage", None)       # the re-written checkpoint carries the next stage (xva_train.py:958-966)
            return await handleTrainer(models_manager, data, websocket, gpus)
        models_manager.models_bank.pop("fastpitch1_1", None)
        raise


class FastPitchTrainer(RankMixin):
    def __init__(self, logger, PROD, gpus, models_manager, websocket=None, compute="bf16", loader_factory=None):
        self.logger, self.PROD, self.gpus, self.models_manager, self.websocket = logger, PROD, gpus, models_manager, websocket
        self.compute = compute
        self.loader_factory = loader_factory
        self.ckpt_path, self.isReady = "None", True
        self.model = None
        self.running = self.is_init = self.logs_are_init = False
        self.JUST_FINISHED_STAGE = self.END_OF_TRAINING = False
        self.training_log, self.training_log_live_line, self.graphs_json = [], "", None
        self.dataset_input = self.dataset_output = self.dataset_id = None
        self.force_stage = None
        self.stage_finished = 0
        self.EPOCH_AVG_SPAN, self.target_delta = 20, 0.0
        self._rank_env()
        root = "./resources/app" if PROD else "."
        self.pretrained_ckpt_male = root + "/python/fastpitch1_1/pretrained_models/f4_nate_FastPitch_checkpoint_5760_67000.pt"
        self.pretrained_ckpt_female = root + "/python/fastpitch1_1/pretrained_models/f4_nora_FastPitch_checkpoint_4520_65550.pt"

    # ---- logs the UI reads from disk (xva_train.py:226-238,546-569) ----
    def print_and_log(self, line=None, end="\n", flush=False, save_to_file=None):
        if line is None:
            line = self.training_log_live_line
        else:
            self.training_log.append(line)
        if self.rank == 0 and save_to_file:
            os.makedirs(save_to_file, exist_ok=True)
            with open(save_to_file + "/training.log", "a") as f:
                f.write(line.rstrip() + "\n")

    def init_logs(self, dataset_output):
        self.dataset_output = dataset_output
        os.makedirs(dataset_output, exist_ok=True)
        gpath = dataset_output + "/graphs.json"
        if os.path.exists(gpath):
            with open(gpath) as f:
                self.graphs_json = json.load(f)
        else:
            self.graphs_json = {"stages": {str(s): {"loss": [], "loss_delta": [], "target_delta": None} for s in range(1, 6)}}
        self.logs_are_init = True

    def _save_graphs(self):
        if self.rank == 0:
            with open(self.dataset_output + "/graphs.json", "w") as f:
                json.dump(self.graphs_json, f)

    def pause(self, websocket=None):
        self.request_stop()

    # ---- xva_train.py:675-755 ----
    async def start(self, data, gpus=None, resume=False):
        if self.running:
            return
        self._begin_run()
        self.running = True
        if not resume:
            if gpus is not None:
                self.gpus = gpus
            self.force_stage = int(data["force_stage"]) if "force_stage" in data else None
            self.dataset_input = data["dataset_path"]
            self.dataset_id = self.dataset_input.split("/")[-1]
            self.dataset_output = data["output_path"] + "/" + self.dataset_id
            os.makedirs(self.dataset_output, exist_ok=True)
            self.checkpoint = data.get("checkpoint")
            self.workers = data.get("num_workers", 0)
            self.batch_size = int(data["batch_size"])
            self.epochs_per_checkpoint = int(data.get("epochs_per_checkpoint", 1))
            self.max_iterations = data.get("max_iterations")          # benchmark / test hook (not in the reference)
            self.synthetic_data = bool(data.get("synthetic_data", False))   # explicit opt-in (bench / tests); never a silent fallback
            opts = trainer_options(data)                                    # tests / bench (a rank worker cannot be handed Python objects)
            self.compute = opts.get("compute", self.compute)
            self.p_dropout = opts.get("p_dropout")
            self.target_delta_override = opts.get("target_delta")
            self.prefetch = bool(opts.get("prefetch", os.environ.get("XVA_PREFETCH", "1") != "0"))
            self.learning_rate, self.weight_decay = 0.1, 1e-6
            self.dur_predictor_loss_scale = self.pitch_predictor_loss_scale = 0.1
            self.attn_loss_scale = 1.0                                 # xva_train.py:704
            self.warmup_steps, self.grad_clip_thresh = 1000, 1000
        while self.running and not self.JUST_FINISHED_STAGE and not self.END_OF_TRAINING:
            await self.iteration()
            self._sync_stop()
        if getattr(self, "is_init", False):
            self._drain_pending()              # a pause lands between two iterations: the last micro-batch's report is still in flight

    def get_target_delta(self, num_data_lines, stage):
        """xva_train.py:589-672 (target deltas; freezing is implemented by the engine's per-stage trainable ranges)."""
        if stage == 1:
            if num_data_lines > 4000: return 2e-5
            if num_data_lines > 2000: return 15e-5
            return 4e-4
        if stage == 2:
            td = 5e-4
            if num_data_lines > 4000: td = 5e-5
            elif num_data_lines > 2000: td = 1e-4
            if num_data_lines < 500: td = 4e-3
            return td * 1.5
        if stage == 3:
            td = 6e-4
            if num_data_lines > 4000: td = 5e-5
            elif num_data_lines > 2000: td = 1e-4
            if num_data_lines < 500: td = 2e-3 if num_data_lines < 250 else 1e-3
            return td * 2.5
        td = 25e-5
        if num_data_lines > 4000: td = 35e-6
        elif num_data_lines > 2000: td = 1e-4
        if num_data_lines < 500: td = 15e-4 if num_data_lines < 250 else 45e-5
        return td * 3.0

    # ---- dataset statistics (xva_train.py:304-335,493-536) ----
    def _dataset_file_lengths(self):
        lengths = []
        meta = os.path.join(self.dataset_input, "metadata.csv")
        if not os.path.exists(meta):
            return lengths
        with open(meta, encoding="utf-8") as f:
            for line in f.read().split("\n"):
                if not line.strip():
                    continue
                fname = line.split("|")[0]
                fname = "%s/wavs/%s" % (self.dataset_input, fname + ("" if fname.endswith(".wav") else ".wav"))
                if os.path.exists(fname):
                    with contextlib.closing(wave.open(fname, "r")) as w:
                        lengths.append(w.getnframes() / float(w.getframerate()))
        return lengths

    def get_or_calculate_pitch_stats(self):
        """pitch_stats.json {mean, std} in Hz (xva_train.py:493-536) -> the model's pitch_mean / pitch_std buffers.  Computing them needs
        librosa.pyin over the whole set: CPU preprocessing of the reference, outside this path — a missing file is reported, not guessed."""
        path = os.path.join(self.dataset_input, "pitch_stats.json")
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            self.print_and_log("pitch_mean: %s | pitch_std: %s" % (d["mean"], d["std"]), save_to_file=self.dataset_output)
            return float(d["mean"]), float(d["std"])
        self.print_and_log("No existing pitch mean/std stats (pitch_stats.json is written by the reference's pyin preprocessing); "
                           "keeping the checkpoint's pitch_mean / pitch_std.", save_to_file=self.dataset_output)
        return None, None

    def _make_loader(self, stage, dm):
        if self.loader_factory:
            loader = self.loader_factory(self)
            if loader is not None:
                return loader
        if self.synthetic_data:
            from ..data import SyntheticFastPitchLoader
            return SyntheticFastPitchLoader(self.per_rank_batch, seed=1234 + self.rank, with_prior=stage == 1)
        if not os.path.exists(os.path.join(self.dataset_input, "metadata.csv")):
            raise FileNotFoundError("%s/metadata.csv not found: the trainer reads the reference's dataset layout (metadata.csv + wavs/); "
                                    "synthetic data needs the explicit `synthetic_data` opt-in" % self.dataset_input)
        from ..data import FastPitchFileLoader, read_metadata
        n_items = len(read_metadata(self.dataset_input)) * max(1, dm) // self.world
        if 0 < n_items < self.per_rank_batch:       # a batch larger than the (repeated) dataset would leave drop_last with nothing to train on
            self.print_and_log("Capping batch size to %d (dataset size x data multiplier)" % n_items, save_to_file=self.dataset_output)
            self.per_rank_batch = n_items
            self.global_batch = n_items * self.world
            self.gam = max(1, round(256 / self.global_batch))
        ld = FastPitchFileLoader(self.dataset_input, self.per_rank_batch, stage, self.device, seed=1234, rank=self.rank, world=self.world, dm=dm)
        if getattr(self, "prefetch", True):       # batch i + 1 is read, staged, uploaded and collated under step i (data.Prefetcher)
            from ..data import Prefetcher
            ld = Prefetcher(ld, self.device, depth=2)
        return ld

    async def init(self):
        self.device = dev = self._init_distributed()
        torch.cuda.set_device(dev)
        torch.manual_seed(1234 + self.rank)
        np.random.seed(1234 + self.rank)
        self.print_and_log("Dataset: %s" % self.dataset_input, save_to_file=self.dataset_output)
        self._barrier()                                       # every rank's writes to the output directory so far are done
        ckpt_path = self._from_rank0(self.last_checkpoint(self.dataset_output) if self.rank == 0 else None)   # one view of "newest": the ranks must load the SAME stage
        if ckpt_path is None:
            ckpt_path = self.checkpoint
            self.print_and_log("Checkpoint: %s" % ckpt_path, save_to_file=self.dataset_output)
        self.ckpt_path = str(ckpt_path)
        self.start_iterations = 50000
        lengths = self._dataset_file_lengths()
        num_data_lines = len(lengths)
        for lim, it in ((1000, 47500), (2000, 45000), (4000, 42500), (8000, 40000)):
            if num_data_lines > lim:
                self.start_iterations = it
        pitch_mean, pitch_std = self.get_or_calculate_pitch_stats()

        self.model = FastPitch(logger=self.logger, compute=self.compute).to(dev)
        if getattr(self, "p_dropout", None) is not None:
            self.model.p_dropout = float(self.p_dropout)
        self.model.seed = 1234 + self.rank                    # dropout masks differ across DP ranks (xva_train.py:294-295 seeds per rank)
        self.model.train()
        self.eng = self.model._get_engine()
        self.optimizer = Lamb(self.model.flat.data, self.model._table, lr=self.learning_rate, betas=(0.9, 0.98), eps=1e-9,
                              weight_decay=self.weight_decay)
        self.grads = torch.zeros_like(self.model.flat.data)
        stage, start_epoch, start_iter, self.avg_loss_per_epoch = 1, 1, 0, []
        if ckpt_path and os.path.exists(str(ckpt_path)):
            stage, start_epoch, start_iter, self.avg_loss_per_epoch = self.load_checkpoint(ckpt_path)
        if pitch_mean is not None:                            # xva_train.py:344-346 (the reference sets them before load_checkpoint; the file wins)
            self.model.pitch_mean[0], self.model.pitch_std[0] = pitch_mean, pitch_std
        if (stage == 5 and self.force_stage is None) or self.force_stage == 5:
            self.END_OF_TRAINING = self.JUST_FINISHED_STAGE = True
            self.stage_finished = 4
            raise RuntimeError("FastPitch training already finished")
        if self.force_stage:
            self.print_and_log("Forcing stage: %d" % self.force_stage, save_to_file=self.dataset_output)
            if stage < self.force_stage and stage != 3:
                start_iter = self.start_iterations
            self.avg_loss_per_epoch = []
            stage = self.force_stage
        self.total_iter = start_iter
        if ckpt_path is None or self.dataset_id not in str(ckpt_path):            # IS_NEW (xva_train.py:380-385)
            self.print_and_log("New voice", save_to_file=self.dataset_output)
            stage = self.force_stage or 1
            self.total_iter = self.start_iterations
            self.avg_loss_per_epoch = []
        stage = max(1, min(4, int(stage)))
        self.model.training_stage = torch.tensor(stage)
        if self.websocket is not None:
            await self.websocket.send("Set stage to: %d " % stage)

        # ---- batch size per stage (xva_train.py:387-407).  batch_size is the GLOBAL batch, split over the ranks.
        if stage == 2:
            self.epochs_per_checkpoint = self.epochs_per_checkpoint * 3
        mult = STAGE_BS_MULT[stage]
        file_lengths_bs_mult = 10 / max(lengths) if lengths else 1.0
        base = self.batch_size
        self.global_batch = max(1, int(base * mult * self.world * file_lengths_bs_mult))
        self.per_rank_batch = max(1, self.global_batch // self.world)
        self.global_batch = self.per_rank_batch * self.world
        self.gam = max(1, round(256 / self.global_batch))
        self.print_and_log(["", "Stage 1: Pre-training only the alignment.", "Stage 2: Pre-training durations predictor",
                            "Stage 3: Fine-tuning pitch/energy/mel", "Stage 4: Fine-tuning mel"][stage], save_to_file=self.dataset_output)
        self.print_and_log("Batch size: %d (Base: %d, Stage mult: %s, File lengths mult: %s, GPUs mult: %d) | GAM: %d -> (%d)" % (
            self.global_batch, base, mult, int(file_lengths_bs_mult * 100) / 100, self.world, self.gam, self.global_batch * self.gam),
            save_to_file=self.dataset_output)
        data_mult = max(1, min(4, int(self.global_batch / base)))
        self.EPOCH_AVG_SPAN = max(1, int(20 / data_mult))
        self.print_and_log("Data multiplier: %d" % data_mult, save_to_file=self.dataset_output)

        if stage >= 2 and not self.synthetic_data and not self.loader_factory:
            # xva_train.py:473-474.  Rank 0 decides (one answer for all ranks: a rank that saw the directory appear must not skip the
            # barriers inside); extract_durations publishes the directory only when it is complete.
            if self._from_rank0(not os.path.exists(self.dataset_input + "/durs_text") if self.rank == 0 else None):
                self.extract_durations()
        self.train_loader = self._make_loader(stage, data_mult)
        if len(self.train_loader) < self.gam:
            # start_new_epoch() drops a partial accumulation like the reference (xva_train.py:737-741): with fewer batches per epoch than
            # GAM no optimizer step would ever happen (the reference livelocks there) — accumulate over one whole epoch instead
            self.gam = max(1, len(self.train_loader))
            self.print_and_log("GAM capped to %d (batches per epoch)" % self.gam, save_to_file=self.dataset_output)
        n_lines = getattr(self.train_loader, "actual_num_lines", None) or len(self.train_loader) * self.global_batch
        self.num_iters = max(1, len(self.train_loader) // self.gam)
        self.target_delta = self.get_target_delta(n_lines, stage)
        if getattr(self, "target_delta_override", None) is not None:
            self.target_delta = float(self.target_delta_override)
        self.target_patience, self.target_patience_count = 3, 0
        self.graphs_json["stages"][str(stage)]["target_delta"] = self.target_delta
        ranges = E.trainable_ranges(stage)
        self.active = {t[0] for t in self.model._table if any(b <= t[1] < e for b, e in ranges)}
        if stage == 2:
            self.active = {n for n in self.active if not n.startswith("energy_emb")}
        if stage == 1:   # only the aligner and the symbol embedding are in the stage-1 graph (every other grad is None in the reference)
            self.active = {t[0] for t in self.model._table if (t[0].startswith("attention.") and "attn_proj" not in t[0]) or t[0] == "encoder.word_emb.weight"}
        self.active_ranges = sorted((t[1], t[1] + t[2]) for t in self.model._table if t[0] in self.active)
        self.sync = None
        if self.world > 1:
            from .dp import GradSync
            self.sync = GradSync(self.eng, self.model.flat.data, self.grads, self.world)
        self.epoch = start_epoch
        self.dataloader_iterator = iter(self.train_loader)
        self.epoch_frames_per_sec, self.avg_frames_s, self.last_loss = [], [], None
        self.start_new_epoch()
        self.print_and_log("Starting training.")
        self.is_init = True

    def start_new_epoch(self):
        self.avg_loss_per_epoch += [0.0]
        self.epoch_frames_per_sec += [0.0]
        self.accumulated_steps, self.iter_loss, self.iter_num_frames = 0, 0.0, 0
        self.iter_start_time = None
        self.iter_losses = []
        self.epoch_iter = 0

    # ---- xva_train.py:757-911 ----
    # The host never waits for the step it has just issued: the 9 floats a micro-batch reports (8 losses + its frame count) go to a pinned
    # buffer behind an event and are READ after the next micro-batch has been enqueued — the reference's NaN check therefore acts one micro-batch late (it
    # already acts after the fact, xva_train.py:825-832), and the parameters are protected meanwhile by the optimizer's on-device skip of non-finite steps.
    async def iteration(self):
        if not self.is_init:
            await self.init()
        try:
            batch = next(self.dataloader_iterator)
        except StopIteration:
            self._drain_pending()
            self.finish_epoch()
            self.start_new_epoch()
            self.dataloader_iterator = iter(self.train_loader)
            batch = next(self.dataloader_iterator)
        stage = int(self.model.training_stage)
        if self.accumulated_steps == 0:
            self.total_iter += 1
            self.epoch_iter += 1
            if self.iter_start_time is None:
                self.iter_start_time = time.perf_counter()
            adjust_learning_rate(self.total_iter, self.optimizer, self.learning_rate, self.warmup_steps)
            self.grads.zero_()
            if getattr(self, "_next_loss_scale", None):         # a new loss scale starts with an accumulation: never two scales inside one gradient sum
                self.eng.set_loss_scale(self._next_loss_scale)
                self._next_loss_scale = None
        b = batch if isinstance(batch, E.DeviceBatch) else E.DeviceBatch.from_dict(batch, self.model.flat.device)
        if self.eng.compute == 2 and self.eng.auto_loss_scale:      # one scale for all micro-batches of an accumulation: fixed from the first batch's geometry, then dynamic
            self.eng.set_loss_scale(self.eng._choose_loss_scale(b))
            self._loss_scale_cap = self.eng.loss_scale * 16
        flat = self.model.flat.data
        last = (self.accumulated_steps + 1) % self.gam == 0
        if stage == 1:
            if b.attn_prior is None:
                raise ValueError("training stage 1 needs `attn_prior` in the batch (TTSCollate, data_function.py:600-609)")
            al_loss, self._last_durs, _, _ = self.eng.align_forward(flat, b.text, b.in_lens, b.mel_tgt, b.mel_lens, b.attn_prior, want_maps=False)
            self.eng.align_backward(flat, self.grads, 1.0 / (self.gam * self.world))
            if self.world > 1:
                import torch.distributed as dist
                dist.all_reduce(al_loss)
                al_loss = al_loss / self.world
                if last:                                   # only the aligner + symbol embedding carry gradients in stage 1: a few MB
                    for rb, re_ in self.active_ranges:
                        dist.all_reduce(self.grads[rb:re_])
            losses = torch.cat([al_loss * self.attn_loss_scale, torch.zeros(7, device=flat.device)])
        elif self.sync is None:
            losses = self.eng.fwd_loss_bwd(flat, self.grads, b, stage, grad_scale=1.0 / self.gam)
        else:
            losses = self.sync.fwd_loss_bwd(b, stage, grad_scale=1.0 / self.gam, sync=last)
        self.accumulated_steps += 1
        rec = self._report_slot()
        rec["dev"][:8].copy_(losses.detach()[:8])
        rec["dev"][8] = b.mel_lens.sum() if b.mel_lens is not None else 0
        rec.update(stage=stage, last=last, total_iter=self.total_iter, epoch=self.epoch)
        if last:
            self.optimizer.step(self.grads, self.active, max_grad_norm=self.grad_clip_thresh, inv_scale=self.eng.grad_inv_scale if stage != 1 else 1.0)
            rec["dev"][9] = self.optimizer.skipped                                          # read with the losses, one micro-batch late
            self.accumulated_steps = 0
        rec["host"].copy_(rec["dev"], non_blocking=True)
        rec["event"].record()                              # behind the optimizer step when there is one: the report's arrival is the step's completion
        if last:
            if self.max_iterations and self.total_iter >= self.max_iterations:
                self.running = False
        prev, self._pending = getattr(self, "_pending", None), rec
        if prev is not None:
            self._account(prev)
        if not self.running or self.JUST_FINISHED_STAGE or self.END_OF_TRAINING:
            self._drain_pending()

    def _report_slot(self):
        """two (device, pinned) pairs of 10 floats (8 losses, the frame count, the optimizer's skipped flag) used alternately: the micro-batch in flight and the one being read"""
        if not getattr(self, "_slots", None):
            dev = self.model.flat.device
            self._slots = [{"dev": torch.zeros(10, device=dev), "host": torch.zeros(10).pin_memory(), "event": torch.cuda.Event()} for _ in range(2)]
            self._slot_i = 0
        self._slot_i ^= 1
        return self._slots[self._slot_i]

    def _drain_pending(self):
        prev, self._pending = getattr(self, "_pending", None), None
        if prev is not None:
            self._account(prev)

    def _account(self, rec):
        """the bookkeeping of one micro-batch, from its (by now finished) 9-float report: xva_train.py:815-832 (the stage's stopping signal, the NaN skip),
        :864-867 (frames / wall time of the optimizer step) and the log line"""
        rec["event"].synchronize()
        t_done = time.perf_counter()                      # the micro-batch (and its optimizer step) has finished by now: the meter's clock
        host = rec["host"].clone()
        stage = rec["stage"]
        mel_loss, dur_loss, pitch_loss = float(host[1]), float(host[2]), float(host[3])
        reduced = {1: float(host[0]), 2: float(host[0]), 3: pitch_loss * self.pitch_predictor_loss_scale, 4: mel_loss}[stage]
        if np.isnan(reduced):                             # xva_train.py:825-832 (the loss is global: every rank takes this branch together)
            self.print_and_log("loss is NaN", save_to_file=self.dataset_output)
            self.grads.zero_()                            # whatever has been accumulated since (the optimizer skipped a step with these gradients on its own)
            self.accumulated_steps, self.iter_loss, self.iter_num_frames = 0, 0.0, 0
            self.nan_streak = getattr(self, "nan_streak", 0) + 1
            if self.nan_streak >= 100:              # the reference would skip micro-batches forever; a diverged model is reported instead
                raise FloatingPointError("FastPitch stage %d: the loss was NaN for 100 consecutive micro-batches" % stage)
            return
        self.nan_streak = 0
        self.iter_loss += reduced / self.gam
        self.iter_num_frames += int(host[8])
        if rec["last"]:
            if self.eng.compute == 2 and stage != 1:
                self._update_loss_scale(float(host[9]))
            iter_time = t_done - self.iter_start_time
            fps = self.iter_num_frames * self.world / iter_time
            self.epoch_frames_per_sec[-1] += fps
            self.avg_frames_s.append(fps)
            self.all_frames_s = getattr(self, "all_frames_s", []) + [fps]       # never reset (bench.py --trainer-leg)
            self.avg_loss_per_epoch[-1] += self.iter_loss
            self.iter_losses.append(self.iter_loss)
            self.training_log_live_line = "Stage: %d | Epoch: %d | iter: %d/%d -> %d | loss: %.6f | frames/s %d | Target: %.6f    " % (
                stage, rec["epoch"], (rec["total_iter"] + 1) % self.num_iters, self.num_iters, rec["total_iter"], self.iter_loss, int(fps), self.target_delta)
            self.print_and_log(save_to_file=self.dataset_output)
            self.iter_loss, self.iter_num_frames = 0.0, 0
            self.iter_start_time = t_done

    # ---- torch.cuda.amp.GradScaler.update (xva_train.py:350,856-859) for the fp16-operand mode ----
    def _update_loss_scale(self, skipped):
        """The optimizer skipped a step with a non-finite gradient on the device (csrc/optim.hip): halve the scale; 2 000 good steps in a row: double it, up to
        the geometry-derived ceiling.  `skipped` travels with the micro-batch's report (no extra synchronisation); the new scale takes effect with the next
        accumulation that STARTS after the report (iteration()), so one gradient sum never mixes two scales."""
        cur = getattr(self, "_next_loss_scale", None) or self.eng.loss_scale      # (the report is one micro-batch late: a change may already be pending)
        if skipped != 0.0:
            self._next_loss_scale = max(1.0, cur / 2)
            self._good_steps = 0
            self.print_and_log("non-finite gradient: step skipped, loss scale -> %g" % self._next_loss_scale, save_to_file=self.dataset_output)
        else:
            self._good_steps = getattr(self, "_good_steps", 0) + 1
            if self._good_steps >= 2000 and cur < self._loss_scale_cap:
                self._next_loss_scale = cur * 2
                self._good_steps = 0

    # ---- xva_train.py:915-977 ----
    def finish_epoch(self):
        stage = int(self.model.training_stage)
        self.epoch += 1
        self.avg_loss_per_epoch[-1] = self._global_mean(self.avg_loss_per_epoch[-1] / max(1, self.epoch_iter))
        self.iter_start_time = None
        losses = self.avg_loss_per_epoch
        deltas = [(losses[i - 1] - losses[i]) / losses[i - 1] for i in range(1, len(losses)) if losses[i - 1]]
        avg_loss = float(np.mean(self.iter_losses)) if self.iter_losses else None
        delta_avg = float(np.mean(deltas[-self.EPOCH_AVG_SPAN:])) if len(deltas) >= 2 else None
        frames_s = float(np.mean(self.avg_frames_s)) if self.avg_frames_s else 0.0
        fpath = "%s/FastPitch_checkpoint_%d_%d.pt" % (self.dataset_output, self.epoch, self.total_iter)
        self.save_checkpoint(force_save=False, frames_s=frames_s, total_iter=self.total_iter, avg_loss=avg_loss, loss_delta=delta_avg,
                             avg_loss_per_epoch=self.avg_loss_per_epoch, fpath=fpath)
        g = self.graphs_json["stages"][str(stage)]
        g["loss"].append([self.total_iter, losses[-1]])
        self._save_graphs()
        if self.iter_losses:
            if self.last_loss and delta_avg is not None:
                g["loss_delta"].append([self.total_iter, delta_avg])
                self._save_graphs()
                min_epochs = 20 if stage == 2 else 1                              # xva_train.py:954
                if len(deltas) >= min_epochs and delta_avg <= self.target_delta:
                    self.target_patience_count += 1
                    if self.target_patience_count >= self.target_patience:
                        self._finish_stage(stage, frames_s, avg_loss, delta_avg, fpath)
                else:
                    self.target_patience_count = 0
            self.last_loss = avg_loss
        self.avg_frames_s = []

    def _finish_stage(self, stage, frames_s, avg_loss, delta_avg, fpath):
        """xva_train.py:956-970: the NEXT stage and an empty loss history go into the regular checkpoint (so the next trainer starts
        there with a clean stopping history) and into Stage_{n}_DONE_*, then the bare raise handleTrainer turns into the next stage."""
        fpath_stage = "%s/Stage_%d_DONE_FastPitch_checkpoint_%d_%d.pt" % (self.dataset_output, stage, self.epoch, self.total_iter)
        if stage == 4:
            self.END_OF_TRAINING = True
        self.JUST_FINISHED_STAGE = True
        self.stage_finished = stage
        self.model.training_stage = torch.tensor(stage + 1)
        self.avg_loss_per_epoch = []
        it = self.total_iter if stage + 1 == 4 else self.start_iterations
        self.save_checkpoint(force_save=True, frames_s=frames_s, total_iter=it, avg_loss=avg_loss, loss_delta=delta_avg,
                             avg_loss_per_epoch=self.avg_loss_per_epoch, fpath=fpath)
        self.save_checkpoint(force_save=True, frames_s=frames_s, total_iter=it, avg_loss=avg_loss, loss_delta=delta_avg,
                             avg_loss_per_epoch=self.avg_loss_per_epoch, fpath=fpath_stage, doPrintLog=False)
        self.running = False
        self._barrier()      # rank 0's re-written checkpoint (next stage) is complete before any rank's next trainer looks for it
        raise RuntimeError("stage %d finished" % stage)     # the reference signals stage completion by raising (xva_train.py:970)

    # ---- xva_train.py:979-1052 ----
    def save_checkpoint(self, force_save=False, frames_s=0, total_iter=0, avg_loss=None, loss_delta=None, avg_loss_per_epoch=[], fpath="out.pt",
                        doPrintLog=True):
        if self.rank != 0:
            return
        intermediate = self.epochs_per_checkpoint > 0 and self.epoch % self.epochs_per_checkpoint == 0
        if not intermediate and not force_save:
            return
        old = sorted([f for f in os.listdir(self.dataset_output) if f.startswith("FastPitch_checkpoint_")], key=sort_fp)
        for ck in old[:-2] if len(old) > 2 else []:
            os.remove(self.dataset_output + "/" + ck)
        sd = self.model.state_dict()
        checkpoint = {"epoch": self.epoch, "iteration": total_iter, "avg_loss_per_epoch": list(avg_loss_per_epoch),
                      "training_stage": self.model.training_stage, "state_dict": sd, "optimizer": self.optimizer.state_dict()}
        _atomic_save(checkpoint, fpath)      # never a half-written file under the final name (other ranks / a resumed run read it)
        _atomic_save({k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()}, "%s/%s.pt" % (self.dataset_output, self.dataset_id))
        with open("%s/%s.json" % (self.dataset_output, self.dataset_id), "w+") as f:
            json.dump({"version": "2.0", "modelVersion": "2.0", "modelType": "FastPitch1.1", "author": "", "lang": "en",
                       "games": [{"gameId": "other", "voiceId": self.dataset_id, "voiceName": self.dataset_output.split("/")[-1],
                                  "resemblyzer": [], "gender": "male"}]}, f, indent=4)
        if doPrintLog:
            line = "Stage: %d | Epoch: %d | %s~%d_%d.pt | frames/s: %d" % (int(self.model.training_stage), self.epoch,
                                                                          self.dataset_output.split("/")[-1], self.epoch, self.total_iter, int(frames_s))
            if avg_loss is not None:
                line += " | Loss: %.5f" % avg_loss
            if loss_delta is not None:
                line += " | Delta: %.5f" % loss_delta
            self.print_and_log(line + " | Target: %.5f      " % self.target_delta, save_to_file=self.dataset_output)

    # ---- xva_train.py:1054-1081 ----
    def load_checkpoint(self, filepath):
        self.print_and_log("Loading model and optimizer state from %s" % filepath, save_to_file=self.dataset_output)
        try:
            checkpoint = torch.load(filepath, map_location="cpu", weights_only=False)
        except Exception:
            self.print_and_log("Failed to load the checkpoint! Full error message: %s" % traceback.format_exc(), save_to_file=self.dataset_output)
            raise
        sd = checkpoint["state_dict"] if "state_dict" in checkpoint else checkpoint
        sd = {k.replace("module.", ""): (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
        try:
            self.model.load_state_dict(sd)
            self.optimizer.load_state_dict(checkpoint["optimizer"])
        except Exception:
            self.print_and_log("========== OPTIM NOT LOADED ==========", save_to_file=self.dataset_output)
        full = isinstance(checkpoint, dict) and "state_dict" in checkpoint
        epoch = checkpoint.get("epoch", 0) + 1 if full else 1
        total_iter = checkpoint.get("iteration", 0) if full else 0
        stage = checkpoint.get("training_stage", 1) if full else 1
        return int(stage), epoch, total_iter, list(checkpoint.get("avg_loss_per_epoch", [])) if full else []

    @staticmethod
    def last_checkpoint(output):
        """xva_train.py:1239-1250: newest FastPitch_checkpoint_* by numeric epoch."""
        if not output or not os.path.isdir(output):
            return None
        saved = sorted([f for f in os.listdir(output) if f.startswith("FastPitch_checkpoint_")], key=sort_fp)
        return output + "/" + saved[-1] if saved else None

    # ---- xva_train.py:1120-1168 ----
    def extract_durations(self, batch_size=8):
        """Run the aligner (training stage 1 graph: ConvAttention + monotonic alignment search, all on the GPU) over the whole dataset
        and write each clip's hard durations to `{dataset}/durs_text/{name}.npy` (and `durs_arpabet/` when the text encoder provides an
        ARPAbet mode), the files `TTSDataset.get_durs` (data_function.py:371-382) loads in stages 2-4.  The reference does this with
        batch size 1 and a CPU MAS; the durations of an item do not depend on its batch."""
        from ..data import FastPitchFileLoader
        self.print_and_log("Extracting durations from alignments (text)...", save_to_file=self.dataset_output)
        final_dir = self.dataset_input + "/durs_text"
        if self.rank != 0:            # rank 0 extracts; the others wait until the directory has been published
            self._barrier()
            return
        # written under a temporary name and renamed when every file is there: the directory's existence is the reference's (and this
        # trainer's) "durations are extracted" test, so it must never exist half-filled
        out_dirs = [final_dir + ".partial"]
        import shutil
        shutil.rmtree(out_dirs[0], ignore_errors=True)
        os.makedirs(out_dirs[0])
        ld = FastPitchFileLoader(self.dataset_input, batch_size, 1, self.device, shuffle=False)
        flat = self.model.flat.data
        n = len(ld.items)
        with torch.no_grad():
            for s in range(0, n, batch_size):
                idx = list(range(s, min(n, s + batch_size)))
                b = ld.collate([ld.item(i) for i in idx], 1)
                _, durs, _, _ = self.eng.align_forward(flat, b.text, b.in_lens, b.mel_tgt, b.mel_lens, b.attn_prior, want_maps=False)
                durs, order, lens = durs.cpu().numpy(), b.order.cpu().numpy(), b.in_lens.cpu().numpy()
                for r, i in enumerate(order):
                    name = ld.items[idx[i]][0]
                    for d in out_dirs:
                        np.save("%s/%s.npy" % (d, name), durs[r, :lens[r]].astype(np.float32))
                self.training_log_live_line = "\r%d/%d " % (min(n, s + batch_size), n)
                self.print_and_log(save_to_file=self.dataset_output)
        self.training_log_live_line = ""
        del ld                       # its clip cache is not the training loader's
        os.replace(out_dirs[0], final_dir)
        self._barrier()
        torch.cuda.empty_cache()
