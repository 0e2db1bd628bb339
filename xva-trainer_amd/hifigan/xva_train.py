"""HiFiTrainer / handleTrainer — the trainer protocol of python/hifigan/xva_train.py:50-128,131-700 on the HIP engine.

Kept from the reference: `async handleTrainer(models_manager, data, websocket, gpus, resume)` -> "done"; class
`HiFiTrainer(logger, PROD, gpus, models_manager, websocket)` with `async start / init / iteration`, `pause`, `finish_epoch`,
`output_checkpoint`; data keys (dataset_path, output_path, hifigan_checkpoint, num_workers, batch_size,
epochs_per_checkpoint); batch = int(batch_size * 1.4) (xva_train.py:228); config_v1.json hyper-parameters; AdamW x2 with
ExponentialLR(0.999) per epoch; checkpoints `hifi/g_{steps:08d}` = {'generator': sd}, `hifi/do_{steps:08d}` = {'mpd', 'msd',
'optim_g', 'optim_d', 'steps', 'epoch', 'avg_loss_per_epoch', 'ckpts_finetuned'} (keep last 2) and `{name}.hg.pt`; the
"Stage 5 | Epoch ... | Mel loss ... | its/s" log line; early stop when the mean of the last 25 epoch mel-error deltas is
<= 1e-4 after >= 25 epochs; ws string "Finished training HiFi-GAN\\n".
Resume restores both AdamW states (torch's own state_dict format over the reference parameter order) and applies
ExponentialLR(last_epoch=epoch)'s initial step; like the reference the trainer never trains from scratch (raises without a
g_ / do_ pair; "[male]" / "[female]" resolve to the pretrained directories).
Changed on purpose: crops, peak normalisation and both mels are computed on the GPU (xva-trainer_amd/data.py, HIP mel) instead of on
the CPU in the dataset; one iteration is a fixed sequence of C calls (hifigan/step.py); multi-GPU is one process per GPU with
bucketed, overlapped RCCL gradient exchange (the reference defines DataParallel here but never wraps, xva_train.py:40), started by
handleTrainer(..., gpus=[0..N-1]) itself (xva-trainer_amd/dp_launch.py)."""
import glob
import json
import os
import time

import numpy as np
import torch

from .. import dp_launch
from ..dp_common import RankMixin, trainer_options
from ..mel import mel_spectrogram
from .step import HifiganStep

CONFIG_V1 = {"resblock": "1", "batch_size": 46, "learning_rate": 0.0002, "adam_b1": 0.8, "adam_b2": 0.99, "lr_decay": 0.999, "seed": 1234,
             "upsample_rates": [8, 8, 2, 2], "upsample_kernel_sizes": [16, 16, 4, 4], "upsample_initial_channel": 512,
             "resblock_kernel_sizes": [3, 7, 11], "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "segment_size": 8192,
             "num_mels": 80, "num_freq": 1025, "n_fft": 1024, "hop_size": 256, "win_size": 1024, "sampling_rate": 22050, "fmin": 0, "fmax": 8000,
             "fmax_for_loss": None}


def sort_ckpt(x):
    return int(x.split("_")[-1])


def scan_checkpoint(cp_dir, prefix):
    """python/hifigan/utils.py:57-62."""
    cp_list = glob.glob(os.path.join(cp_dir, prefix + "????????"))
    return sorted(cp_list)[-1] if cp_list else None


async def handleTrainer(models_manager, data, websocket, gpus, resume=False):
    """python/hifigan/xva_train.py:50-128.  gpus=[0, 1, ...] in the server process: one rank worker per GPU (dp_launch)."""
    if dp_launch.wants_rank_group("hifigan", models_manager, gpus, resume):
        return await dp_launch.handle_trainer("hifigan", models_manager, data, websocket, gpus, resume)
    if not resume:
        models_manager.sync_init_model("hifigan", websocket=websocket, gpus=gpus)
        trainer = models_manager.models_bank["hifigan"]
        dataset_id = data["dataset_path"].split("/")[-1]
        trainer.init_logs(dataset_output=data["output_path"] + "/" + dataset_id)
    else:
        trainer = models_manager.models_bank["hifigan"]
    try:
        await trainer.start(data, gpus=gpus, resume=resume)
    except KeyboardInterrupt:
        trainer.running = False
        raise
    except RuntimeError:
        if trainer.END_OF_TRAINING:
            trainer.print_and_log("Finished training HiFi-GAN\n", save_to_file=trainer.dataset_output)
            if trainer.websocket is not None:
                await trainer.websocket.send("Finished training HiFi-GAN\n")
            del models_manager.models_bank["hifigan"]
            return "done"
        raise
    return None


class HiFiTrainer(RankMixin):
    def __init__(self, logger, PROD, gpus, models_manager, websocket=None, compute="bf16", loader_factory=None):
        self.logger, self.PROD, self.gpus, self.models_manager, self.websocket = logger, PROD, gpus, models_manager, websocket
        self.compute, self.loader_factory = compute, loader_factory
        self.ckpt_path, self.isReady = "None", True
        self.running = self.is_init = False
        self.JUST_FINISHED_STAGE = self.END_OF_TRAINING = False
        self.training_log, self.training_log_live_line = [], ""
        self.h = dict(CONFIG_V1)
        self.EPOCH_AVG_SPAN, self.target_delta = 25, 1e-4
        self._rank_env()
        self.dataset_output = None
        self.allow_random_init = False              # tests / benchmarks only: the reference refuses to train from scratch (xva_train.py:276-277)
        root = "./resources/app" if PROD else "."
        self.pretrained_ckpt_male = root + "/python/hifigan/pretrained_models/male"
        self.pretrained_ckpt_female = root + "/python/hifigan/pretrained_models/female"

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
        os.makedirs(dataset_output + "/hifi", exist_ok=True)
        gpath = dataset_output + "/graphs.json"
        self.graphs_json = json.load(open(gpath)) if os.path.exists(gpath) else \
            {"stages": {str(s): {"loss": [], "loss_delta": [], "target_delta": None} for s in range(1, 6)}}

    def pause(self, websocket=None):
        self.request_stop()

    async def start(self, data, gpus=None, resume=False):
        if self.running:
            return
        self._begin_run()
        self.running = True
        if not resume:
            if gpus is not None:
                self.gpus = gpus
            self.dataset_input = data["dataset_path"]
            self.dataset_id = self.dataset_input.split("/")[-1]
            self.dataset_output = data["output_path"] + "/" + self.dataset_id
            os.makedirs(self.dataset_output + "/hifi", exist_ok=True)
            self.hifigan_checkpoint = data.get("hifigan_checkpoint")
            self.workers = data.get("num_workers", 0)
            self.batch_size = int(data["batch_size"])
            self.epochs_per_checkpoint = int(data.get("epochs_per_checkpoint", 1))
            self.max_iterations = data.get("max_iterations")
            self.synthetic_data = bool(data.get("synthetic_data", False))      # explicit opt-in (bench / tests); never a silent fallback
            opts = trainer_options(data)                                       # tests / bench (a rank worker cannot be handed Python objects)
            self.compute = opts.get("compute", self.compute)
            self.allow_random_init = bool(opts.get("allow_random_init", self.allow_random_init))
            self.prefetch = bool(opts.get("prefetch", os.environ.get("XVA_PREFETCH", "1") != "0"))
        while self.running and not self.END_OF_TRAINING:
            await self.iteration()
            self._sync_stop()
        if getattr(self, "is_init", False):
            self._drain_pending()              # a pause lands between two iterations: the last iteration's report is still in flight

    async def init(self):
        dev = self._init_distributed()
        torch.cuda.set_device(dev)
        torch.manual_seed(self.h["seed"] + self.rank)
        self.device = dev
        self.h["batch_size"] = int(self.batch_size * 1.4)                                   # xva_train.py:228
        self.core = HifiganStep(dev, self.compute, lr=self.h["learning_rate"], betas=(self.h["adam_b1"], self.h["adam_b2"]))
        checkpoint_path = self.dataset_output + "/hifi"
        self.print_and_log("Output checkpoints directory: %s" % checkpoint_path, save_to_file=self.dataset_output)
        self.print_and_log("Stage 5: HiFi-GAN fine-tuning", save_to_file=self.dataset_output)
        self.print_and_log("Batch size: %d (Base: %d, Stage mult: 1.5)" % (self.h["batch_size"], self.batch_size), save_to_file=self.dataset_output)
        if self.websocket is not None:
            await self.websocket.send("Set stage to: 5 ")
        self.training_steps, self.training_epoch, self.ckpts_finetuned = 0, -1, 0
        self.avg_loss_per_epoch = []
        self.target_patience, self.target_patience_count = 3, 0
        self.graphs_json["stages"]["5"]["target_delta"] = self.target_delta
        self._barrier()                    # one view of the output directory: every rank resumes from the pair rank 0 sees
        cp_g, cp_do = self._from_rank0((scan_checkpoint(checkpoint_path, "g_"), scan_checkpoint(checkpoint_path, "do_")) if self.rank == 0 else None)
        if cp_g is None:                                                                    # xva_train.py:255-264
            self.print_and_log("No existing HiFi-GAN checkpoints for this voice.", save_to_file=self.dataset_output)
            src = {"[male]": self.pretrained_ckpt_male, "[female]": self.pretrained_ckpt_female}.get(self.hifigan_checkpoint, self.hifigan_checkpoint)
            if src and os.path.isdir(str(src)):
                cp_g, cp_do = scan_checkpoint(src, "g_"), scan_checkpoint(src, "do_")
        if cp_g is None or cp_do is None:
            if not self.allow_random_init:
                raise RuntimeError("HiFi-GAN: no g_ / do_ checkpoint pair to fine-tune from (hifigan_checkpoint=%r) — the trainer never trains "
                                   "from scratch (python/hifigan/xva_train.py:276-277)" % (self.hifigan_checkpoint,))
        else:
            self.print_and_log("Loading checkpoint from: %s" % cp_g, save_to_file=self.dataset_output)
            sd_g = torch.load(cp_g, map_location="cpu", weights_only=False)
            sd_do = torch.load(cp_do, map_location="cpu", weights_only=False)
            self.core.load_state_dicts(generator=sd_g["generator"], mpd=sd_do["mpd"], msd=sd_do["msd"])
            self.training_steps, self.training_epoch = sd_do["steps"] + 1, sd_do["epoch"]
            self.ckpts_finetuned = sd_do.get("ckpts_finetuned", 0)
            self.core.optim_g.load_state_dict(sd_do["optim_g"])                             # xva_train.py:302-304
            self.core.optim_d.load_state_dict(sd_do["optim_d"])
            self.ckpt_path = cp_g
        if self.training_epoch >= 0:       # ExponentialLR(optimizer, gamma, last_epoch=epoch): its initial step() decays the restored lr once (:306-307)
            for opt in (self.core.optim_g, self.core.optim_d):
                opt.param_groups[0]["lr"] *= self.h["lr_decay"]
        self.train_loader = self._make_loader()
        self.dataloader_iterator = iter(self.train_loader)
        self.start_new_epoch()
        self.is_init = True

    def _make_loader(self):
        if self.loader_factory:
            loader = self.loader_factory(self)
            if loader is not None:
                return loader
        if self.synthetic_data:
            from ..data import SyntheticHifiLoader
            return SyntheticHifiLoader(self.h["batch_size"], segment=self.h["segment_size"], seed=self.h["seed"] + 100 * self.rank)
        if not os.path.exists(os.path.join(self.dataset_input, "metadata.csv")):
            raise FileNotFoundError("%s/metadata.csv not found: the trainer reads the reference's dataset layout (metadata.csv + wavs/); "
                                    "synthetic data needs the explicit `synthetic_data` opt-in" % self.dataset_input)
        from ..data import HifiFileLoader
        ld = HifiFileLoader(self.dataset_input, self.h["batch_size"], self.device, segment=self.h["segment_size"], seed=self.h["seed"],
                            rank=self.rank, world=self.world)
        self.print_and_log("Training items: %d" % len(ld.files), save_to_file=self.dataset_output)
        if getattr(self, "prefetch", True):       # crops of batch i + 1 are read, staged, uploaded and normalised under step i (data.Prefetcher)
            from ..data import Prefetcher
            ld = Prefetcher(ld, self.device, depth=2)
        return ld

    def start_new_epoch(self):
        self.epoch_start_time = time.time()
        self.avg_loss_per_epoch += [0.0]
        self.epoch_iter = 0

    async def iteration(self):
        """The host does not wait for the iteration it has just issued: the mel loss goes to a pinned float behind an event and is read —
        with the log line and the its/s meter — after the NEXT iteration has been enqueued (the reference reads it with .item() in place,
        python/hifigan/xva_train.py:517-528)."""
        if not self.is_init:
            await self.init()
        try:
            wav = next(self.dataloader_iterator)
        except StopIteration:
            self._drain_pending()
            self.finish_epoch()
            self.start_new_epoch()
            self.dataloader_iterator = iter(self.train_loader)
            wav = next(self.dataloader_iterator)
        start_b = time.time()
        h = self.h
        y = wav.to(self.device, non_blocking=True)
        x = mel_spectrogram(y, h["n_fft"], h["num_mels"], h["sampling_rate"], h["hop_size"], h["win_size"], h["fmin"], h["fmax"])
        y_mel = mel_spectrogram(y, h["n_fft"], h["num_mels"], h["sampling_rate"], h["hop_size"], h["win_size"], h["fmin"], h["fmax_for_loss"])
        out = self.core.train_step(x, y, y_mel)
        if not getattr(self, "_slots", None):
            self._slots = [{"host": torch.zeros(1).pin_memory(), "event": torch.cuda.Event()} for _ in range(2)]
            self._slot_i = 0
        self._slot_i ^= 1
        rec = self._slots[self._slot_i]
        rec["host"].copy_(out["loss_mel"].detach().reshape(1), non_blocking=True)
        rec["event"].record()
        rec.update(start_b=start_b, step=self.training_steps + 1, epoch=self.training_epoch + 1)
        self.training_steps += 1
        if self.max_iterations and self.training_steps >= self.max_iterations:
            self.running = False
        prev, self._pending = getattr(self, "_pending", None), rec
        if prev is not None:
            self._account(prev)
        if not self.running or self.END_OF_TRAINING:
            self._drain_pending()

    def _drain_pending(self):
        prev, self._pending = getattr(self, "_pending", None), None
        if prev is not None:
            self._account(prev)

    def _account(self, rec):
        rec["event"].synchronize()
        h = self.h
        mel_error = float(rec["host"][0]) / 45.0
        self.epoch_iter += 1
        self.avg_loss_per_epoch[-1] += int(mel_error * 1000) / 1000
        now = time.time()
        last = getattr(self, "_last_account_t", None)
        s_per_b = max(now - (last if last is not None and last > rec["start_b"] - 60.0 and self.epoch_iter > 1 else rec["start_b"]), 1e-9)   # wall time per iteration incl. the loader
        self._last_account_t = now
        its_p_s = int(100 * h["batch_size"] * self.world / s_per_b) / 100
        self.avg_samples_s = getattr(self, "avg_samples_s", [])
        self.avg_samples_s.append(h["batch_size"] * self.world * h["segment_size"] / s_per_b)
        self.training_log_live_line = "Stage 5 | Epoch: %d | It: %d/%d (%d) | Mel loss: %.3f | its/s: %s " % (
            rec["epoch"], rec["step"] % max(1, len(self.train_loader)), len(self.train_loader), rec["step"], mel_error, its_p_s)
        self.print_and_log(save_to_file=self.dataset_output)

    def output_checkpoint(self):
        """xva_train.py:570-601.  Called BEFORE the epoch loss is normalised, like the reference: the log line divides the running sum."""
        if self.rank != 0 or self.training_epoch % self.epochs_per_checkpoint != 0:
            return
        sds = self.core.state_dicts()
        cpu = lambda sd: {k: v.cpu() for k, v in sd.items()}
        hifi = self.dataset_output + "/hifi"
        torch.save({"generator": cpu(sds["generator"])}, "%s/g_%08d" % (hifi, self.training_steps))
        self.print_and_log("Stage 5 |Epoch: %d | It: %d | g_%08d | Mel loss: %s" % (self.training_epoch, self.training_steps, self.training_steps,
                                                                                   self.avg_loss_per_epoch[-1] / max(1, self.epoch_iter)),
                           save_to_file=self.dataset_output)
        self.ckpts_finetuned += 1
        torch.save({"mpd": cpu(sds["mpd"]), "msd": cpu(sds["msd"]), "optim_g": self.core.optim_g.state_dict(), "optim_d": self.core.optim_d.state_dict(),
                    "steps": self.training_steps, "epoch": self.training_epoch, "avg_loss_per_epoch": [], "ckpts_finetuned": self.ckpts_finetuned},
                   "%s/do_%08d" % (hifi, self.training_steps))
        for prefix in ("do_", "g_"):
            for ck in sorted([f for f in os.listdir(hifi) if f.startswith(prefix)], key=sort_ckpt)[:-2]:
                os.remove(hifi + "/" + ck)
        torch.save({"generator": cpu(sds["generator"])}, "%s/%s.hg.pt" % (self.dataset_output, self.dataset_output.split("/")[-1]))

    def finish_epoch(self):
        """xva_train.py:607-650."""
        for opt in (self.core.optim_g, self.core.optim_d):                                # scheduler_g / scheduler_d .step(): ExponentialLR(gamma=lr_decay)
            opt.param_groups[0]["lr"] *= self.h["lr_decay"]
        self.training_epoch += 1
        self.output_checkpoint()
        self.avg_loss_per_epoch[-1] /= max(1, self.epoch_iter)
        self.avg_loss_per_epoch[-1] = self._global_mean(self.avg_loss_per_epoch[-1])      # identical stopping decisions on every rank
        losses = self.avg_loss_per_epoch
        deltas = [(losses[i - 1] - losses[i]) / losses[i - 1] for i in range(1, len(losses)) if losses[i - 1]]
        self.graphs_json["stages"]["5"]["loss"].append([self.training_steps, losses[-1]])
        if len(deltas) >= 2:
            avg = float(np.mean(deltas[-self.EPOCH_AVG_SPAN:]))
            self.graphs_json["stages"]["5"]["loss_delta"].append([self.training_steps, avg])
            if self.rank == 0:
                with open(self.dataset_output + "/graphs.json", "w") as f:
                    json.dump(self.graphs_json, f)
            if avg <= self.target_delta and len(deltas) >= 25:
                self.target_patience_count += 1
                if self.target_patience_count >= self.target_patience:
                    self.training_log_live_line = ""
                    self.print_and_log("HiFi-GAN training finished", save_to_file=self.dataset_output)
                    self.END_OF_TRAINING = True
                    self.running = False
                    raise RuntimeError("HiFi-GAN training finished")
            else:
                self.target_patience_count = 0
