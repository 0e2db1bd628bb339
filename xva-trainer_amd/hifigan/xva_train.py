"""HiFiTrainer / handleTrainer — the trainer protocol of python/hifigan/xva_train.py:50-128,131-700 on the HIP engine.

Kept from the reference: `async handleTrainer(models_manager, data, websocket, gpus, resume)` -> "done"; class
`HiFiTrainer(logger, PROD, gpus, models_manager, websocket)` with `async start / init / iteration`, `pause`, `finish_epoch`,
`output_checkpoint`; data keys (dataset_path, output_path, hifigan_checkpoint, num_workers, batch_size,
epochs_per_checkpoint); batch = int(batch_size * 1.4) (xva_train.py:228); config_v1.json hyper-parameters; AdamW x2 with
ExponentialLR(0.999) per epoch; checkpoints `hifi/g_{steps:08d}` = {'generator': sd}, `hifi/do_{steps:08d}` = {'mpd', 'msd',
'optim_g', 'optim_d', 'steps', 'epoch', 'avg_loss_per_epoch', 'ckpts_finetuned'} (keep last 2) and `{name}.hg.pt`; the
"Stage 5 | Epoch ... | Mel loss ... | its/s" log line; early stop when the mean of the last 25 epoch mel-error deltas is
<= 1e-4 after >= 25 epochs; ws string "Finished training HiFi-GAN\\n".
Changed on purpose: mels are computed on the GPU (HIP mel) instead of on the CPU in the dataset; one iteration is a fixed
sequence of C calls (hifigan/step.py)."""
import glob
import json
import os
import time

import numpy as np
import torch

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


class HiFiTrainer(object):
    def __init__(self, logger, PROD, gpus, models_manager, websocket=None, compute="bf16", loader_factory=None):
        self.logger, self.PROD, self.gpus, self.models_manager, self.websocket = logger, PROD, gpus, models_manager, websocket
        self.compute, self.loader_factory = compute, loader_factory
        self.ckpt_path, self.isReady = "None", True
        self.running = self.is_init = False
        self.JUST_FINISHED_STAGE = self.END_OF_TRAINING = False
        self.training_log, self.training_log_live_line = [], ""
        self.h = dict(CONFIG_V1)
        self.EPOCH_AVG_SPAN, self.target_delta = 25, 1e-4
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.dataset_output = None

    def print_and_log(self, line=None, end="\n", flush=False, save_to_file=None):
        if line is None:
            line = self.training_log_live_line
        else:
            self.training_log.append(line)
        if self.rank == 0 and save_to_file is not None:
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
        self.running = False

    async def start(self, data, gpus=None, resume=False):
        if self.running:
            return
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
        while self.running and not self.END_OF_TRAINING:
            await self.iteration()

    async def init(self):
        dev = torch.device("cuda", self.gpus[0] if self.world == 1 else int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        torch.manual_seed(self.h["seed"] + self.rank)
        self.device = dev
        self.h["batch_size"] = int(self.batch_size * 1.4)                                   # xva_train.py:228
        self.core = HifiganStep(dev, self.compute, lr=self.h["learning_rate"], betas=(self.h["adam_b1"], self.h["adam_b2"]))
        self.training_steps, self.training_epoch, self.ckpts_finetuned = 0, 0, 0
        self.avg_loss_per_epoch = []
        cp_g = scan_checkpoint(self.dataset_output + "/hifi", "g_") or self.hifigan_checkpoint
        cp_do = scan_checkpoint(self.dataset_output + "/hifi", "do_")
        if cp_g and os.path.exists(str(cp_g)):
            sd = torch.load(cp_g, map_location="cpu", weights_only=False)
            self.core.load_state_dicts(generator=sd["generator"])
        if cp_do:
            sd = torch.load(cp_do, map_location="cpu", weights_only=False)
            self.core.load_state_dicts(mpd=sd["mpd"], msd=sd["msd"])
            self.training_steps, self.training_epoch = sd["steps"] + 1, sd["epoch"]
            self.ckpts_finetuned = sd.get("ckpts_finetuned", 0)
        loader = self.loader_factory(self) if self.loader_factory else None
        if loader is None:
            from ..data import SyntheticHifiLoader
            loader = SyntheticHifiLoader(self.h["batch_size"], segment=self.h["segment_size"], seed=self.h["seed"] + 100 * self.rank)
        self.train_loader = loader
        self.dataloader_iterator = iter(loader)
        self.start_new_epoch()
        self.is_init = True

    def start_new_epoch(self):
        self.epoch_start_time = time.time()
        self.avg_loss_per_epoch += [0.0]
        self.epoch_iter = 0

    async def iteration(self):
        if not self.is_init:
            await self.init()
        try:
            wav = next(self.dataloader_iterator)
        except StopIteration:
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
        mel_error = float(out["loss_mel"].item()) / 45.0                               # the one host sync per iteration
        self.epoch_iter += 1
        self.avg_loss_per_epoch[-1] += int(mel_error * 1000) / 1000
        s_per_b = max(time.time() - start_b, 1e-9)
        its_p_s = int(100 * h["batch_size"] * self.world / s_per_b) / 100
        self.training_log_live_line = "Stage 5 | Epoch: %d | It: %d/%d (%d) | Mel loss: %.3f | its/s: %s " % (
            self.training_epoch + 1, (self.training_steps + 1) % max(1, len(self.train_loader)), len(self.train_loader), self.training_steps + 1,
            mel_error, its_p_s)
        self.print_and_log(save_to_file=self.dataset_output)
        self.training_steps += 1
        if self.max_iterations and self.training_steps >= self.max_iterations:
            self.running = False

    def output_checkpoint(self):
        if self.rank != 0 or self.training_epoch % self.epochs_per_checkpoint != 0:
            return
        sds = self.core.state_dicts()
        cpu = lambda sd: {k: v.cpu() for k, v in sd.items()}
        hifi = self.dataset_output + "/hifi"
        torch.save({"generator": cpu(sds["generator"])}, "%s/g_%08d" % (hifi, self.training_steps))
        self.ckpts_finetuned += 1
        torch.save({"mpd": cpu(sds["mpd"]), "msd": cpu(sds["msd"]), "optim_g": self.core.optim_g.param_groups, "optim_d": self.core.optim_d.param_groups,
                    "steps": self.training_steps, "epoch": self.training_epoch, "avg_loss_per_epoch": [], "ckpts_finetuned": self.ckpts_finetuned},
                   "%s/do_%08d" % (hifi, self.training_steps))
        for prefix in ("do_", "g_"):
            for ck in sorted([f for f in os.listdir(hifi) if f.startswith(prefix)], key=sort_ckpt)[:-2]:
                os.remove(hifi + "/" + ck)
        torch.save({"generator": cpu(sds["generator"])}, "%s/%s.hg.pt" % (self.dataset_output, self.dataset_output.split("/")[-1]))
        self.print_and_log("Stage 5 |Epoch: %d | It: %d | g_%08d | Mel loss: %s" % (self.training_epoch, self.training_steps, self.training_steps,
                                                                                   self.avg_loss_per_epoch[-1] / max(1, self.epoch_iter)),
                           save_to_file=self.dataset_output)

    def finish_epoch(self):
        for opt in (self.core.optim_g, self.core.optim_d):                                # ExponentialLR(gamma=lr_decay) per epoch
            opt.param_groups[0]["lr"] *= self.h["lr_decay"]
        self.training_epoch += 1
        self.avg_loss_per_epoch[-1] /= max(1, self.epoch_iter)
        self.output_checkpoint()
        losses = self.avg_loss_per_epoch
        deltas = [(a - b) / a for a, b in zip(losses[:-1], losses[1:]) if a]
        if len(deltas) >= self.EPOCH_AVG_SPAN and len(losses) >= 25:
            if all(float(np.mean(deltas[max(0, i - self.EPOCH_AVG_SPAN):i])) <= self.target_delta for i in range(len(deltas) - 2, len(deltas) + 1)):
                self.END_OF_TRAINING = True
                self.running = False
                raise RuntimeError("HiFi-GAN training finished")
