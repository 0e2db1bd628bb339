"""FastPitchTrainer / handleTrainer — the trainer protocol of python/fastpitch1_1/xva_train.py:57-176,185-1081 on the HIP
engine (training stages 1-4).

Kept from the reference, because the Electron UI / server.py consume them: module-level `async handleTrainer(models_manager,
data, websocket, gpus, resume)` returning None | "move to hifi"; class `FastPitchTrainer(logger, PROD, gpus, models_manager,
websocket)` with `async start / init / iteration`, `pause`, `finish_epoch`, `save_checkpoint`, `load_checkpoint` and the flags
`running / is_init / JUST_FINISHED_STAGE / END_OF_TRAINING`; the `data` dict keys; `training.log` + `graphs.json`; ws strings
"Set stage to: N "; checkpoint files `FastPitch_checkpoint_{epoch}_{iter}.pt` (keep last 2), `{dataset_id}.pt` (fp16
state_dict) and `{dataset_id}.json`; LAMB(lr 0.1, betas (0.9, 0.98), eps 1e-9, wd 1e-6), clip 1000, warm-up 1000,
gam = max(1, round(256 / batch)); stage freezing; loss-delta early stopping.
Changed on purpose: the step is 3 C calls (no autograd graph, no GradScaler: bf16 needs no loss scaling); training.log is
appended, not rewritten, each step; multi-GPU is one process per GPU (dp.GradSync) instead of nn.DataParallel.
"""
import json
import os
import time
import traceback

import numpy as np
import torch

from . import engine as E
from . import params as P
from .lamb import Lamb
from .loss_function import FastPitchLoss  # noqa: F401  (re-exported like the reference module)
from .model import FastPitch


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


async def handleTrainer(models_manager, data, websocket, gpus, resume=False):
    """python/fastpitch1_1/xva_train.py:57-176."""
    if not resume:
        models_manager.sync_init_model("fastpitch1_1", websocket=websocket, gpus=gpus)
    trainer = models_manager.models_bank["fastpitch1_1"]
    if not resume:
        dataset_id = data["dataset_path"].split("/")[-1]
        trainer.init_logs(dataset_output=data["output_path"] + "/" + dataset_id)
    try:
        return await trainer.start(data, gpus=gpus, resume=resume)
    except KeyboardInterrupt:
        trainer.running = False
        raise
    except RuntimeError:
        if trainer.JUST_FINISHED_STAGE:
            trainer.JUST_FINISHED_STAGE = False
            trainer.is_init = False
            finished = int(trainer.model.training_stage)
            if finished >= 4:
                trainer.print_and_log("Finished training FastPitch", save_to_file=trainer.dataset_output)
                del models_manager.models_bank["fastpitch1_1"]
                return "move to hifi"
            data = dict(data)
            data["force_stage"] = finished + 1
            del models_manager.models_bank["fastpitch1_1"]
            return await handleTrainer(models_manager, data, websocket, gpus)
        raise


class FastPitchTrainer(object):
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
        self.EPOCH_AVG_SPAN, self.target_delta = 20, 0.0
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))

    # ---- logs the UI reads from disk (xva_train.py:226-238,546-569) ----
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
        self.running = False

    # ---- xva_train.py:675-755 ----
    async def start(self, data, gpus=None, resume=False):
        if self.running:
            return
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
            self.learning_rate, self.weight_decay = 0.1, 1e-6
            self.dur_predictor_loss_scale = self.pitch_predictor_loss_scale = 0.1
            self.attn_loss_scale = 1.0                                 # xva_train.py:704
            self.warmup_steps, self.grad_clip_thresh = 1000, 1000
        while self.running and not self.JUST_FINISHED_STAGE:
            await self.iteration()

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

    async def init(self):
        dev = torch.device("cuda", self.gpus[0] if self.world == 1 else int(os.environ.get("LOCAL_RANK", "0")))
        torch.cuda.set_device(dev)
        torch.manual_seed(1234 + self.rank)
        np.random.seed(1234 + self.rank)
        self.model = FastPitch(logger=self.logger, compute=self.compute).to(dev)
        self.eng = self.model._get_engine()
        self.optimizer = Lamb(self.model.flat.data, self.model._table, lr=self.learning_rate, betas=(0.9, 0.98), eps=1e-9,
                              weight_decay=self.weight_decay)
        self.grads = torch.zeros_like(self.model.flat.data)
        self.epoch, self.total_iter, self.avg_loss_per_epoch = 1, 0, []
        stage = 1                                             # a fresh model starts with the aligner (model.py:176: training_stage = 1)
        ckpt = self.last_checkpoint(self.dataset_output) or self.checkpoint
        if ckpt and os.path.exists(str(ckpt)):
            stage, self.epoch, self.total_iter, self.avg_loss_per_epoch = self.load_checkpoint(ckpt)
        if self.force_stage:
            stage = self.force_stage
        stage = max(1, int(stage))
        self.model.training_stage = torch.tensor(stage)
        if self.websocket is not None:
            await self.websocket.send("Set stage to: %d " % stage)
        self.gam = max(1, round(256 / (self.batch_size * self.world)))
        loader = self.loader_factory(self) if self.loader_factory else None
        if loader is None:
            from ..data import SyntheticFastPitchLoader
            loader = SyntheticFastPitchLoader(self.batch_size, seed=1234 + self.rank)
        self.train_loader = loader
        self.num_iters = max(1, len(loader) // self.gam)
        self.target_delta = self.get_target_delta(len(loader) * self.batch_size, stage)
        self.graphs_json["stages"][str(stage)]["target_delta"] = self.target_delta
        ranges = E.trainable_ranges(stage)
        self.active = {t[0] for t in self.model._table if any(b <= t[1] < e for b, e in ranges)}
        if stage == 2:
            self.active = {n for n in self.active if not n.startswith("energy_emb")}
        if stage == 1:   # only the aligner and the symbol embedding are in the stage-1 graph (every other grad is None in the reference)
            self.active = {t[0] for t in self.model._table if (t[0].startswith("attention.") and "attn_proj" not in t[0]) or t[0] == "encoder.word_emb.weight"}
        self.sync = None
        if self.world > 1:
            from .dp import GradSync
            self.sync = GradSync(self.eng, self.model.flat.data, self.grads, self.world)
        self.dataloader_iterator = iter(self.train_loader)
        self.accumulated_steps, self.iter_loss, self.iter_num_frames, self.epoch_iter = 0, 0.0, 0, 0
        self.iter_start_time = None
        self.avg_loss_per_epoch.append(0.0)
        self.epoch_frames_per_sec = [0.0]
        self.avg_frames_s = []
        self.is_init = True

    # ---- xva_train.py:757-911 ----
    async def iteration(self):
        if not self.is_init:
            await self.init()
        try:
            batch = next(self.dataloader_iterator)
        except StopIteration:
            self.finish_epoch()
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
        b = E.DeviceBatch.from_dict(batch, self.model.flat.device)
        flat = self.model.flat.data
        last = (self.accumulated_steps + 1) % self.gam == 0
        if stage == 1:
            if b.attn_prior is None:
                raise ValueError("training stage 1 needs `attn_prior` in the batch (TTSCollate, data_function.py:600-609)")
            al_loss, self._last_durs, _, _ = self.eng.align_forward(flat, b.text, b.in_lens, b.mel_tgt, b.mel_lens, b.attn_prior, want_maps=False)
            self.eng.align_backward(flat, self.grads, 1.0 / (self.gam * self.world))
            if self.world > 1 and last:   # the aligner's gradients are a few MB: one plain all-reduce at the end of the accumulation
                import torch.distributed as dist
                dist.all_reduce(self.grads)
            losses = torch.cat([al_loss * self.attn_loss_scale, torch.zeros(7, device=flat.device)])
        elif self.sync is None:
            losses = self.eng.fwd_loss_bwd(flat, self.grads, b, stage, grad_scale=1.0 / self.gam)
        else:
            losses = self.sync.fwd_loss_bwd(b, stage, grad_scale=1.0 / self.gam, sync=last)
        self.accumulated_steps += 1
        self._pending = (losses, b.mel_lens if b.mel_lens is not None else None)
        host = losses.detach().cpu()                      # the one host sync per micro-batch (reference: 5 x .item())
        mel_loss, dur_loss, pitch_loss = float(host[1]), float(host[2]), float(host[3])
        reduced = {1: float(host[0]), 2: float(host[0]), 3: pitch_loss * self.pitch_predictor_loss_scale, 4: mel_loss}[stage]
        if np.isnan(reduced):
            self.print_and_log("loss is NaN", save_to_file=self.dataset_output)
            self.grads.zero_()
            self.accumulated_steps = 0
            return
        self.iter_loss += reduced / self.gam
        self.iter_num_frames += int(b.mel_lens.sum().item()) if b.mel_lens is not None else 0
        if last:
            self.optimizer.step(self.grads, self.active, max_grad_norm=self.grad_clip_thresh)
            iter_time = time.perf_counter() - self.iter_start_time
            fps = self.iter_num_frames * self.world / iter_time
            self.epoch_frames_per_sec[-1] += fps
            self.avg_frames_s.append(fps)
            self.avg_loss_per_epoch[-1] += self.iter_loss
            self.training_log_live_line = "Stage: %d | Epoch: %d | iter: %d/%d -> %d | loss: %.6f | frames/s %d | Target: %.6f    " % (
                stage, self.epoch, (self.total_iter + 1) % self.num_iters, self.num_iters, self.total_iter, self.iter_loss, int(fps), self.target_delta)
            self.print_and_log(save_to_file=self.dataset_output)
            self.accumulated_steps, self.iter_loss, self.iter_num_frames = 0, 0.0, 0
            self.iter_start_time = time.perf_counter()
            if self.max_iterations and self.total_iter >= self.max_iterations:
                self.running = False

    # ---- xva_train.py:915-977 ----
    def finish_epoch(self):
        stage = int(self.model.training_stage)
        avg = self.avg_loss_per_epoch[-1] / max(1, self.epoch_iter)
        self.avg_loss_per_epoch[-1] = avg
        deltas = [(a - b) / a for a, b in zip(self.avg_loss_per_epoch[:-1], self.avg_loss_per_epoch[1:]) if a]
        delta = float(np.mean(deltas[-self.EPOCH_AVG_SPAN:])) if deltas else None
        g = self.graphs_json["stages"][str(stage)]
        g["loss"].append([self.total_iter, avg])
        if delta is not None:
            g["loss_delta"].append([self.total_iter, delta])
        self._save_graphs()
        fpath = "%s/FastPitch_checkpoint_%d_%d.pt" % (self.dataset_output, self.epoch, self.total_iter)
        self.save_checkpoint(frames_s=self.epoch_frames_per_sec[-1] / max(1, self.epoch_iter), total_iter=self.total_iter, avg_loss=avg,
                             loss_delta=delta, avg_loss_per_epoch=self.avg_loss_per_epoch, fpath=fpath)
        finished = len(deltas) >= 3 and all(d <= self.target_delta for d in deltas[-3:]) and delta is not None and delta <= self.target_delta
        self.epoch += 1
        self.epoch_iter = 0
        self.avg_loss_per_epoch.append(0.0)
        self.epoch_frames_per_sec.append(0.0)
        if finished:
            self.save_checkpoint(force_save=True, total_iter=self.total_iter, avg_loss_per_epoch=self.avg_loss_per_epoch,
                                 fpath="%s/Stage_%d_DONE_%d_%d.pt" % (self.dataset_output, stage, self.epoch, self.total_iter))
            self.JUST_FINISHED_STAGE = True
            self.running = False
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
        checkpoint = {"epoch": self.epoch, "iteration": total_iter, "avg_loss_per_epoch": avg_loss_per_epoch,
                      "training_stage": self.model.training_stage, "state_dict": sd, "optimizer": self.optimizer.state_dict()}
        torch.save(checkpoint, fpath)
        torch.save({k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()}, "%s/%s.pt" % (self.dataset_output, self.dataset_id))
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
        epoch = checkpoint.get("epoch", 0) + 1 if isinstance(checkpoint, dict) else 1
        total_iter = checkpoint.get("iteration", 0) if isinstance(checkpoint, dict) else 0
        stage = checkpoint.get("training_stage", 1) if isinstance(checkpoint, dict) else 1
        return int(stage), epoch, total_iter, list(checkpoint.get("avg_loss_per_epoch", [])) if isinstance(checkpoint, dict) else []

    @staticmethod
    def last_checkpoint(output):
        """xva_train.py:1239-1250: newest FastPitch_checkpoint_* by numeric epoch."""
        if not output or not os.path.isdir(output):
            return None
        saved = sorted([f for f in os.listdir(output) if f.startswith("FastPitch_checkpoint_")], key=sort_fp)
        return output + "/" + saved[-1] if saved else None
