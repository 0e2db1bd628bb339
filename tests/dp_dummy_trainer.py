"""A stand-in trainer for tests/test_dp_launch_cpu.py: the trainer protocol (handleTrainer / start / pause / flags) around a loop whose
"iteration" is one gloo all-reduce on the CPU, so the rank-group launcher (xva-trainer_amd/dp_launch.py) can be exercised without a GPU —
process spawn, websocket relay, pause / resume / stop, error propagation, the agreed stop iteration."""
import os
import time

import torch

from xva_trainer_amd import dp_launch
from xva_trainer_amd.dp_common import RankMixin

KEY = "dummy"


class DummyTrainer(RankMixin):
    def __init__(self, websocket):
        self._rank_env()
        self.websocket, self.gpus = websocket, [0]
        self.running = self.is_init = self.END_OF_TRAINING = self.JUST_FINISHED_STAGE = False
        self.it, self.sum = 0, 0.0

    def pause(self, websocket=None):
        self.request_stop()

    async def start(self, data, gpus=None, resume=False):
        self._begin_run()
        self.running = True
        if not resume:
            self.data = data
        while self.running and not self.END_OF_TRAINING:
            await self.iteration()
            self._sync_stop()
        with open(os.path.join(self.data["output_path"], "rank%d.txt" % self.rank), "a") as f:
            f.write("%d %.1f\n" % (self.it, self.sum))

    async def iteration(self):
        import torch.distributed as dist
        if not self.is_init:
            self.device = self._init_distributed()
            self.is_init = True
            await self.websocket.send("Set stage to: 1 ")
            await self.websocket.send("rank %d of %d on %s" % (self.rank, self.world, self.device))
        t = torch.tensor([float(self.rank + 1)])
        dist.all_reduce(t)
        self.sum += float(t.item())
        self.it += 1
        time.sleep(float(self.data.get("sleep", 0.01)))
        if self.data.get("fail_rank") == self.rank and self.it == self.data.get("fail_at"):
            raise ValueError("boom on rank %d" % self.rank)
        if self.data.get("oom_batch") and int(self.data["batch_size"]) >= int(self.data["oom_batch"]) and self.rank == 1:
            raise RuntimeError("HIP out of memory. Tried to allocate")
        if self.it >= int(self.data["iters"]):
            self.END_OF_TRAINING = True
            self.running = False


async def handleTrainer(models_manager, data, websocket, gpus, resume=False):
    if dp_launch.wants_rank_group(KEY, models_manager, gpus, resume):
        return await dp_launch.handle_trainer(KEY, models_manager, data, websocket, gpus, resume, worker_module="dp_dummy_trainer")
    if not resume:
        models_manager.models_bank[KEY] = DummyTrainer(websocket)
    trainer = models_manager.models_bank[KEY]
    await trainer.start(data, gpus=gpus, resume=resume)
    if trainer.END_OF_TRAINING:
        await websocket.send("Finished training\n")
        del models_manager.models_bank[KEY]
        return "done"
    return None
