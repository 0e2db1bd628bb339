"""What the three trainers share once there is more than one rank: process-group set-up from the launcher's environment, the host-side
agreements that keep every rank's control flow identical (which checkpoint is the newest, when to pause) and the test / bench options a
`data` dict may carry.

The reference runs multi-GPU inside ONE process (nn.DataParallel, python/fastpitch1_1/xva_train.py:48-53,465-466; python/xvapitch/xva_train.py:
77-82,427-428), so `pause()` flipping `running` on the websocket thread (server.py:173-181) stops the one training loop there is.  With one process
per GPU a rank that stops an iteration earlier than its peers leaves them inside a collective forever: `pause()` therefore only REQUESTS the stop and the
ranks agree on the iteration after which they all leave the loop (`RankMixin._sync_stop`)."""
import os

import torch


def dp_backend():
    """`nccl` (= RCCL over xGMI) unless XVA_DP_BACKEND says otherwise.  `gloo` exists for the tests: two ranks can then share one device, which RCCL
    refuses (tests/test_dp2_gpu.py, tests/test_trainers_dp_gpu.py)."""
    return os.environ.get("XVA_DP_BACKEND", "nccl")


def init_process_group(world):
    """Initialise torch.distributed from the launcher's env (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT — set by
    `python -m torch.distributed.run` or by dp_launch.RankGroup) when nobody has yet; returns this rank's device."""
    import torch.distributed as dist
    if not dist.is_available():
        raise RuntimeError("WORLD_SIZE=%d but torch.distributed is not available" % world)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    if not dist.is_initialized():
        for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            if k not in os.environ:
                raise RuntimeError("WORLD_SIZE=%d but %s is not set: start the trainer with gpus=[0..N-1] through handleTrainer (dp_launch.RankGroup "
                                   "spawns the ranks) or under `python -m torch.distributed.run`" % (world, k))
        backend = dp_backend()
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    if dist.get_world_size() != world:
        raise RuntimeError("process group size %d != WORLD_SIZE %d" % (dist.get_world_size(), world))
    return dev


_CTL = {}


def control_group():
    """A gloo group next to the RCCL one for host-side scalars (pause agreement, epoch-loss means, rank 0's view of the output directory): a CPU
    collective neither touches a stream nor costs a device sync.  Created collectively the first time (every rank calls this in init())."""
    import torch.distributed as dist
    world = dist.group.WORLD
    if _CTL.get("of") is not world:                  # a process group that was torn down and re-created (tests) gets a fresh control group
        _CTL["of"] = world
        _CTL["g"] = dist.new_group(backend="gloo") if dist.get_backend() != "gloo" else world
    return _CTL["g"]


class RankMixin:
    """rank / world from the environment + the cross-rank agreements.  Expects `self.device` once init() has run."""

    def _rank_env(self):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._stop_req, self._stop_work, self._stop_flag = False, None, None

    def _begin_run(self):
        """Top of start(): a stop request left over from the previous run (a pause forwarded while the ranks were already parked, or after the agreement
        was reached) must not end the next run two iterations in; a pending agreement is waited for, never abandoned."""
        if getattr(self, "_stop_work", None) is not None:
            try:
                self._stop_work.wait()
            except Exception:
                pass
        self._stop_req, self._stop_work, self._stop_flag = False, None, None
        self._runs = getattr(self, "_runs", 0) + 1            # dp_launch: "exists, not running, never ran" = still starting (a pause must wait for it)

    def _init_distributed(self):
        """One process per GPU.  WORLD_SIZE > 1: join (or create) the process group.  A gpus list with several entries in a process that is not
        a rank worker never reaches this point through handleTrainer (dp_launch spawns the ranks); a direct trainer.start(gpus=[0, 1]) is refused —
        there is no single-process nn.DataParallel mode here."""
        if self.world > 1:
            dev = init_process_group(self.world)
            control_group()
            return dev
        if self.gpus is not None and len(self.gpus) > 1:
            raise NotImplementedError("gpus=%s in one process: the MI355X path is one process per GPU — call handleTrainer(models_manager, data, websocket, "
                                      "gpus=%s) (it spawns %d rank workers, dp_launch.RankGroup) instead of trainer.start()" % (self.gpus, self.gpus, len(self.gpus)))
        return torch.device("cuda", int(self.gpus[0]) if self.gpus else 0)

    def _barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            if dist.is_initialized():
                dist.barrier()

    def _from_rank0(self, obj):
        """rank 0's value of a small host object on every rank (decisions that must not differ between ranks: which checkpoint is the newest,
        whether the durations still have to be extracted)."""
        if self.world == 1:
            return obj
        import torch.distributed as dist
        box = [obj]
        dist.broadcast_object_list(box, src=0, group=control_group())
        return box[0]

    def _global_mean(self, value):
        """mean over the DP ranks of a host scalar (keeps every rank's stopping / NaN decisions identical)."""
        if self.world == 1:
            return value
        import torch.distributed as dist
        t = torch.tensor([value], dtype=torch.float64)
        dist.all_reduce(t, group=control_group())
        return float(t.item()) / self.world

    def request_stop(self):
        """pause(): single process — leave the loop after the current iteration, like the reference.  Rank worker — ask; `_sync_stop` decides."""
        if self.world > 1:                  # also during init(): a rank that left on its own would strand its peers in their first collective
            self._stop_req = True
        else:
            self.running = False

    def _sync_stop(self):
        """Called by every rank after every iteration.  The MAX over ranks of "somebody was asked to stop", issued asynchronously after iteration i on
        the gloo group, is read after iteration i + 1: no host stall, one iteration of latency, and every rank leaves the loop at the same iteration."""
        if self.world == 1:
            return
        import torch.distributed as dist
        if self._stop_work is not None:
            self._stop_work.wait()
            self._stop_work = None
            if int(self._stop_flag.item()):
                self.running = False
                self._stop_req = False
                return
        if not self.running:            # every rank got here by the same decision (max_iterations, end of training): nothing to agree on
            return
        self._stop_flag = torch.tensor([1 if self._stop_req else 0], dtype=torch.int32)
        self._stop_work = dist.all_reduce(self._stop_flag, op=dist.ReduceOp.MAX, group=control_group(), async_op=True)


def trainer_options(data):
    """`data["trainer_options"]`: a JSON-able dict for tests and benchmarks (not in the reference; `max_iterations` / `synthetic_data` are its
    older siblings): compute ("bf16" | "fp32"), p_dropout, allow_random_init, model_kwargs, world_invariant_noise, target_delta (overrides the
    stopping threshold(s) so that a test can reach a stage transition in a few epochs), prefetch (false: the loaders run on the training thread).  A rank worker cannot be handed
    Python objects (loader factories, patched attributes), so everything a multi-rank test needs to set travels here."""
    opts = data.get("trainer_options") or {}
    unknown = set(opts) - {"compute", "p_dropout", "allow_random_init", "model_kwargs", "world_invariant_noise", "target_delta", "prefetch"}
    if unknown:
        raise ValueError("unknown trainer_options: %s" % sorted(unknown))
    return opts
