"""Multi-GPU through the reference's own entry point: `handleTrainer(models_manager, data, websocket, gpus=[0, 1, ...])`, called ONCE in the
process server.py owns (server.py:171,211-227), trains on every listed GPU.

The reference wraps its model in nn.DataParallel inside that one process (python/fastpitch1_1/xva_train.py:409,465-466; python/xvapitch/
xva_train.py:427-428).  The MI355X path is one process per GPU over RCCL, so the server process becomes the PARENT of N rank workers:

    server.py thread ── handleTrainer(gpus=[0..N-1]) ── RankGroup (in models_bank[key], what server.py pauses / deletes)
                                                          ├─ rank 0 worker: its own ModelsManager + trainer on gpus[0]; ws / training.log / graphs.json / checkpoints
                                                          ├─ rank 1 worker: trainer on gpus[1]
                                                          └─ ...                      (RANK, LOCAL_RANK = gpus[i], WORLD_SIZE, MASTER_ADDR = 127.0.0.1, MASTER_PORT)

* each worker runs the package's ordinary `handleTrainer(..., gpus=[gpus[rank]])` — stage-to-stage recursion, checkpoint resolution and the
  trainer protocol are the single-GPU code; only rank 0 writes files (the trainers' `rank == 0` guards) and only rank 0's websocket lines are relayed;
* the parent holds no device context: all of HBM belongs to the workers;
* `pause()` (server.py:173-181) is forwarded to every worker; the trainers agree on the iteration they leave their loops at (dp_common.RankMixin.
  _sync_stop) and stay alive, so `resume` (handleTrainer(resume=True), server.py:171) continues them in place; deleting the RankGroup from
  `models_bank` (server.py's "stop") ends the workers;
* a worker's exception travels back with its traceback and is re-raised in the parent as RuntimeError — server.py turns it into
  `TRAINING_ERROR:<tb>` (server.py:232-234); the other ranks, stranded in a collective, are terminated; an out-of-memory error restarts the whole group
  with the base batch size - 3 for FastPitch, mirroring python/fastpitch1_1/xva_train.py:131-145;
* the value handleTrainer returns (None | "move to hifi" | "done") is rank 0's.

Messages are JSON lines: parent -> worker on the worker's stdin ({"cmd": "pause" | "resume" | "stop"}), worker -> parent on a dedicated pipe
(XVA_DP_MSG_FD; stdout stays free for logging): {"t": "ws" | "result" | "error", ...}."""
import asyncio
import json
import os
import queue
import socket
import subprocess
import sys
import tempfile
import threading
import time
import traceback

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRAINER_MODULES = {"fastpitch1_1": "xva_trainer_amd.fastpitch.xva_train", "hifigan": "xva_trainer_amd.hifigan.xva_train",
                   "xvapitch": "xva_trainer_amd.xvapitch.xva_train"}


def in_rank_worker():
    return int(os.environ.get("WORLD_SIZE", "1")) > 1


def wants_rank_group(key, models_manager, gpus, resume):
    """True when this handleTrainer call is the server process's and has to fan out: several GPUs asked for and we are not already a rank, or a
    resume of a group that is parked in models_bank."""
    if in_rank_worker():
        return False
    if resume:
        return isinstance(models_manager.models_bank.get(key), RankGroup)
    return gpus is not None and len(gpus) > 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _is_oom(text):
    t = text.lower()
    return "out of memory" in t or "alloc_conf" in t


def _read_messages(q, rank, fd):
    with os.fdopen(fd, "r") as f:
        for line in f:
            line = line.strip()
            if line:
                try:
                    q.put((rank, json.loads(line)))
                except ValueError:
                    pass
    q.put((rank, {"t": "eof"}))


class AlreadyRunning(Exception):
    """start() on a group whose workers are training: the caller leaves models_bank alone (the reference's trainer.start() just returns)."""


class RankGroup(object):
    """What `models_bank[key]` holds in the server process while N rank workers train: the trainer surface server.py touches (`pause`, `start`,
    `running`, deletion) mapped onto the workers."""

    def __init__(self, key, logger, PROD, gpus, models_manager, websocket=None, worker_module=None):
        self.key, self.logger, self.PROD, self.gpus, self.models_manager, self.websocket = key, logger, PROD, list(gpus), models_manager, websocket
        self.worker_module = worker_module or TRAINER_MODULES[key]
        self.world = len(self.gpus)
        self.running = self.is_init = False
        self.JUST_FINISHED_STAGE = self.END_OF_TRAINING = False
        self.isReady, self.ckpt_path, self.model = True, "None", None
        self.procs, self.q, self._readers, self._logs = [], queue.Queue(), [], []
        self.parked = False                  # workers alive and idle after a pause
        self.closed = False                  # close() ran: a pump still waiting on the queue gives up
        self.dataset_output = None

    # ---- process management ----
    def _spawn(self, data):
        port = _free_port()
        self.closed = False
        self._tmp = tempfile.mkdtemp(prefix="xva_dp_")
        cfg = json.dumps({"key": self.key, "module": self.worker_module, "data": data, "PROD": bool(self.PROD)})
        for rank, gpu in enumerate(self.gpus):
            rfd, wfd = os.pipe()
            env = dict(os.environ)
            env.update(RANK=str(rank), LOCAL_RANK=str(int(gpu)), WORLD_SIZE=str(self.world), LOCAL_WORLD_SIZE=str(self.world), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), XVA_DP_MSG_FD=str(wfd), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"),
                       PYTHONPATH=_ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), PYTHONUNBUFFERED="1")
            log = open(os.path.join(self._tmp, "rank%d.stderr" % rank), "w+")
            p = subprocess.Popen([sys.executable, "-c", "from xva_trainer_amd.dp_launch import worker_main; worker_main()"], stdin=subprocess.PIPE,
                                 stderr=log, env=env, pass_fds=(wfd,), cwd=os.getcwd(), text=True)
            os.close(wfd)
            p.stdin.write(cfg + "\n")
            p.stdin.flush()
            t = threading.Thread(target=_read_messages, args=(self.q, rank, rfd), daemon=True)     # holds the queue, not the group: deleting the group must end it
            t.start()
            self.procs.append(p)
            self._readers.append(t)
            self._logs.append(log)

    def _send(self, cmd):
        for p in self.procs:
            if p.poll() is None:
                try:
                    p.stdin.write(json.dumps(cmd) + "\n")
                    p.stdin.flush()
                except (BrokenPipeError, OSError, ValueError):
                    pass

    def _stderr_tail(self, rank, n=3000):
        try:
            f = self._logs[rank]
            f.flush()
            f.seek(0)
            return f.read()[-n:]
        except Exception:
            return ""

    def close(self, kill=False):
        """End the workers ("stop" in server.py deletes the trainer from models_bank; so does every terminal path here).  kill: a rank failed — its
        peers are stranded in a collective and will not answer a command."""
        procs, self.procs = self.procs, []
        self.closed = True
        for p in procs:
            if p.poll() is None:
                try:
                    p.stdin.write(json.dumps({"cmd": "stop"}) + "\n")
                    p.stdin.flush()
                    p.stdin.close()
                except Exception:
                    pass
        for p in procs:
            try:
                p.wait(timeout=0.2 if kill else 20)
            except Exception:
                p.terminate()
                try:
                    p.wait(timeout=5)
                except Exception:
                    p.kill()
        for f in self._logs:
            try:
                f.close()
            except Exception:
                pass
        tmp, self._tmp = getattr(self, "_tmp", None), None
        if tmp:                                    # the workers' stderr files (quoted in the error when a rank dies without a message)
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
        self._logs, self.parked, self.running = [], False, False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- the trainer surface ----
    def pause(self, websocket=None):
        self._send({"cmd": "pause"})

    async def start(self, data, gpus=None, resume=False):
        """Returns (result, parked): rank 0's handleTrainer value, and whether the workers stay alive (paused).  Raises RuntimeError carrying a
        worker's traceback."""
        if self.running:
            raise AlreadyRunning(self.key)
        if resume:
            if not self.parked:
                raise RuntimeError("resume: the rank workers of '%s' are gone" % self.key)
            self._send({"cmd": "resume"})
        else:
            self.dataset_output = data["output_path"] + "/" + data["dataset_path"].split("/")[-1]
            self._spawn(data)
        self.running, self.parked = True, False
        try:
            return await self._pump()
        except RuntimeError:
            raise
        except BaseException:                      # anything else leaving the pump (cancellation, a relay error): no stranded workers
            self.close(kill=True)
            raise
        finally:
            self.running = False

    async def _pump(self):
        loop = asyncio.get_event_loop()
        results, eof = {}, set()

        def get():
            try:
                return self.q.get(True, 0.25)
            except queue.Empty:
                return None
        while len(results) < self.world:
            item = await loop.run_in_executor(None, get)
            if self.closed:                        # close() from another task (stop, a replaced group): nothing more will arrive
                raise RuntimeError("the rank workers of '%s' were closed while training" % self.key)
            if item is None:
                for r, p in enumerate(self.procs):
                    if p.poll() is not None and r not in results and r in eof:
                        self._fail("rank %d exited with code %s before reporting a result\n%s" % (r, p.returncode, self._stderr_tail(r)))
                continue
            rank, msg = item
            kind = msg.get("t")
            if kind == "ws":
                if rank == 0 and self.websocket is not None:
                    try:
                        await self.websocket.send(msg["m"])
                    except Exception as e:         # a closed / stale UI socket must not strand the workers: log, drop, keep pumping
                        self.websocket = None
                        try:
                            self.logger.info("dp_launch: websocket relay failed (%s: %s); messages dropped until the next resume" % (type(e).__name__, e))
                        except Exception:
                            pass
            elif kind == "result":
                results[rank] = msg
            elif kind == "error":
                self._fail("rank %d:\n%s" % (rank, msg.get("tb", "")), oom=bool(msg.get("oom")))
            elif kind == "eof":
                eof.add(rank)
        r0 = results[0]
        if any(results[r]["paused"] != r0["paused"] for r in results):
            self._fail("the ranks disagree on whether training is paused or finished: %s" % {r: results[r]["paused"] for r in results})
        self.parked = bool(r0["paused"])
        if not self.parked:
            self.close()
        return r0.get("v"), self.parked

    def _fail(self, text, oom=False):
        self.parked = False
        self.close(kill=True)
        err = RuntimeError(text)
        err.xva_oom = oom or _is_oom(text)
        raise err


async def handle_trainer(key, models_manager, data, websocket, gpus, resume=False, worker_module=None):
    """The body of the three packages' handleTrainer when it runs in the server process with several GPUs (or resumes such a run).
    worker_module: the module whose handleTrainer the workers run (default: the package registered for `key`; tests pass a stand-in)."""
    bank = models_manager.models_bank
    live = bank.get(key)
    if isinstance(live, RankGroup) and live.running:
        return None                                # a second start / resume while training: the reference's trainer.start() just returns
    if resume:
        group = bank[key]
        group.websocket = websocket                # the UI may have reconnected since the pause
    else:
        old = bank.get(key)
        if isinstance(old, RankGroup):
            old.close()
        group = RankGroup(key, models_manager.logger, models_manager.PROD, gpus, models_manager, websocket=websocket, worker_module=worker_module)
        bank[key] = group
    try:
        result, parked = await group.start(data, gpus=gpus, resume=resume)
    except AlreadyRunning:
        return None
    except RuntimeError as e:
        bank.pop(key, None)
        if getattr(e, "xva_oom", False) and key == "fastpitch1_1" and int(data.get("batch_size", 0)) > 3:      # python/fastpitch1_1/xva_train.py:131-145
            data = dict(data, batch_size=int(data["batch_size"]) - 3)
            return await handle_trainer(key, models_manager, data, websocket, gpus, worker_module=worker_module)
        raise
    if not parked:
        bank.pop(key, None)
        if result == "move to hifi":
            bank[key] = "move to hifi"                                                                         # python/fastpitch1_1/xva_train.py:160-161
    return result


# ------------------------------------------------------------------------------------------------ the rank worker
class _WorkerWS(object):
    def __init__(self, emit):
        self._emit = emit

    async def send(self, msg):
        self._emit({"t": "ws", "m": str(msg)})


def worker_main():
    """One rank: the package's handleTrainer on gpus[rank], commands from the parent on stdin, messages to it on XVA_DP_MSG_FD."""
    import importlib
    import logging
    out = os.fdopen(int(os.environ["XVA_DP_MSG_FD"]), "w", buffering=1)
    lock = threading.Lock()

    def emit(obj):
        with lock:
            out.write(json.dumps(obj) + "\n")
            out.flush()
    cfg = json.loads(sys.stdin.readline())
    key, data = cfg["key"], cfg["data"]
    local = int(os.environ["LOCAL_RANK"])
    cmds, state = queue.Queue(), {"mm": None, "stop": False}

    def pause_trainer():
        mm = state["mm"]
        tr = mm.models_bank.get(key) if mm is not None else None
        if tr is not None and hasattr(tr, "pause") and getattr(tr, "running", True):
            tr.pause()
        elif (tr is None or getattr(tr, "_runs", 0) == 0) and not state.get("early"):
            # a pause that beats the trainer into models_bank (the worker is still importing / building it), or that lands between its insertion and
            # its first start() (RankMixin._begin_run counts the runs), is remembered and applied once the trainer runs — dropped silently, the UI
            # would show "paused" over ranks that keep training.  A pause on a PARKED trainer (it has run before) stays a no-op: remembered, it would
            # end the next resume two iterations in.
            state["early"] = True

            def later():
                # The request belongs to the trainer's FIRST run.  If that run has already ended when this thread looks (a peer rank's watcher asked first and
                # the ranks agreed to stop within two iterations — the watchers poll 0.1 s apart), the request is spent: kept alive, it would stop the NEXT run
                # (a resume) a few iterations in.
                for _ in range(3000):
                    t = state["mm"].models_bank.get(key) if state["mm"] is not None else None
                    if t is not None and hasattr(t, "pause"):
                        runs = getattr(t, "_runs", None)
                        if getattr(t, "running", False):
                            if runs is None or runs <= 1:
                                t.pause()
                            break
                        if runs is not None and runs >= 1:
                            break
                    time.sleep(0.1)
                state["early"] = False
            threading.Thread(target=later, daemon=True).start()

    def reader():
        for line in sys.stdin:
            try:
                cmd = json.loads(line)
            except ValueError:
                continue
            if cmd.get("cmd") == "pause":
                pause_trainer()
            else:
                if cmd.get("cmd") == "stop":
                    state["stop"] = True
                    pause_trainer()
                cmds.put(cmd)
        state["stop"] = True                       # EOF: the parent is gone
        pause_trainer()
        cmds.put({"cmd": "stop"})
    threading.Thread(target=reader, daemon=True).start()
    code = 0
    try:
        import torch
        from xva_trainer_amd.models_manager import ModelsManager
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        mod = importlib.import_module(cfg["module"])
        mm = state["mm"] = ModelsManager(logging.getLogger("xva.rank%s" % os.environ.get("RANK")), cfg["PROD"], "cuda:%d" % local)
        ws = _WorkerWS(emit)
        resume = False
        while not state["stop"]:
            result = asyncio.run(mod.handleTrainer(mm, data, ws, gpus=[local], resume=resume))
            tr = mm.models_bank.get(key)
            paused = tr is not None and not isinstance(tr, str) and not getattr(tr, "END_OF_TRAINING", False) and not state["stop"]
            emit({"t": "result", "v": result, "paused": bool(paused)})
            if not paused:
                break
            cmd = cmds.get()
            if cmd.get("cmd") != "resume":
                break
            resume = True
    except BaseException:
        tb = traceback.format_exc()
        emit({"t": "error", "tb": tb, "oom": _is_oom(tb)})
        code = 1
    try:
        import torch.distributed as dist
        if code == 0 and dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:
        pass
    out.close()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)          # no interpreter teardown: a peer that died mid-collective must not hang this rank in a destructor
