"""CPU, no GPU: the rank-group launcher behind handleTrainer(gpus=[0, 1]) (xva-trainer_amd/dp_launch.py) with a stand-in trainer whose
iteration is a gloo all-reduce (tests/dp_dummy_trainer.py).  What server.py relies on (server.py:171-234): ONE call in ONE process trains on
every listed GPU, rank 0's websocket lines come back, pause() parks the run on the same iteration on every rank, resume continues it in place,
deleting the bank entry ends the workers, a rank's exception comes back as a RuntimeError carrying its traceback, an out-of-memory error restarts
FastPitch-style with the base batch size - 3.  The three real trainers run through the same launcher in tests/test_trainers_dp_gpu.py."""
import asyncio
import logging
import os
import sys
import threading
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


class _WS:
    def __init__(self):
        self.sent = []

    async def send(self, msg):
        self.sent.append(msg)


@pytest.fixture()
def env(monkeypatch):
    monkeypatch.setenv("XVA_DP_BACKEND", "gloo")
    monkeypatch.setenv("PYTHONPATH", HERE + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    sys.path.insert(0, HERE)
    yield
    sys.path.remove(HERE)


def _mm():
    from xva_trainer_amd.models_manager import ModelsManager
    return ModelsManager(logging.getLogger("t"), False, "cpu")


def _data(tmp_path, **kw):
    out = str(tmp_path / "out")
    os.makedirs(out, exist_ok=True)
    return dict({"dataset_path": str(tmp_path / "in" / "voice"), "output_path": out, "batch_size": 8, "iters": 5}, **kw)


def _ranks(data):
    return [[tuple(float(v) for v in line.split()) for line in open(os.path.join(data["output_path"], "rank%d.txt" % r)).read().split("\n") if line] for r in (0, 1)]


@pytest.mark.timeout(300)
def test_one_call_runs_every_rank_and_relays_rank0(env, tmp_path):
    import dp_dummy_trainer as T
    mm, ws, data = _mm(), _WS(), _data(tmp_path)
    assert asyncio.run(T.handleTrainer(mm, data, ws, gpus=[0, 1])) == "done"
    assert "dummy" not in mm.models_bank
    assert ws.sent == ["Set stage to: 1 ", "rank 0 of 2 on cuda:0", "Finished training\n"]          # rank 1's lines stay with rank 1
    r0, r1 = _ranks(data)
    assert r0 == r1 == [(5.0, 15.0)]                                                               # 5 iterations, all-reduce of 1 + 2 each


@pytest.mark.timeout(300)
def test_pause_parks_every_rank_on_the_same_iteration_then_resume_and_stop(env, tmp_path):
    import dp_dummy_trainer as T
    from xva_trainer_amd.dp_launch import RankGroup
    mm, ws, data = _mm(), _WS(), _data(tmp_path, iters=400, sleep=0.02)

    def pause_later():
        while not isinstance(mm.models_bank.get("dummy"), RankGroup) or len(ws.sent) < 2:
            time.sleep(0.05)
        time.sleep(0.5)
        mm.models_bank["dummy"].pause()                       # what server.py's websocket thread does (server.py:173-181)
    th = threading.Thread(target=pause_later)
    th.start()
    assert asyncio.run(T.handleTrainer(mm, data, ws, gpus=[0, 1])) is None
    th.join()
    group = mm.models_bank["dummy"]
    assert isinstance(group, RankGroup) and group.parked and not group.running and all(p.poll() is None for p in group.procs)
    r0, r1 = _ranks(data)
    assert r0 == r1 and 1 <= r0[0][0] < 400, (r0, r1)         # both ranks left their loops after the SAME iteration
    # resume (server.py:171 with resume=True) continues in place: same processes, the counter goes on
    pids = [p.pid for p in group.procs]
    data2 = dict(data)
    th = threading.Thread(target=lambda: (time.sleep(0.6), mm.models_bank["dummy"].pause()))
    th.start()
    assert asyncio.run(T.handleTrainer(mm, data2, ws, gpus=[0, 1], resume=True)) is None
    th.join()
    assert [p.pid for p in mm.models_bank["dummy"].procs] == pids
    r0b, r1b = _ranks(data)
    assert r0b == r1b and len(r0b) == 2 and r0b[1][0] > r0b[0][0]
    # stop = delete the bank entry (server.py:182-188): the workers end
    procs = list(group.procs)
    del mm.models_bank["dummy"]
    del group
    deadline = time.time() + 30
    while any(p.poll() is None for p in procs) and time.time() < deadline:
        time.sleep(0.1)
    assert all(p.poll() is not None for p in procs)


@pytest.mark.timeout(300)
def test_a_rank_error_comes_back_with_its_traceback_and_ends_the_group(env, tmp_path):
    import dp_dummy_trainer as T
    mm, ws, data = _mm(), _WS(), _data(tmp_path, iters=1000, fail_rank=1, fail_at=3)
    with pytest.raises(RuntimeError) as ei:
        asyncio.run(T.handleTrainer(mm, data, ws, gpus=[0, 1]))
    msg = str(ei.value)
    assert "rank 1" in msg and "ValueError: boom on rank 1" in msg and "Traceback" in msg           # server.py sends TRAINING_ERROR:<this>
    assert "dummy" not in mm.models_bank


@pytest.mark.timeout(300)
def test_out_of_memory_restarts_the_group_with_a_smaller_base_batch(env, tmp_path, monkeypatch):
    """python/fastpitch1_1/xva_train.py:131-145 at group level: no rank retries on its own (its peers would be stranded)."""
    import dp_dummy_trainer as T
    from xva_trainer_amd import dp_launch
    mm, ws, data = _mm(), _WS(), _data(tmp_path, iters=3, oom_batch=7, batch_size=8)
    real = dp_launch.handle_trainer

    async def as_fastpitch(key, *a, **kw):                     # the back-off is FastPitch's; run the stand-in under that key
        return await real("fastpitch1_1", *a, **kw)
    monkeypatch.setattr(dp_launch, "handle_trainer", as_fastpitch)
    monkeypatch.setattr(T, "KEY", "fastpitch1_1")
    assert asyncio.run(T.handleTrainer(mm, data, ws, gpus=[0, 1])) == "done"
    r0, r1 = _ranks(data)
    assert r0 == r1 == [(3.0, 9.0)]                                                                # 8 -> 5 (< 7): the second group trained


@pytest.mark.timeout(300)
def test_second_start_while_training_leaves_the_group_alone_and_a_dead_socket_does_not_strand_it(env, tmp_path):
    """a `startTraining` / `resume` that arrives while the ranks train must neither remove the live group from models_bank nor close it
    (the reference's trainer.start() just returns); a websocket that raises is logged and dropped; a pause forwarded while the ranks are already
    parked does not end the next run early."""
    import dp_dummy_trainer as T
    from xva_trainer_amd.dp_launch import RankGroup

    class _DeadWS(_WS):
        async def send(self, msg):
            raise ConnectionError("socket closed")
    mm, ws, data = _mm(), _DeadWS(), _data(tmp_path, iters=100000, sleep=0.01)
    second = {}

    def poke():
        while not isinstance(mm.models_bank.get("dummy"), RankGroup) or not mm.models_bank["dummy"].running:
            time.sleep(0.05)
        time.sleep(1.0)
        group = mm.models_bank["dummy"]
        second["start"] = asyncio.run(T.handleTrainer(mm, data, _WS(), gpus=[0, 1]))                    # a second start ...
        second["resume"] = asyncio.run(T.handleTrainer(mm, data, _WS(), gpus=[0, 1], resume=True))      # ... and a stray resume
        second["same"] = mm.models_bank.get("dummy") is group and group.running and all(p.poll() is None for p in group.procs)
        time.sleep(0.3)
        group.pause()
    th = threading.Thread(target=poke)
    th.start()
    assert asyncio.run(T.handleTrainer(mm, data, ws, gpus=[0, 1])) is None                              # the relay error did not end the run
    th.join()
    assert second == {"start": None, "resume": None, "same": True}
    group = mm.models_bank["dummy"]
    assert group.parked and all(p.poll() is None for p in group.procs)
    n0 = _ranks(data)[0][0][0]
    group.pause()                                                # a double click: pause while parked
    time.sleep(0.3)
    ws2 = _WS()
    th = threading.Thread(target=lambda: (time.sleep(1.5), mm.models_bank["dummy"].pause()))
    th.start()
    assert asyncio.run(T.handleTrainer(mm, data, ws2, gpus=[0, 1], resume=True)) is None
    th.join()
    r0, r1 = _ranks(data)
    assert r0 == r1 and r0[1][0] - n0 > 20, (r0, n0)            # ran for the 1.5 s, not for the two iterations a stale request would leave
    assert mm.models_bank["dummy"].websocket is ws2             # the resumed run relays to the socket that asked for it
    mm.models_bank["dummy"].close()
    del mm.models_bank["dummy"]


def test_single_gpu_and_rank_workers_do_not_fan_out(monkeypatch):
    from xva_trainer_amd import dp_launch
    mm = _mm()
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert not dp_launch.wants_rank_group("hifigan", mm, [0], False) and not dp_launch.wants_rank_group("hifigan", mm, None, False)
    assert dp_launch.wants_rank_group("hifigan", mm, [0, 1], False)
    assert not dp_launch.wants_rank_group("hifigan", mm, [0, 1], True)        # resume without a parked group: the ordinary single-process path
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert not dp_launch.wants_rank_group("hifigan", mm, [0, 1], False)       # inside a rank worker
