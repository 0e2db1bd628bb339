"""CPU, world_size 2 over gloo: the data-parallel semantics the GPU path implements (fastpitch/dp.py): per-rank shards +
an all-reduce of the loss numerators/denominators (GLOBAL normalisation, as the reference computes the loss on the gathered
outputs: python/fastpitch1_1/xva_train.py:788-790) + a SUM all-reduce of bucketed gradients == the single-process
full-batch gradient.  The per-rank math here is the oracle's (no GPU kernels in this container)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _global_loss_parts(ofp, out, batch):
    """numerators / denominators of FastPitchLoss stage 3 (loss_function.py:95-117), un-normalised."""
    mel_out, _, _, _, pitch_pred, pitch_tgt, energy_pred, energy_tgt, _, _, _, _, in_lens = out
    mt = batch["mel_tgt"].transpose(1, 2)
    mm = mt.ne(0).float()
    mo = torch.nn.functional.pad(mel_out, (0, 0, 0, mt.size(1) - mel_out.size(1)))
    dm = ofp.mask_from_lens(in_lens, batch["text"].size(1)).float()
    nums = torch.stack([((mo - mt) ** 2 * mm).sum(), ((pitch_tgt - pitch_pred) ** 2 * dm.unsqueeze(1)).sum(), ((energy_tgt - energy_pred) ** 2 * dm).sum()])
    dens = torch.stack([mm.sum(), dm.sum(), dm.sum()])
    return nums, dens


def _full_batch(ofp):
    """Global batch of 4 = two ragged pairs that each contain a full-length item, so every shard pads to the same lengths
    as the global batch.  (The reference's conv-FFN does not mask its inner activation — transformer.py:59-77 — so an item's
    output depends slightly on how far it is padded: re-padding per rank would change the math, not only the layout.)"""
    a, b = ofp.synth_batch(2, 10, 36, 22), ofp.synth_batch(2, 10, 36, 23)
    return {k: torch.cat([a[k], b[k]]) for k in a}


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import dp
    torch.manual_seed(0)
    sd = ofp.init_state_dict(21)
    full = _full_batch(ofp)
    shard = {k: v[rank * 2:(rank + 1) * 2] for k, v in full.items()}
    names = ofp.trainable_names(sd.keys(), 3)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    out = ofp.forward({**sd, **leaves}, shard, 3)
    nums, dens = _global_loss_parts(ofp, out, shard)
    dens_g = dens.clone()
    dist.all_reduce(dens_g)                                 # exchange 1: denominators (the 8-float acc all-reduce on the GPU path)
    loss_local = (nums / dens_g * torch.tensor([1.0, 0.1, 0.1])).sum()
    loss_local.backward()
    keys = sorted(k for k in names if leaves[k].grad is not None)
    flat = torch.cat([leaves[k].grad.reshape(-1) for k in keys])
    n = flat.numel()
    ranges = [(0, n // 3), (n // 3, n // 3), (n // 3, 2 * n // 3), (2 * n // 3, n)]    # incl. an empty bucket
    dp.allreduce_flat_buckets(flat, ranges)                 # exchange 2: SUM of gradients, bucketed
    total = loss_local.detach().clone()
    dist.all_reduce(total)
    if rank == 0:
        torch.save({"flat": flat, "keys": keys, "loss": total}, tmp)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_world2_equals_full_batch(tmp_path):
    sys.path.insert(0, ROOT)
    from oracle import fastpitch as ofp
    out_file = str(tmp_path / "dp.pt")
    port = 29500 + (os.getpid() % 1000)
    mp.spawn(_worker, args=(2, port, out_file), nprocs=2, join=True)
    res = torch.load(out_file, weights_only=False)
    sd = ofp.init_state_dict(21)
    full = _full_batch(ofp)
    names = ofp.trainable_names(sd.keys(), 3)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    loss, _ = ofp.loss(ofp.forward({**sd, **leaves}, full, 3), full, 3)
    loss.backward()
    ref = torch.cat([leaves[k].grad.reshape(-1) for k in res["keys"]])
    assert abs(res["loss"].item() - loss.item()) < 1e-4 * abs(loss.item())
    assert ((res["flat"] - ref).norm() / ref.norm()).item() < 1e-4


def test_bucket_ranges_cover_trainable_params():
    """The 13 gradient buckets of the engine tile [encoder .. proj] without overlap, in backward-completion order."""
    from xva_trainer_amd.fastpitch import dp, engine as E
    rng = dp.bucket_ranges()
    assert len(rng) == 13
    srt = sorted(rng)
    assert all(a[1] == b[0] for a, b in zip(srt, srt[1:]))
    table = E.tensor_table()
    proj_end = max(off + n for name, off, n, _, _ in table if name.startswith("proj."))
    assert srt[0][0] == 0 and srt[-1][1] >= proj_end
    assert dp.buckets_for_stage(3) == list(range(13))
    assert dp.buckets_for_stage(2) == [6, 7, 8, 9, 10, 11, 12]
    assert rng[0][1] > rng[5][0] and rng[12][0] == 0       # decoder buckets come first, the embedding bucket last


def test_hifigan_bucket_ranges_tile_the_trainable_prefix():
    """hifigan/step.py:BucketSync exchanges each flat gradient buffer in the engine's buckets: 8 discriminators / 6 generator
    ranges that tile [0, trainable) without overlap (the spectral-norm buffers behind it are never exchanged)."""
    from xva_trainer_amd.hifigan import engine as HE
    for which, nb in ((HE.G, 6), (HE.D, 8)):
        rng = HE.bucket_ranges(which)
        assert len(rng) == nb
        srt = sorted(rng)
        assert srt[0][0] == 0 and srt[-1][1] == int(HE.lib.xva_hg_trainable_floats(which))
        assert all(a[1] == b[0] for a, b in zip(srt, srt[1:]))
    d = HE.bucket_ranges(HE.D)
    assert d == sorted(d)                                   # discriminators finish in buffer order (MPD 0..4, MSD 0..2)
    g = HE.bucket_ranges(HE.G)
    assert g[0][0] > g[3][0] and g[4][0] == 0               # generator: last stage's resblocks first, conv_pre + ups last


def _mean_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xva_trainer_amd.xvapitch.train_step import allreduce_mean_
    g = torch.Generator().manual_seed(100 + rank)
    a, b, c = torch.randn(3, 4, generator=g), torch.randn(5, generator=g), torch.randn(2, 1, 3, generator=g)
    keep = [t.clone() for t in (a, b, c)]
    allreduce_mean_([a, None, b, c[:, :, :2]])                        # a None entry and a non-contiguous view (the padded posterior weight's gradient)
    torch.save({"after": [a, b, c], "before": keep}, os.path.join(tmp, "mean_%d.pt" % rank))
    dist.destroy_process_group()


def test_xvapitch_gradient_mean_world2(tmp_path):
    """xvapitch/train_step.py:allreduce_mean_ (the C5 iteration's data-parallel gradient reduction) on 2 gloo ranks: every listed tensor ends as
    the mean over ranks, in place, including a non-contiguous view; tensors outside the list (the last column of c) keep their values."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_mean_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), "mean_%d.pt" % r)) for r in (0, 1))
    for i in range(2):
        want = (r0["before"][i] + r1["before"][i]) / 2
        assert torch.allclose(r0["after"][i], want) and torch.allclose(r1["after"][i], want)
    want = (r0["before"][2] + r1["before"][2]) / 2
    for r in (r0, r1):
        assert torch.allclose(r["after"][2][:, :, :2], want[:, :, :2]) and torch.equal(r["after"][2][:, :, 2], r["before"][2][:, :, 2])


def test_bench_two_rank_dry_run_over_gloo():
    """`python bench.py --gpus 2 --dry-run-gloo`: the script re-executes itself under torch.distributed.run (127.0.0.1 rendezvous), two CPU ranks
    build their own shard (seed 1234 + rank), all-reduce the stage-3 gradient bucket ranges of the engine over gloo (each rank checks the sum),
    and rank 0 prints ONE contract line whose `value` is the whole job's frames (both shards) over the slowest rank's time."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run-gloo", "--steps", "2", "--warmup", "1"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    dp = out["dp"]            # the multi-rank line explains its exchange: the same keys the RCCL run fills from HIP events
    assert dp["rccl_ranks"] == 2 and dp["backend"] == "gloo" and "rccl_version" in dp
    assert len(dp["bucket_start_ms"]) == out["config"]["buckets"] and dp["allreduce_ms_exposed"] >= 0 and dp["backward_ms"] >= 0
    assert out["config"]["parallelism"] == "dp2" and out["config"]["global_batch"] == 64 and out["config"]["buckets"] >= 12
    per_rank = out["config"]["per_gpu_frames_per_step"]
    # whole-job aggregate: both ranks' frames (their shards differ by seed, so within a few percent of 2 x rank 0's) over the max-over-ranks time
    total = out["value"] * out["ms_per_step"] / 1e3
    assert 1.8 * per_rank < total < 2.2 * per_rank, (total, per_rank)


def _load_bench():
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    argv, sys.argv = sys.argv, ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("xva_bench_module", os.path.join(root, "bench.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod, root


def test_bench_contract_line_is_compact_and_complete():
    """the stdout line of bench.py is the contract keys + `roofline` + `cpu_baseline` + one small object per extra leg, at most
    MAX_LINE_BYTES (6 kB; the round-4 line was 25 kB and the driver could not read it back).  Built here from the full measurement committed as
    profiles/r04_final_bench.json — the same dictionary `emit` receives — so the reduction is checked without a GPU."""
    import json
    bench, root = _load_bench()
    full = json.load(open(os.path.join(root, "profiles", "r04_final_bench.json")))
    s = bench.compact_line(full)
    assert "\n" not in s and len(s) <= bench.MAX_LINE_BYTES <= 8192, len(s)
    line = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in line, k
    assert line["value"] == full["value"] and line["ms_per_step"] == full["ms_per_step"] and line["config"]["workload"] == full["config"]["workload"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "avg_launch_us", "algorithmic_gflop_per_launch", "algorithmic_mbytes_per_launch"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = line["cpu_baseline"]
    assert set(cb) >= {"value", "unit", "cores", "kind"} and cb["kind"] in ("port", "reference")
    for leg in ("hifigan", "xvapitch_c5", "fastpitch_fp32_parity", "hifigan_fp32_parity"):
        assert "ms_per_step" in line[leg] and "parity" in line[leg], leg
    assert line["hifigan"]["roofline_stack"]["frac"] == pytest.approx(full["hifigan"]["roofline_stack"]["frac"], rel=1e-4)
    # nothing long survives: no tables, notes, methods or samples
    assert not any(k in s for k in ('"by_kernel"', '"note"', '"method"', '"sample"'))
    # a pathological input still yields a line under the bound, contract keys intact
    fat = dict(full, config=dict(full["config"]), hifigan=dict(full["hifigan"], value=1.0))
    fat["hbm_kernels"] = {("k%d" % i): {"frac": 0.1, "achieved": 1.0, "avg_launch_us": 2.0} for i in range(400)}
    s2 = bench.compact_line(fat)
    assert len(s2) <= bench.MAX_LINE_BYTES and json.loads(s2)["roofline"]["frac"] == rf["frac"]
