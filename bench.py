#!/usr/bin/env python
"""bench.py — BASELINE.json metric on MI355X: FastPitch1.1 training throughput in mel-frames/s.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full optimizer step of the hot path on one synthetic micro-batch per GPU (weak scaling):
forward + FastPitchLoss + backward + gradient all-reduce (N > 1) + grad-norm clip + fused LAMB.
Workload = BASELINE.json configs[1]: FastPitch1.1, bf16-input MFMA, batch 32/GPU, 150 tokens x 860 mel frames per clip,
training stage 3 (all heads active).  Inputs are resident in HBM before the timed region.
value = sum over ranks of true mel frames per step * K / max-over-ranks time.
Extra objects: "roofline" (MFMA GEMM, per-launch average from HIP events in an extra profiled pass of the same step) and
"cpu_baseline" (the CPU oracle = our port of the reference step, timed on a bounded sample on this host's cores).
"""
import argparse
import math
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--t-text", type=int, default=150)
    ap.add_argument("--t-mel", type=int, default=860)
    ap.add_argument("--stage", type=int, default=3)
    ap.add_argument("--compute", default="bf16", choices=["bf16", "fp32", "f16"])
    ap.add_argument("--trainer-leg", action="store_true", help="run ONLY the trainer leg: FastPitchTrainer / HiFiTrainer through handleTrainer from a dataset directory of files, "
                    "the trainer's OWN frames/s / samples/s meter next to the engine step on the same batch shape")
    ap.add_argument("--no-trainer-leg", action="store_true", help="skip the trainer leg of the default run")
    ap.add_argument("--trainer-iters", type=int, default=40, help="optimizer iterations of the trainer leg")
    ap.add_argument("--no-f16", action="store_true", help="skip the fp16-operand FastPitch leg (the mode whose outputs meet north_star's 1e-3: `value_at_tolerance`)")
    ap.add_argument("--hg-timeout", type=int, default=300, help="multi-rank runs: seconds after which the line is printed without the HiFi-GAN leg")
    ap.add_argument("--dropout", type=float, default=0.1, help="FastPitch dropout probability (reference trains with 0.1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-hifigan", action="store_true", help="skip the HiFi-GAN audio-samples/s leg")
    ap.add_argument("--no-xvapitch", action="store_true", help="skip the xVAPitch (BASELINE configs[4]) iteration timing")
    ap.add_argument("--no-fp32-parity", action="store_true", help="skip the fp32 parity-mode FastPitch timing")
    ap.add_argument("--xvapitch-leg-only", action="store_true", help="(internal) run the xVAPitch leg alone and print its object: the default run times that leg in a fresh process")
    ap.add_argument("--dry-run-gloo", action="store_true",
                    help="no GPU: run the multi-rank plumbing of this script (rank environment, process group, per-rank shards, the bucketed gradient "
                         "all-reduce over the engine's real bucket ranges, barrier + max-over-ranks timing, whole-job aggregation, the JSON line) on "
                         "CPU over gloo with a stand-in for the compute; the line carries \"dry_run\": true and is NOT a measurement")
    ap.add_argument("--share-gpu-gloo", action="store_true",
                    help="TEST ONLY (a box with one GPU): every rank uses cuda:0 and the process group is gloo instead of RCCL, so that the multi-rank "
                         "path of this script — per-rank shards, event-driven bucket all-reduces on the side stream, barriers, MAX over ranks, the rank-0-only "
                         "roofline passes next to the other ranks' collectives — executes end to end; the line carries \"shared_gpu_gloo\": true and its "
                         "value is NOT a scaling measurement")
    ap.add_argument("--hg-batch", type=int, default=64)
    ap.add_argument("--hg-steps", type=int, default=0, help="timed HiFi-GAN steps (default: min(steps, 10))")
    return ap.parse_args()


def _thread_candidates():
    n = os.cpu_count() or 1
    return sorted({t for t in (8, 32) if t <= n} or {n})     # 128 threads took 5 - 25 s per probe step and never won


def _best_of_threads(step, n_timed=3):
    """Time `step()` (one full CPU training step) honestly: for each thread count in {8, 32} (those the host has) one warm-up +
    one timed step picks the best count (more threads is NOT faster for these small GEMMs), then n_timed steps are timed at it."""
    probe = {}
    for t in _thread_candidates():
        torch.set_num_threads(t)
        step()
        t0 = time.perf_counter()
        step()
        probe[t] = time.perf_counter() - t0
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    t0 = time.perf_counter()
    for _ in range(n_timed):
        step()
    return (time.perf_counter() - t0) / n_timed, best, {str(k): round(v, 3) for k, v in probe.items()}


def c1_batch(seed=1234):
    """BASELINE.json configs[0]: batch 4, 22050 Hz clips of U[2, 10] s -> T_mel = 1 + N // 256 frames, T_text = round(15 * seconds)
    (SURVEY.md §8d), collated like TTSCollate (sorted by text length, zero padded)."""
    import numpy as np
    from oracle import fastpitch as ofp
    rng = np.random.RandomState(seed)
    secs = sorted(rng.uniform(2.0, 10.0, size=4), reverse=True)
    items = [ofp.synth_batch(1, int(round(15 * s)), 1 + int(s * 22050) // 256, seed + i, ragged=False) for i, s in enumerate(secs)]
    Tt, Tm = items[0]["text"].size(1), max(it["mel_tgt"].size(2) for it in items)
    pad = lambda t, n: torch.nn.functional.pad(t, (0, n - t.size(-1)))
    batch = {"text": torch.cat([pad(it["text"], Tt) for it in items]), "in_lens": torch.cat([it["in_lens"] for it in items]),
             "mel_tgt": torch.cat([pad(it["mel_tgt"], Tm) for it in items]), "mel_lens": torch.cat([it["mel_lens"] for it in items]),
             "pitch": torch.cat([pad(it["pitch"], Tm) for it in items]), "energy": torch.cat([pad(it["energy"], Tm) for it in items]),
             "durs": torch.cat([pad(it["durs"], Tt) for it in items])}
    return batch, [round(float(s), 2) for s in secs]


def cpu_baseline(stage):
    """Reference-equivalent CPU step (oracle/fastpitch.py: fwd + loss + autograd bwd + clip + LAMB, fp32, torch CPU ops arranged as the
    reference's graph) at BASELINE.json configs[0] (C1): batch 4, clips U[2, 10] s."""
    from oracle import fastpitch as ofp
    sd = ofp.init_state_dict(1234)
    batch, secs = c1_batch()
    frames = int(batch["mel_lens"].sum())
    state, it = {}, [50000]

    def step():
        it[0] += 1
        ofp.train_step(sd, batch, stage, state, it[0])
    dt, threads, probe = _best_of_threads(step)
    return {"value": frames / dt, "unit": "mel-frames/s", "cores": threads, "kind": "port", "s_per_step": dt, "host_cpu_count": os.cpu_count(),
            "thread_probe_s_per_step": probe,
            "sample": "C1: B=4 clips of %s s (%d true mel frames, T_text %s) stage-%d full train step (fwd+loss+bwd+clip+LAMB), fp32 torch-CPU; "
                      "per thread count 1 warm-up + 1 timed step, then 3 timed steps at the best count (%d threads)"
                      % (secs, frames, batch["in_lens"].tolist(), stage, threads)}


def hifigan_cpu_baseline():
    """Reference-equivalent CPU iteration (oracle/hifigan.py: G fwd, D step, G step, 2 x AdamW; fp32 torch-CPU) on B=2 x 8192 samples (a
    bounded sample of configs[2]'s workload: the per-item cost of the D+G iteration does not depend on the batch)."""
    from oracle import hifigan as ohg
    g_sd, mpd_sd, msd_sd = ohg.init_generator_sd(1), ohg.init_mpd_sd(2), ohg.init_msd_sd(3)
    x, y, ym = ohg.synth_batch(2, 4)
    og, od = {}, {}
    dt, threads, probe = _best_of_threads(lambda: ohg.train_step(g_sd, mpd_sd, msd_sd, x, y, ym, og, od))
    return {"value": 2 * 8192 / dt, "unit": "audio-samples/s", "cores": threads, "kind": "port", "s_per_step": dt, "host_cpu_count": os.cpu_count(),
            "thread_probe_s_per_step": probe,
            "sample": "B=2 x 8192-sample segments, full D+G iteration (G fwd, MPD+MSD x2, losses, bwd, 2 x AdamW), fp32 torch-CPU; per thread "
                      "count 1 warm-up + 1 timed step, then 3 timed steps at the best count (%d threads)" % threads}


def csrc_fingerprint():
    """sha256 over the kernel sources: a PMC summary is only quoted while it still describes the kernels that ran."""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, "xva-trainer_amd", "csrc"), os.path.join(ROOT, "include")):
        for f in sorted(os.listdir(d)):
            if f.endswith((".hip", ".h")):
                h.update(f.encode())
                h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def git_head():
    """HEAD of the tree this line was measured on (the GPU box gets a snapshot without .git: the builder's last `git rev-parse HEAD` travels in
    profiles/HEAD, an untracked file the builder.s post-commit hook rewrites; None when neither is there — the csrc fingerprint next to it always is)."""
    import subprocess
    try:
        r = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True, timeout=10)
        if r.returncode == 0 and r.stdout.strip():
            return r.stdout.strip()
    except Exception:
        pass
    try:
        return open(os.path.join(ROOT, "profiles", "HEAD")).read().strip() or None
    except Exception:
        return None


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the newest round that committed one (the readers refuse it unless its csrc fingerprint matches the sources that run)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return os.path.basename(c[-1]) if c else "r05_" + suffix


def rocprof_frac(family_regex, flops_per_launch, stats_csv, peak_tflops=2500.0):
    """roofline fraction of a kernel family from the COMMITTED rocprofv3 --kernel-trace --stats summary of this workload (profiles/<stats_csv>, lanes
    off): algorithmic FLOPs per launch / the trace's average duration.  The live `frac` next to it divides by a HIP-event pair per launch, which adds
    ~10 us of event overhead to a 100 us kernel; the two bracket the truth.  None (with the reason) when the summary is missing or describes other
    kernel sources (csrc fingerprint in the .meta.json next to it)."""
    import csv
    import re
    path = os.path.join(ROOT, "profiles", stats_csv)
    meta = path[:-4] + ".meta.json"
    if not os.path.exists(path):
        return None, "profiles/%s not committed" % stats_csv
    if not os.path.exists(meta) or json.load(open(meta)).get("csrc") != csrc_fingerprint():
        return None, "profiles/%s is stale (kernel sources changed since that trace): not quoted" % stats_csv
    pat, calls, tot = re.compile(family_regex), 0, 0.0
    for r in csv.DictReader(open(path)):
        if pat.search(r["Name"]):
            calls += int(r["Calls"])
            tot += float(r["TotalDurationUs"]) if "TotalDurationUs" in r else float(r["TotalDurationNs"]) / 1e3
    if not calls:
        return None, "no launch of %s in profiles/%s" % (family_regex, stats_csv)
    avg_us = tot / calls
    return {"frac": flops_per_launch / avg_us / 1e6 / peak_tflops, "avg_launch_us": avg_us, "launches": calls,
            "source": "profiles/%s @ %s" % (stats_csv, json.load(open(meta)).get("commit", "?"))}, None


def golden_parity(leg, compute):
    """`parity` object of a bench leg: the engine in the MODE BEING TIMED (`compute`) on the reference-recorded golden case of that leg
    (tests/golden/*.npz: inputs, outputs and losses written by a run of the reference's own classes, oracle/gen_golden_*.py), dropout off — so the
    line itself says what the timed mode does to the outputs north_star names (mel frames, waveforms, loss values; tolerance 1e-3).  CHECKER use of
    oracle/: only its seeded weight generators (`init_state_dict`, `init_*_sd`: the goldens store a checksum of the weights, not the weights) are
    called; nothing measured goes through it."""
    import numpy as np
    gdir = os.path.join(ROOT, "tests", "golden")
    relmax = lambda a, b: float((a.double().cpu() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    rel2 = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm().clamp_min(1e-30))
    if leg == "fastpitch":
        from oracle import fastpitch as ofp
        from xva_trainer_amd.fastpitch import engine as E, params as P
        g = np.load(os.path.join(gdir, "fp_stage3_small.npz"))
        batch = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in_")}
        eng = E.FastPitchEngine("cuda", compute, p_dropout=0.0)
        flat = torch.zeros(eng.total, device="cuda")
        P.to_flat(ofp.init_state_dict(int(g["seed"])), eng.table, flat)
        grads = torch.zeros_like(flat)
        b = E.DeviceBatch.from_dict(batch, "cuda")
        losses = eng.fwd_loss_bwd(flat, grads, b, 3).cpu()
        o = eng.outputs(b, 3)
        return {"case": "tests/golden/fp_stage3_small.npz (reference FastPitch + FastPitchLoss, stage 3, dropout off)", "mode": compute,
                "mel_rel": relmax(o["mel_out"].float(), torch.from_numpy(g["mel_out"])), "mel_rel_l2": rel2(o["mel_out"].float(), torch.from_numpy(g["mel_out"])),
                "pitch_pred_rel": relmax(o["pitch_pred"].float(), torch.from_numpy(g["pitch_pred"])),
                "loss_rel": abs(float(losses[0]) - float(g["loss"])) / abs(float(g["loss"])), "tolerance_north_star": 1e-3,
                "metric": "max |x - ref| / max |ref| (mel_rel, pitch_pred_rel), relative L2 (mel_rel_l2), relative scalar error (loss_rel)"}
    from oracle import hifigan as ohg
    if leg == "xvapitch":
        # the whole C5 iteration's forward on the case recorded from the reference's own xVAPitch.train_step + VitsGeneratorLoss / VitsDiscriminatorLoss
        # (oracle/gen_golden_xvapitch_*.py), in the timed mode: WaveNet stacks bf16-stored, the transformers' products bf16 MFMA, decoder / discriminator bf16
        from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
        from xva_trainer_amd.xvapitch.decoder import VitsDecoder
        from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
        from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
        from xva_trainer_amd.xvapitch.train_step import XVAPitchStep
        g, g5 = np.load(os.path.join(gdir, "xvapitch_genpass.npz")), np.load(os.path.join(gdir, "xvapitch_c5.npz"))
        c = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
        ac = AcousticTrainPath(c["vocab"], c["langs"], latent_size=c["latent"], embedded_language_dim=c["lang_dim"], d_vector_dim=c["dvec"], hidden_channels_ffn=c["ffn"],
                               num_heads=c["heads"], text_layers=c["te_layers"], posterior_layers=c["pe_layers"], flow_layers=c["flow_layers"], num_flows=c["num_flows"],
                               spec_bins=c["spec_bins"], pitch=True, compute=compute)
        ac.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
        dec = VitsDecoder(c["latent"], c["dvec"], compute=compute)
        dec.load_state_dict(ohg.init_vits_decoder_sd(int(g["dec_seed"]), c["latent"], c["dvec"]))
        D = VitsDiscriminator(compute=compute)
        D.load_state_dict(ohg.init_vits_disc_sd(int(g5["disc_seed"])))
        step = XVAPitchStep(GeneratorPass(ac, dec, spec_segment_size=int(g["seg"])), D)
        t = lambda k: torch.from_numpy(g[k]).cuda()
        step.gen.zero_grad(); D.zero_grad()
        o = step.generator_pass(t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("wav"), t("dvec"), t("lids"), pitch_padded=t("pitch"), eps=t("eps"), noise=t("noise"),
                                slice_ids=t("slice_ids"), eager_disc=True)
        o["loss"].backward()
        ld = step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
        names = ("loss_mel", "loss_kl", "loss_duration", "loss_pitch", "loss_gen", "loss_feat", "loss")
        lr = {k: abs(float(o[k]) - float(g5[k])) / abs(float(g5[k])) for k in names}
        lr["loss_disc"] = abs(float(ld) - float(g5["loss_disc"])) / abs(float(g5["loss_disc"]))
        res = {"case": "tests/golden/xvapitch_genpass.npz + xvapitch_c5.npz (reference xVAPitch.train_step + both loss classes, one iteration, the reference's draws)", "mode": compute,
               "wave_rel": rel2(o["model_outputs"].float(), torch.from_numpy(g5["model_outputs"])), "loss_rel": max(lr.values()), "loss_rel_by_name": lr,
               "tolerance_north_star": 1e-3, "metric": "relative L2 of the decoded segment (wave_rel), worst relative error of the eight reported losses (loss_rel)"}
        del step, ac, dec, D
        torch.cuda.empty_cache()
        return res
    from xva_trainer_amd.hifigan.step import HifiganStep
    g = np.load(os.path.join(gdir, "hg_step_b2.npz"))
    seed = int(g["seed"])
    st = HifiganStep("cuda", compute)
    st.load_state_dicts(ohg.init_generator_sd(seed), ohg.init_mpd_sd(seed + 1), ohg.init_msd_sd(seed + 2))
    out = st.train_step(torch.from_numpy(g["x_mel"]).cuda(), torch.from_numpy(g["y_wav"]).cuda(), torch.from_numpy(g["y_mel"]).cuda())
    ref = dict(zip([str(k) for k in g["loss_names"]], (float(v) for v in g["losses"])))
    mine = {"loss_disc_all": out["loss_disc_all"], "loss_mel": out["loss_mel"], "loss_gen": out["loss_gen"], "loss_fm": out["loss_fm"], "loss_gen_all": out["loss_gen_all"]}
    want = {"loss_disc_all": ref["loss_disc_all"], "loss_mel": ref["loss_mel"], "loss_gen": ref["loss_gen_f"] + ref["loss_gen_s"],
            "loss_fm": ref["loss_fm_f"] + ref["loss_fm_s"], "loss_gen_all": ref["loss_gen_all"]}
    lr = {k: abs(float(mine[k]) - want[k]) / abs(want[k]) for k in want}
    res = {"case": "tests/golden/hg_step_b2.npz (reference Generator + MPD + MSD + losses, one D+G iteration, B = 2)", "mode": compute,
           "wave_rel": rel2(out["y_g_hat"].float(), torch.from_numpy(g["y_g_hat"]).squeeze(1)), "loss_rel": max(lr.values()), "loss_rel_by_name": lr,
           "tolerance_north_star": 1e-3, "metric": "relative L2 of the generated waveform (wave_rel), worst relative error of the five reported losses (loss_rel)"}
    del st
    torch.cuda.empty_cache()
    return res


def timed_us(fn, iters=10, warm=2):
    """Average duration of ONE fn() in microseconds: a HIP event pair around EACH call on torch's current stream (the stream every libxvahip launch
    of this process goes to) — what the library's own xva_prof_* pairs do around a GEMM launch.  A pair around `iters` back-to-back calls (rounds
    1 - 3) also timed the host's launch-to-launch gaps: LayerNorm 19 us against 13.6 us in the kernel trace."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    pairs = []
    for _ in range(iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    return 1000.0 * sum(a.elapsed_time(b) for a, b in pairs) / iters


def hbm_kernel_rooflines(dev, opt, grads, active, compute):
    """Roofline lines of the HBM-bound kernels SURVEY.md §8(d) lists next to the GEMMs, each timed live at the C2 shapes:
    algorithmic bytes (one read + one write of each operand) / measured duration against 8 TB/s."""
    import ctypes as C
    from xva_trainer_amd import _lib
    from xva_trainer_amd.mel import TacotronSTFT
    lib = _lib.lib
    out = {}
    # what an event pair costs with nothing between its two records: the floor every per-launch duration below sits on
    pair_us = timed_us(lambda: None, iters=20)
    def line(name, us, nbytes, note, flops=None, mfma_peak=None):
        gbs = nbytes / us / 1e3
        net = max(us - pair_us, 1e-3)
        d = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": None, "avg_launch_us": us,
             "event_pair_overhead_us": pair_us, "frac_net_of_event_overhead": nbytes / net / 1e3 / 8000.0,
             "algorithmic_mbytes_per_launch": nbytes / 1e6, "note": note}
        if flops:
            d["tflops"] = flops / us / 1e6
            d["mfma_frac"] = d["tflops"] / mfma_peak
        out[name] = d
    # ---- LAMB (3 launches over the flat buffers): pass 1 reads p, g, m, v (16 B) and writes m, v (8 B); pass 2 reads p, m, v (12 B), writes p (4 B)
    n_active = sum(t[2] for t in opt.table if t[0] in active)
    us = timed_us(lambda: opt.step(grads, active, max_grad_norm=1000.0))
    line("lamb_step", us, 44.0 * n_active, "clip + LAMB over %d active parameters: grad-norm pass (4 B) + moments pass (16 B read, 8 B written) + "
         "update pass (12 B read, 4 B written) per parameter" % n_active)
    # ---- LayerNorm forward / backward at the decoder shape: 32 x (860 + 2) rows of 384 channels in the activation dtype
    rows, Cc = 32 * 862, 384
    es = 2 if compute == "bf16" else 4
    adt = torch.bfloat16 if compute == "bf16" else torch.float32
    X = torch.randn(rows, Cc, device=dev).to(adt)
    Y, dY, dX = torch.empty_like(X), torch.randn(rows, Cc, device=dev).to(adt), torch.empty_like(X)
    gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    dg, db = torch.zeros(Cc, device=dev), torch.zeros(Cc, device=dev)
    lib.xva_fp_layernorm_fwd.restype = C.c_int32
    lib.xva_fp_layernorm_fwd.argtypes = [C.c_void_p] * 4 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                         C.c_float, C.c_uint64, C.c_uint32, C.c_void_p]
    lib.xva_fp_layernorm_bwd.restype = C.c_int32
    lib.xva_fp_layernorm_bwd.argtypes = [C.c_void_p] * 7 + [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                         C.c_int32, C.c_float, C.c_uint64, C.c_uint32, C.c_float, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    dtc = 1 if compute == "bf16" else 0
    P = _lib.ptr
    fwd = lambda: _lib.check(lib.xva_fp_layernorm_fwd(P(X), P(gam), P(bet), P(Y), dtc, P(mean), P(rstd), rows, Cc, 0, None, 862, 0.0, 0, 0,
                                                      _lib.stream_ptr()), "layernorm_fwd")
    bwd = lambda: _lib.check(lib.xva_fp_layernorm_bwd(P(dY), P(X), P(mean), P(rstd), P(gam), P(dX), None, dtc, P(dg), P(db), rows, Cc, 0, None, 862, 0,
                                                      0.0, 0, 0, 0.0, 0, 0, None, None, _lib.stream_ptr()), "layernorm_bwd")
    line("layernorm_fwd", timed_us(fwd), rows * Cc * 2.0 * es + rows * 8.0, "decoder shape %d rows x %d, %s: read X, write Y (+ mean, rstd)" % (rows, Cc, compute))
    line("layernorm_bwd", timed_us(bwd), rows * Cc * 3.0 * es + rows * 8.0, "same shape: read dY, X (+ mean, rstd), write dX; dgamma / dbeta by atomics")
    del X, Y, dY, dX
    # ---- mel-STFT (M1) on the C2 clips: reflect pad, 1024-point real FFT per frame, magnitude, mel GEMM + log — an HBM-bound pipeline whose
    # intermediate spectrum (re | im, 4.1 KB per frame) is written and re-read once
    stft = TacotronSTFT().to(dev)
    wav = torch.rand(32, 219904, device=dev) * 1.6 - 0.8
    # the engine call (tap-table pre-kernel + the fused kernel): TacotronSTFT.mel_spectrogram() in front of it mirrors the reference's two range asserts
    # (common/layers.py:129-130: torch.min / torch.max of the batch read on the host), which are two reductions and two device synchronisations, not the kernel
    us = timed_us(lambda: stft.engine(wav), iters=5, warm=1)
    frames = 32 * 860
    line("mel_stft_m1", us, frames * (256 * 4.0 + 80 * 4.0), "32 clips x 219 904 samples -> 27 520 frames, ONE fused kernel (+ a one-block tap-table pre-kernel): "
         "reflect-indexed frame -> 1024-point FFT in LDS -> magnitude -> mel filterbank (per-bin taps) -> log; algorithmic 256 new samples read + 80 log-mels "
         "written per frame; nothing else leaves the CU (the four-launch pipeline it replaces moved 12.6 KB per frame, 9.4x the algorithmic bytes and took "
         "140 us of kernel time against 88).  The fused kernel is VALU-bound, not HBM-bound: ~1 000 fp32 vector instructions per frame (512-point complex FFT as three "
         "radix-8 passes + the real-input recombination + 513 magnitudes) at 4 cycles per wave64 issue = 51 us at 100 % issue; measured 80 us (rocprofv3, tools/mel_time.py)")
    return out


TILE_NAMES = {"128128": "128x128", "256256": "256x256", "64128": "128x64", "64064": "64x64", "32128": "128x32", "128384": "384x128"}


def pmc_traffic(family, pmc_csv):
    """HBM bytes per launch of one kernel family from the committed PMC summary (profiles/*_pmc_hbm_bytes.csv: separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE passes over this same workload, FETCH_SIZE in KB and x2 per the gfx950 calibration of
    MI355X_MICROARCH.md, WRITE_SIZE in KB; tools/profile_round.sh).  Counters cannot be read in-process, so this is the last
    committed measurement, or None when the file is missing."""
    import csv
    import re
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", pmc_csv)
    if not os.path.exists(path):
        return None, None
    meta = path[:-4] + ".meta.json"                       # {"csrc": fingerprint of the kernel sources the PMC pass ran, "commit": ...}
    if not os.path.exists(meta) or json.load(open(meta)).get("csrc") != csrc_fingerprint():
        return None, "profiles/%s is stale (kernel sources changed since that PMC pass): not quoted" % pmc_csv
    m = re.match(r"xva_gemm_glds_kernel<(\d+)x(\d+)>", family)
    if m:
        # both K loops of a tile: the lock-step kernel and the staggered one (xva_gemm_glds8_kernel<LAYOUT, BM, BN, WM, WN>: 256x256 and 384x128)
        pat = re.compile(r"xva_gemm_glds8?_kernel<\d, %s, %s," % (m.group(1), m.group(2)) + (r"|xva_gemm_glds8w_kernel<" if m.group(1) == m.group(2) == "256" else ""))   # (8w: the NT 256x256 loop with whole-line DMA pieces)
    elif family.startswith("xva_conv_res_kernel<CIN="):
        pat = re.compile(r"xva_conv_res_kernel<\d, %s," % family[len("xva_conv_res_kernel<CIN="):-1])
    elif family.startswith("xva_wgrad_res_kernel"):
        pat = re.compile(r"xva_wgrad_res_kernel<")
    else:
        pat = re.compile(r"xva_gemm_kernel<\d, \d, %s>" % family[len("xva_gemm_kernel<BN="):-1])
    n, by = 0, 0.0
    for r in csv.DictReader(open(path)):
        if pat.search(r["Kernel"]):
            d = int(r["Dispatches"])
            n += d
            by += d * (float(r["FETCH_SIZE_KB_mean_raw"]) * 2.0 + float(r["WRITE_SIZE_KB_mean_raw"])) * 1000.0
    return (by / n if n else None), "profiles/%s @ %s" % (pmc_csv, json.load(open(meta)).get("commit", "?"))


def pmc_iteration_traffic(pmc_csv, per_iteration_kernel="adamw_kernel", launches_per_iteration=2):
    """HBM bytes of ONE whole iteration from the committed PMC summary (every kernel: dispatches x (FETCH_SIZE x 2 + WRITE_SIZE), MI355X_MICROARCH.md's gfx950
    correction), divided by the iterations the pass ran (counted by a kernel that runs a known number of times per iteration); torch's own fill / copy kernels (workspace set-up of that
    run) are left out.  None when the file is missing
    or describes other kernel sources."""
    import csv
    path = os.path.join(ROOT, "profiles", pmc_csv)
    meta = path[:-4] + ".meta.json"
    if not os.path.exists(path) or not os.path.exists(meta) or json.load(open(meta)).get("csrc") != csrc_fingerprint():
        return None
    tot, iters = 0.0, 0
    for r in csv.DictReader(open(path)):
        d = int(r["Dispatches"])
        if "at::native" not in r["Kernel"] and "__amd_rocclr" not in r["Kernel"]:      # torch's allocation fills / copies of that run's set-up are not the iteration's
            tot += d * (float(r["FETCH_SIZE_KB_mean_raw"]) * 2.0 + float(r["WRITE_SIZE_KB_mean_raw"])) * 1000.0
        if per_iteration_kernel in r["Kernel"]:
            iters += d
    iters //= launches_per_iteration
    return tot / iters if iters else None


def gemm_roofline(run, nprof, bound, peak, note, pmc_csv=None):
    """Live per-launch timing of the GEMM kernels (HIP event pair around every xva_gemm launch, on the launch stream, recorded by the
    library itself: xva_prof_* in csrc/core.hip).  The DOMINANT kernel = the (main loop, tile) family with the largest total time;
    `achieved` = its algorithmic FLOPs (2MNK) or bytes (every distinct operand / result element once) / its summed launch time."""
    import collections
    import csv
    import tempfile
    from xva_trainer_amd import _lib
    lib = _lib.lib
    # per-kernel durations: the engines' stream lanes are put back on one stream for these passes (concurrent kernels share the CUs, a
    # launch's event pair would time its neighbours too); the timed region above runs with the lanes on
    old_hg, old_fp = lib.xva_hg_set_streams(1), lib.xva_fp_set_streams(1)
    lib.xva_prof_enable(1)
    for _ in range(nprof):
        run()
    torch.cuda.synchronize()
    lib.xva_prof_enable(0)
    lib.xva_hg_set_streams(old_hg); lib.xva_fp_set_streams(old_fp)
    path = os.path.join(tempfile.gettempdir(), "xva_gemm_launches_%d.csv" % os.getpid())
    lib.xva_prof_dump(path.encode())
    rows = list(csv.DictReader(open(path)))
    os.remove(path)
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    lay = ["NT", "NN", "TN"]
    for r in rows:
        glds = int(r["bn"]) > 1000
        if int(r["bn"]) >= 900000:
            key = "xva_conv_res_kernel<CIN=%d>" % (int(r["bn"]) - 900000)
        elif int(r["bn"]) >= 800000:
            key = "xva_wgrad_res_kernel<CIN=%d>" % (int(r["bn"]) - 800000)
        elif int(r["bn"]) >= 700000:
            key = "xva_conv_pair_kernel<C=%d>" % (int(r["bn"]) - 700000)
        else:
            key = ("xva_gemm_glds_kernel<%s>" % TILE_NAMES.get(r["bn"], r["bn"])) if glds else ("xva_gemm_kernel<BN=%s>" % r["bn"])
        f = fam[key]
        f[0] += 1; f[1] += float(r["ms"]); f[2] += float(r["gflop"]); f[3] += float(r["mbytes"])
    tot_ms = sum(f[1] for f in fam.values())
    name, f = max(fam.items(), key=lambda kv: kv[1][1])
    if bound == "auto":   # the dominant family's own side of the ridge (2.5 PFLOP/s : 8 TB/s = 312 flop per algorithmic byte)
        bound = "mfma" if f[2] * 1e9 / max(f[3] * 1e6, 1.0) > 2500.0e12 / 8000.0e9 else "hbm"
        peak = 2500.0 if bound == "mfma" else 8000.0
    if bound == "mfma":
        ach, unit = f[2] / f[1], "TFLOP/s"                   # GFLOP / ms = TFLOP/s
    else:
        ach, unit = f[3] / f[1], "GB/s"                       # MB / ms = GB/s
    traffic, traffic_src = pmc_traffic(name, pmc_csv) if pmc_csv else (None, None)
    res = {"bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak, "traffic": traffic,
           "traffic_unit": ("HBM bytes per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE; offline passes over this workload: %s)" % traffic_src) if traffic
                           else traffic_src,
           "kernel": name + (" (direct-to-LDS MFMA implicit-convolution GEMM, all layouts; 256x256: xva_gemm_glds8_kernel / xva_gemm_glds8w_kernel, the staggered K loop)" if "glds" in name else
                             (" (resident-input MFMA convolution, forward + backward-data)" if "conv_res" in name else
                              (" (resident-operand MFMA convolution weight gradient)" if "wgrad_res" in name else ""))),
           "launches_per_step": f[0] / nprof, "avg_launch_us": 1e3 * f[1] / f[0], "kernel_ms_per_step": f[1] / nprof,
           "share_of_gemm_time": f[1] / tot_ms if tot_ms else None,
           "algorithmic_gflop_per_launch": f[2] / f[0], "algorithmic_mbytes_per_launch": f[3] / f[0],
           "all_gemm": {"launches_per_step": len(rows) / nprof, "ms_per_step": tot_ms / nprof,
                        "tflops": sum(x[2] for x in fam.values()) / tot_ms if tot_ms else None,
                        "algorithmic_gbytes_per_s": sum(x[3] for x in fam.values()) / tot_ms if tot_ms else None,
                        "by_kernel": {k: {"launches_per_step": v[0] / nprof, "ms_per_step": v[1] / nprof, "tflops": v[2] / v[1],
                                          "algorithmic_gbytes_per_s": v[3] / v[1]} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])}},
           "method": "hipEvent pair around every xva_gemm launch on the launch stream (csrc/core.hip xva_prof_*); " + note +
                     "; PMC counters cannot be read in-process: `traffic` is the committed rocprofv3 --pmc measurement of this workload (profiles/)"}
    return res


def init_hifigan_weights(st, seed=1234):
    """random-init weights of the v1 architecture (no checkpoints offline): weight_v ~ N(0, 0.5 / sqrt(fan_in)), weight_g = ||v||,
    spectral-norm u / v unit vectors, zero biases."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    for which, flat in ((0, st.flat_g), (1, st.flat_d)):
        for name, off, n, shape, kind in st.eng.table[which]:
            if name.endswith("weight_g"):
                continue
            if name.endswith("bias"):
                t = torch.zeros(shape)
            elif name.endswith("weight_u") or (name.endswith("weight_v") and kind == 2):
                t = torch.nn.functional.normalize(torch.randn(shape, generator=g), dim=0)
            else:
                fan_in = 1
                for s_ in shape[1:]:
                    fan_in *= s_
                t = torch.randn(shape, generator=g) * (0.5 / max(fan_in, 1) ** 0.5)
            flat[off:off + n].copy_(t.reshape(-1))
        for name, off, n, shape, kind in st.eng.table[which]:
            if name.endswith("weight_g"):
                vname = name[:-1] + "v"
                voff, vn, vshape = next((o, nn, sh) for nm, o, nn, sh, kd in st.eng.table[which] if nm == vname)
                v = flat[voff:voff + vn].view(vshape)
                flat[off:off + n].copy_(v.reshape(vshape[0], -1).norm(dim=1))


def hifigan_inputs(B, rank, dev, seg=8192):
    """B synthetic 22050 Hz crops, peak-normalised x 0.95 (MelDataset.__getitem__), with the input mel (fmax 8000) and the loss mel (fmax None)."""
    import numpy as np
    from xva_trainer_amd import synthetic
    from xva_trainer_amd.mel import mel_spectrogram
    wav = np.stack([synthetic.synth_wave(seg, 5000 + rank * 1000 + i) for i in range(B)])
    wav = wav / np.abs(wav).max(axis=1, keepdims=True) * 0.95
    y = torch.from_numpy(wav.astype(np.float32)).to(dev)
    return mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, 8000), y, mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, None)


def hifigan_leg(a, dev, rank, world):
    """audio-samples/s of the full HiFi-GAN v1 D+G iteration (BASELINE.json configs[2]: batch 64, 8192-sample segments)."""
    from xva_trainer_amd.hifigan.step import HifiganStep
    st = HifiganStep(dev, a.compute)
    init_hifigan_weights(st)
    B, seg = a.hg_batch, 8192
    x, y, y_mel = hifigan_inputs(B, rank, dev, seg)
    steps = a.hg_steps or min(a.steps, 10)
    for _ in range(2):
        out = st.train_step(x, y, y_mel)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = st.train_step(x, y, y_mel)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = t.item()
    res = {"metric": "audio-samples/sec (HiFi-GAN v1 full D+G iteration)", "value": B * seg * world * steps / dt, "unit": "audio-samples/s",
           "ms_per_step": 1000.0 * dt / steps, "steps": steps, "dtype": a.compute,
           "config": {"workload": "HiFi-GAN v1 generator + MPD + MSD, batch %d/GPU x %d samples, D step + G step + 2 x fused AdamW" % (B, seg)},
           "loss_mel": float(out["loss_mel"].item()), "loss_disc_all": float(out["loss_disc_all"].item())}
    if rank == 0 and not a.no_roofline:
        # the conv stack is priced against the HBM roofline (north_star): algorithmic bytes of every conv-as-GEMM launch / its time.
        # The extra profiled iteration runs on rank 0 ALONE: it must not contain a collective (the other ranks have left the leg and would never
        # answer: the all-reduce — and with it this process — would hang), so the gradient exchange is switched off for it.
        st.sync_d = st.sync_g = None
        res["roofline"] = gemm_roofline(lambda: st.train_step(x, y, y_mel), 1, "auto", 8000.0, "one extra profiled D+G iteration (stream lanes off)",
                                        pmc_csv=latest_profile("hifigan_pmc_hbm_bytes.csv"))
        # the HiFi-GAN conv stack is priced on HBM — ALGORITHMIC bytes of the whole iteration (every distinct operand /
        # result element of every convolution launch once, forward + both backward products: the sum the profiled pass above recorded
        # per launch) over the TIMED iteration (stream lanes on, everything included: losses, reparametrisations, AdamW)
        ag = res["roofline"]["all_gemm"]
        alg_gb = ag["algorithmic_gbytes_per_s"] * ag["ms_per_step"] / 1e3
        alg_tf = ag["tflops"] * ag["ms_per_step"] / 1e3
        res["roofline_stack"] = {"bound": "hbm", "achieved": alg_gb / res["ms_per_step"] * 1e3, "peak": 8000.0, "unit": "GB/s",
                                 "frac": alg_gb / res["ms_per_step"] * 1e3 / 8000.0, "algorithmic_gbytes_per_step": alg_gb,
                                 "algorithmic_tflop_per_step": alg_tf, "mfma_tflops": alg_tf / res["ms_per_step"] * 1e3,
                                 "mfma_frac": alg_tf / res["ms_per_step"] * 1e3 / 2500.0,
                                 "traffic_gbytes_per_step": (lambda t: None if t is None else t / 1e9)(pmc_iteration_traffic(latest_profile("hifigan_pmc_hbm_bytes.csv"))),
                                 "note": "whole D+G iteration: algorithmic bytes of all convolution launches (B=%d) / timed ms_per_step; the workload sits on the "
                                         "ridge (17 TFLOP : 55 GB = 313 flop/B), so the MFMA fraction of the same time is given beside it" % B}
    if isinstance(res.get("roofline_stack"), dict) and res["roofline_stack"].get("traffic_gbytes_per_step"):
        rs = res["roofline_stack"]       # offline PMC passes over this workload (profiles/, csrc fingerprint checked): includes the set-up kernels of that run
        rs["traffic_over_algorithmic"] = rs["traffic_gbytes_per_step"] / rs["algorithmic_gbytes_per_step"]
    del st
    torch.cuda.empty_cache()
    if rank == 0 and world == 1:
        try:
            res["parity"] = golden_parity("hifigan", a.compute)
        except Exception as e:                               # an extra measurement: never at the price of the contract line
            res["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def hifigan_fp32_leg(a, dev, steps=3, warm=2):
    """The same D+G iteration (B x 8192 samples of the `hifigan` object) in the PARITY mode: fp32 storage, exact-fp32 MFMA — the mode whose
    waveform and nine losses meet north_star's 1e-3 against the reference's own classes (tests/test_hifigan_gpu.py::
    test_full_step_against_reference_golden).  Priced against the 157.3 TFLOP/s fp32 matrix peak."""
    from xva_trainer_amd.hifigan.step import HifiganStep
    st = HifiganStep(dev, "fp32")
    init_hifigan_weights(st)
    B, seg = a.hg_batch, 8192
    x, y, y_mel = hifigan_inputs(B, 0, dev, seg)
    for _ in range(warm):
        out = st.train_step(x, y, y_mel)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = st.train_step(x, y, y_mel)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"metric": "audio-samples/sec (HiFi-GAN v1 full D+G iteration, fp32 parity mode)", "value": B * seg * steps / dt, "unit": "audio-samples/s",
           "ms_per_step": 1000.0 * dt / steps, "steps": steps, "dtype": "fp32",
           "config": {"workload": "HiFi-GAN v1 generator + MPD + MSD, batch %d x %d samples, D step + G step + 2 x fused AdamW, fp32 storage + exact-fp32 MFMA" % (B, seg)},
           "loss_mel": float(out["loss_mel"].item()), "loss_disc_all": float(out["loss_disc_all"].item()),
           "tolerance": "waveform and all nine losses within 1e-3 (feature loss 2e-3) of the reference goldens (tests/test_hifigan_gpu.py)"}
    if not a.no_roofline:
        res["roofline"] = gemm_roofline(lambda: st.train_step(x, y, y_mel), 1, "mfma", 157.3, "one extra profiled D+G iteration in the fp32 parity mode (stream lanes off)")
        ag = res["roofline"]["all_gemm"]
        alg_tf = ag["tflops"] * ag["ms_per_step"] / 1e3
        res["roofline_stack"] = {"bound": "mfma", "achieved": alg_tf / res["ms_per_step"] * 1e3, "peak": 157.3, "unit": "TFLOP/s",
                                 "frac": alg_tf / res["ms_per_step"] * 1e3 / 157.3, "algorithmic_tflop_per_step": alg_tf,
                                 "note": "whole iteration: algorithmic FLOPs of every convolution launch / timed ms_per_step, against the exact-fp32 MFMA peak"}
    # The same fp32-stored iteration with every product formed by three bf16 MFMAs on operands split into hi + lo bf16 while staged (csrc/gemm_core.h MODE 3,
    # xva_gemm_set_fp32_products(1)): the cheapest mode whose WAVEFORM meets 1e-3 — no single-pass 16-bit format does on the 78-layer generator
    # (profiles/r06_hifigan_precision_probe.txt: bf16 1.3e-2, fp16 1.7e-3, either with an fp32 residual stream 1.1e-2 / 1.5e-3).
    from xva_trainer_amd import _lib
    old_mode = _lib.lib.xva_gemm_set_fp32_products(1)
    try:
        for _ in range(warm):
            out = st.train_step(x, y, y_mel)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = st.train_step(x, y, y_mel)
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t0
        res["split_products"] = {"ms_per_step": 1000.0 * dt3 / steps, "value": B * seg * steps / dt3, "unit": "audio-samples/s", "steps": steps,
                                 "note": "fp32 storage, products as three bf16 MFMAs on hi + lo split operands (register-staged kernel)"}
    finally:
        _lib.lib.xva_gemm_set_fp32_products(old_mode)
    del st
    torch.cuda.empty_cache()
    try:
        res["parity"] = golden_parity("hifigan", "fp32")
    except Exception as e:
        res["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    old_mode = _lib.lib.xva_gemm_set_fp32_products(1)
    try:
        par = golden_parity("hifigan", "fp32")
        par["mode"] = "fp32 storage, split-bf16 products"
        res["split_products"]["parity"] = par
    except Exception as e:
        res["split_products"]["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        _lib.lib.xva_gemm_set_fp32_products(old_mode)
    return res


def fastpitch_fp32_leg(a, dev, steps=5, warm=2):
    """The same FastPitch step (B x tokens x frames of the headline line) in the PARITY mode: fp32 storage, exact-fp32 MFMA
    (v_mfma_f32_16x16x4_f32) — the mode that meets north_star's 1e-3 against the reference end to end (tests/test_fastpitch_gpu.py,
    tests/test_fullsize_gpu.py).  Dropout 0.1, LAMB, same batch; its roofline is priced against the 157.3 TFLOP/s fp32 matrix peak."""
    from xva_trainer_amd import synthetic
    from xva_trainer_amd.fastpitch import engine as E, params as P
    from xva_trainer_amd.fastpitch.lamb import Lamb
    eng = E.FastPitchEngine(dev, "fp32", p_dropout=a.dropout, seed=1234)
    flat = torch.zeros(eng.total, device=dev)
    P.default_init_(flat, eng.table, seed=1234)
    grads = torch.zeros_like(flat)
    opt = Lamb(flat, eng.table, lr=0.1, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    ranges = E.trainable_ranges(a.stage)
    active = {t[0] for t in eng.table if any(b <= t[1] < e for b, e in ranges)}
    batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(a.batch, a.t_text, a.t_mel, 1234), dev)
    frames = int(batch.mel_lens.sum().item())
    it = [50000]

    def step():
        it[0] += 1
        opt.param_groups[0]["lr"] = 0.1 / it[0] ** 0.5
        grads.zero_()
        eng.fwd_loss_bwd(flat, grads, batch, a.stage, grad_scale=1.0)
        opt.step(grads, active, max_grad_norm=1000.0)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"metric": "mel-frames/sec (FastPitch1.1 train step, fp32 parity mode)", "value": frames * steps / dt, "unit": "mel-frames/s",
           "ms_per_step": 1000.0 * dt / steps, "steps": steps, "dtype": "fp32", "final_loss": eng.slot("LOSSES", (8,)).cpu()[0].item(),
           "config": {"workload": "FastPitch1.1 stage-%d full train step, batch %d, %d tokens x %d mel frames, dropout p=%g, fp32 storage + exact-fp32 MFMA"
                                  % (a.stage, a.batch, a.t_text, a.t_mel, a.dropout)},
           "tolerance": "outputs / losses / gradients within 1e-3 of the reference goldens (tests/test_fastpitch_gpu.py) and of the oracle at full length "
                        "(tests/test_fullsize_gpu.py::test_fastpitch_full_length_against_the_oracle)"}
    if not a.no_roofline:
        def run_profiled():
            grads.zero_()
            eng.fwd_loss_bwd(flat, grads, batch, a.stage)
        res["roofline"] = gemm_roofline(run_profiled, 1, "mfma", 157.3, "FastPitch fwd+bwd in the fp32 parity mode: 1 extra profiled pass")
    # The same fp32-stored step with every product formed by three bf16 MFMAs on operands split into hi + lo bf16 while staged
    # (csrc/gemm_core.h MODE 3, xva_gemm_set_fp32_products(1)): ~1e-5 per product; meets the same 1e-3 / 2e-3 bounds against the reference
    # goldens (tests/test_fastpitch_gpu.py::test_against_reference_golden[*-fp32_split3]).
    from xva_trainer_amd import _lib
    old_mode = _lib.lib.xva_gemm_set_fp32_products(1)
    try:
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt3 = time.perf_counter() - t0
    finally:
        _lib.lib.xva_gemm_set_fp32_products(old_mode)
    res["split_products"] = {"ms_per_step": 1000.0 * dt3 / steps, "value": frames * steps / dt3, "unit": "mel-frames/s", "steps": steps,
                             "final_loss": eng.slot("LOSSES", (8,)).cpu()[0].item(),
                             "note": "fp32 storage, products as three bf16 MFMAs on hi + lo split operands (16 mantissa bits per operand, fp32 accumulation); the feed-forward "
                                     "convolutions on split-bf16 pairs through the direct-to-LDS kernels (round 5), the attention chain / projections / predictors on the "
                                     "register-staged kernel that splits while staging; same parity bounds as the exact mode except the post-LAMB parameter norms (1e-4 instead of 1e-5)"}
    del eng, opt, grads, flat
    torch.cuda.empty_cache()
    try:
        res["parity"] = golden_parity("fastpitch", "fp32")
    except Exception as e:
        res["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    old_mode = _lib.lib.xva_gemm_set_fp32_products(1)
    try:
        par = golden_parity("fastpitch", "fp32")
        par["mode"] = "fp32 storage, split-bf16 products"
        res["split_products"]["parity"] = par
    except Exception as e:
        res["split_products"]["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        _lib.lib.xva_gemm_set_fp32_products(old_mode)
    return res


def fastpitch_f16_leg(a, dev):
    """The same FastPitch step (batch, lengths, dropout, LAMB, learning-rate schedule of the headline loop) in the fp16-OPERAND mode (round 6):
    v_mfma_f32_16x16x32_f16 on IEEE-half operand copies, the residual stream / LayerNorm inputs and outputs and their gradients in fp32, a power-of-two loss
    scale over the fp16 gradient buffers with the optimizer unscaling and skipping non-finite steps on the device (the reference's own GPU arithmetic:
    fp16 autocast + GradScaler, python/fastpitch1_1/xva_train.py:350,787,856-859).  The cheapest format found whose outputs stay within north_star's 1e-3 of
    the fp32 reference (profiles/r06_precision_probe.txt); `parity` is measured here, in this mode, on the reference-recorded golden case."""
    from xva_trainer_amd import synthetic
    from xva_trainer_amd.fastpitch import engine as E, params as P
    from xva_trainer_amd.fastpitch.lamb import Lamb
    eng = E.FastPitchEngine(dev, "f16", p_dropout=a.dropout, seed=1234)
    flat = torch.zeros(eng.total, device=dev)
    P.default_init_(flat, eng.table, seed=1234)
    grads = torch.zeros_like(flat)
    opt = Lamb(flat, eng.table, lr=0.1, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    ranges = E.trainable_ranges(a.stage)
    active = {t[0] for t in eng.table if any(b <= t[1] < e for b, e in ranges)}
    batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(a.batch, a.t_text, a.t_mel, 1234), dev)
    frames = int(batch.mel_lens.sum().item())
    it = [50000]
    skipped = torch.zeros((), device=dev)

    def step():
        it[0] += 1
        opt.param_groups[0]["lr"] = 0.1 / it[0] ** 0.5
        grads.zero_()
        eng.fwd_loss_bwd(flat, grads, batch, a.stage, grad_scale=1.0)
        opt.step(grads, active, max_grad_norm=1000.0, inv_scale=eng.grad_inv_scale)
        skipped.add_(opt.skipped)
    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"metric": "mel-frames/sec (FastPitch1.1 train step, fp16 operands + fp32 residual stream)", "value": frames * a.steps / dt, "unit": "mel-frames/s",
           "ms_per_step": 1000.0 * dt / a.steps, "steps": a.steps, "dtype": "f16", "final_loss": eng.slot("LOSSES", (8,)).cpu()[0].item(),
           "loss_scale": eng.loss_scale, "skipped_steps": int(skipped.item()),
           "config": {"workload": "FastPitch1.1 stage-%d full train step, batch %d, %d tokens x %d mel frames, dropout p=%g, fp16 MFMA operands, fp32 residual stream, loss scale 2^%d"
                                  % (a.stage, a.batch, a.t_text, a.t_mel, a.dropout, int(round(math.log2(eng.loss_scale))))},
           "tolerance": "outputs / losses within 1e-3 of the reference goldens and of the oracle (tests/test_f16_gpu.py, tests/test_fullsize_gpu.py)"}
    if not a.no_roofline:
        def run_profiled():
            grads.zero_()
            eng.fwd_loss_bwd(flat, grads, batch, a.stage)
        res["roofline"] = gemm_roofline(run_profiled, 3, "mfma", 2500.0, "FastPitch fwd+bwd in the fp16-operand mode: 3 extra profiled passes (dense fp16 MFMA peak = the bf16 one)")
    del eng, opt, grads, flat
    torch.cuda.empty_cache()
    try:
        res["parity"] = golden_parity("fastpitch", "f16")
    except Exception as e:
        res["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


class _BenchSocket:
    def __init__(self):
        self.sent = []

    async def send(self, msg):
        self.sent.append(msg)


def trainer_leg(a, dev, compute="bf16"):
    """What a user of the reference's trainer sees: `handleTrainer` -> FastPitchTrainer stage 3 / HiFiTrainer from a DATASET DIRECTORY
    (metadata.csv + wavs/ + pitch/ written by data.write_synthetic_dataset: 256 clips of 9.8 s, int16 files; durations extracted by the trainer's own
    stage-1 aligner pass), the loaders, the accumulation loop and the log lines in the loop — reported by the trainer's OWN meter (the reference's
    python/fastpitch1_1/xva_train.py:864-867: frames of the optimizer step / wall time of the step; hifigan/xva_train.py:526-528) next to the engine-only
    step on one of the same batches.  mel-frames/s and audio-samples/s as the headline line's metrics."""
    import asyncio
    import logging
    import shutil
    import tempfile
    import numpy as np
    from xva_trainer_amd import data as D
    from xva_trainer_amd.models_manager import ModelsManager
    from xva_trainer_amd.fastpitch import xva_train as FT
    from xva_trainer_amd.hifigan import xva_train as HT
    root = tempfile.mkdtemp(prefix="xva_trainer_leg_")
    res = {}
    try:
        t0 = time.perf_counter()
        n_samp = 844 * 256                                   # 845 mel frames (9.80 s): int(9 * 3.5 * 10 / 9.80) = 32 clips per micro-batch (xva_train.py:387-404)
        text = ("the quick brown fox jumps over the lazy dog and keeps running through the quiet forest until the evening sun sets behind the distant hills of home ok")[:146] + "."
        ds = D.write_synthetic_dataset(os.path.join(root, "in", "voice_bench"), n_items=256, seed=11, min_s=n_samp / 22050.0, max_s=n_samp / 22050.0, fixed_text=text)
        json.dump({"mean": 180.0, "std": 40.0}, open(os.path.join(ds, "pitch_stats.json"), "w"))
        res["dataset"] = {"clips": 256, "seconds_per_clip": n_samp / 22050.0, "write_s": time.perf_counter() - t0, "text_symbols": len(text) + 2}
        out = os.path.join(root, "out")
        mm = ModelsManager(logging.getLogger("bench"), False, str(dev))
        # ---- FastPitch, stage 3
        data = {"dataset_path": ds, "output_path": out, "checkpoint": None, "num_workers": 0, "batch_size": 9, "epochs_per_checkpoint": 100000, "force_stage": 3,
                "max_iterations": 50000 + a.trainer_iters, "trainer_options": {"compute": compute, "allow_random_init": True}}
        ws = _BenchSocket()
        t0 = time.perf_counter()
        asyncio.run(FT.handleTrainer(mm, data, ws, [dev.index or 0]))
        wall = time.perf_counter() - t0
        tr = mm.models_bank["fastpitch1_1"]
        fps = list(getattr(tr, "all_frames_s", tr.avg_frames_s))
        steady = fps[len(fps) // 4:] or fps                  # the first quarter carries the clip cache's cold file reads and the first-touch allocations
        b = next(iter(tr.train_loader))
        from xva_trainer_amd.fastpitch import engine as E
        b = b if isinstance(b, E.DeviceBatch) else E.DeviceBatch.from_dict(b, dev)
        frames = int(b.mel_lens.sum().item())
        flat = tr.model.flat.data

        def estep():
            tr.grads.zero_()
            tr.eng.fwd_loss_bwd(flat, tr.grads, b, 3, grad_scale=1.0)
        for _ in range(3):
            estep()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(16):
            estep()
        tr.optimizer.step(tr.grads, tr.active, max_grad_norm=1000.0, inv_scale=tr.eng.grad_inv_scale)        # one LAMB step per gam micro-batches, as the trainer
        torch.cuda.synchronize()
        e_ms = (time.perf_counter() - t0) / 16 * 1e3
        res["fastpitch"] = {"metric": "mel-frames/sec by the trainer's own meter (FastPitchTrainer stage 3 from files)", "value": float(np.mean(steady)), "unit": "mel-frames/s",
                            "all_iterations_mean": float(np.mean(fps)), "iterations": len(fps), "micro_batch": tr.per_rank_batch, "gam": tr.gam, "compute": compute,
                            "frames_per_micro_batch": frames, "t_text": int(b.Tt), "t_mel": int(b.Tm), "wall_s_incl_init_and_duration_extraction": wall,
                            "engine_same_shape": {"value": frames / e_ms * 1e3, "ms_per_micro_batch": e_ms, "note": "fwd + loss + bwd of one of the trainer's batches, device-resident, "
                                                  "one LAMB step per %d micro-batches amortised over 16" % tr.gam},
                            "trainer_over_engine": float(np.mean(steady)) / (frames / e_ms * 1e3), "prefetch": bool(getattr(tr, "prefetch", False))}
        del tr, b
        mm.models_bank.pop("fastpitch1_1", None)
        torch.cuda.empty_cache()
        # ---- HiFi-GAN
        hdata = {"dataset_path": ds, "output_path": out + "_hg", "checkpoint": None, "num_workers": 0, "batch_size": 46, "epochs_per_checkpoint": 100000,
                 "max_iterations": a.trainer_iters, "trainer_options": {"compute": compute, "allow_random_init": True}}
        t0 = time.perf_counter()
        asyncio.run(HT.handleTrainer(mm, hdata, _BenchSocket(), [dev.index or 0]))
        wall = time.perf_counter() - t0
        th = mm.models_bank["hifigan"]
        sps = [float(x) for x in getattr(th, "avg_samples_s", [])]
        steady = sps[len(sps) // 4:] or sps
        res["hifigan"] = {"metric": "audio-samples/sec by the trainer's own meter (HiFiTrainer from files)", "value": float(np.mean(steady)) if steady else None, "unit": "audio-samples/s",
                          "iterations": len(sps), "batch": getattr(th, "per_rank_batch", None), "wall_s": wall, "prefetch": bool(getattr(th, "prefetch", False))}
    finally:
        shutil.rmtree(root, ignore_errors=True)
    return res


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-exec this script under torch.distributed.run, one rank per GPU
    (the reference's DP entry is single-process nn.DataParallel, python/fastpitch1_1/xva_train.py:465-466; here N processes
    over RCCL).  Fails loudly when the node has fewer than N devices."""
    import socket
    have = torch.cuda.device_count()
    if have < n and "--dry-run-gloo" not in sys.argv and "--share-gpu-gloo" not in sys.argv:
        sys.exit("bench.py: --gpus %d requested but only %d GPU(s) are visible" % (n, have))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def xvapitch_c5_fresh_process(a, dev):
    """The xVAPitch leg as the trainer runs it: in a process of its own.  The iteration runs on five HIP streams, and which hardware queue a stream lands on
    depends on how many streams the process has created before (the FastPitch and HiFi-GAN engines create eight lanes): in THIS process, after the other legs,
    the same code takes 27.7 ms instead of 24.7.  XVA_BENCH_C5_INPROCESS=1 (the rocprofv3 runs of tools/profile_round.sh: one trace database) times it here."""
    import subprocess
    if os.environ.get("XVA_BENCH_C5_INPROCESS", "0") != "1":
        cmd = [sys.executable, os.path.abspath(__file__), "--xvapitch-leg-only"] + (["--no-roofline"] if a.no_roofline else []) + (["--no-cpu-baseline"] if a.no_cpu_baseline else [])
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if r.returncode == 0 and lines:
                res = json.loads(lines[-1])
                res["process"] = "fresh (python bench.py --xvapitch-leg-only), like a trainer process"
                return res
            err = "exit %d: %s" % (r.returncode, r.stderr[-400:])
        except Exception as e:
            err = "%s: %s" % (type(e).__name__, e)
        res = xvapitch_c5_leg(dev, roofline=not a.no_roofline, cpu_base=not a.no_cpu_baseline)
        res["process"] = "in the bench process (the fresh-process run failed: %s)" % err
        return res
    res = xvapitch_c5_leg(dev, roofline=not a.no_roofline, cpu_base=not a.no_cpu_baseline)
    res["process"] = "in the bench process, after the other legs (XVA_BENCH_C5_INPROCESS=1)"
    return res


def xvapitch_c5_leg(dev, B=16, Tt=100, Ty=400, iters=20, warm=3, roofline=True, cpu_base=True):
    """One xVAPitch training iteration (BASELINE configs[4] on one GPU: linear spectrograms from the raw clips, generator pass fwd + bwd,
    discriminator pass fwd + bwd, the two AdamW updates;
    xva-trainer_amd/xvapitch/train_step.py) at the reference's model size (python/xvapitch/model.py:55-149) on a synthetic batch with random weights,
    throughput mode (decoder / discriminator / WaveNet stacks bf16, transformer products bf16 MFMA on fp32 storage).  An extra measurement beside
    the two legs of `metric`: the path is parity-first (pinned to the reference's own train_step, tests/test_xvapitch_gpu.py), not tuned."""
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
    from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
    from xva_trainer_amd.xvapitch.train_step import XVAPitchStep
    VOCAB, LANGS, SEG = 256, 31, 32
    gen = torch.Generator().manual_seed(1)
    ac = AcousticTrainPath(VOCAB, LANGS, pitch=True, compute="bf16", device=dev, dropout_p=0.1, sdp_dropout_p=0.5)     # train mode: the reference's dropout (model.py:88,128,166)
    dec, D = VitsDecoder(192, 512, compute="bf16", device=dev), VitsDiscriminator(compute="bf16", device=dev)
    for eng in (dec, D):                                    # random weights: weight_v ~ N(0, 0.02), weight_g = the row norms
        sd = {k: torch.randn(shape, generator=gen) * 0.02 for k, (off, numel, shape) in eng.table.items()}
        for k in list(sd):
            if k.endswith("weight_g"):
                v = sd[k[:-1] + "v"]
                sd[k] = v.reshape(v.size(0), -1).norm(dim=1).reshape(sd[k].shape)
        eng.load_state_dict(sd)
    step = XVAPitchStep(GeneratorPass(ac, dec, SEG), D)
    x_lens = torch.randint(Tt // 2, Tt + 1, (B,), generator=gen); x_lens[0] = Tt
    y_lens = torch.randint(max(Ty // 2, SEG + 1), Ty + 1, (B,), generator=gen); y_lens[0] = Ty
    tokens = (torch.randint(1, VOCAB, (B, Tt), generator=gen) * (torch.arange(Tt)[None, :] < x_lens[:, None])).to(dev)
    frame_mask = torch.arange(Ty)[None, None, :] < y_lens[:, None, None]
    wav_lens = ((y_lens - 1) * 256 + torch.randint(0, 256, (B,), generator=gen)).to(dev)          # raw clips of y_lens frames each (1 + N // 256)
    wavs = (torch.rand(B, (Ty - 1) * 256 + 255, generator=gen) * 0.1 - 0.05).to(dev)              # quiet noise: spectrogram magnitudes O(1), like speech
    wavs = wavs * (torch.arange(wavs.size(1), device=dev)[None, :] < wav_lens[:, None])
    dvec, lids = torch.randn(B, 512, generator=gen).to(dev), torch.randint(0, LANGS, (B,), generator=gen).to(dev)
    pitch = ((torch.rand(B, 1, Ty, generator=gen) * 3 - 1.2).clamp_min(0) * frame_mask).to(dev)
    xl = x_lens.to(dev)

    def iteration():
        step.gen.zero_grad(); D.zero_grad()
        y, yl, wav = step.gen.batch_from_wav(wavs, wav_lens)        # the posterior encoder's linear spectrograms from the raw clips, on the fly
        o = step.generator_pass(tokens, xl, y, yl, wav, dvec, lids, pitch_padded=pitch, eager_disc=True)   # as the trainer calls it
        o["loss"].backward()
        ld = step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
        step.optimizer_step(lr=1e-6, lr_disc=1e-6)
        return o, ld
    for _ in range(warm):
        iteration()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(iters):
        o, ld = iteration()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / iters * 1e3
    res = {"metric": "segment audio-samples/sec (xVAPitch iteration from raw clips: spectrograms + generator pass + discriminator pass + 2 x AdamW)", "value": B * SEG * 256 / ms * 1e3,
           "unit": "audio-samples/s", "ms_per_step": ms, "steps": iters, "dtype": "bf16 (duration predictor, attention, LayerNorm, MAS, losses fp32)",
           "config": {"workload": "xVAPitch (python/xvapitch/model.py:55-149 sizes) B=%d x %d symbols x %d spectrogram frames (513 bins), 8192-sample segments, --pitch 1"
                                  % (B, Tt, Ty), "spectrogram_frames_per_s": float(y_lens.sum()) / ms * 1e3},
           "loss": float(o["loss"]), "loss_disc": float(ld), "note": "as the trainer runs it: five streams, the discriminator pass inside the generator pass (eager_disc), transformer / DDS / WaveNet stacks as engine calls; "
                   "bound by the latency of ~3 000 dependent small launches; random weights"}
    if roofline:
        # dominant convolution / GEMM kernel family of one extra profiled iteration (every xva_gemm launch of the acoustic modules, the decoder and
        # the discriminator; stream lanes off) against its own roof, and the whole iteration's algorithmic FLOPs / bytes over the timed iteration
        r = gemm_roofline(lambda: iteration(), 1, "auto", 8000.0, "one extra profiled xVAPitch iteration (stream lanes off)")
        ag = r["all_gemm"]
        r["iteration"] = {"algorithmic_tflop_per_step": ag["tflops"] * ag["ms_per_step"] / 1e3, "algorithmic_gbytes_per_step": ag["algorithmic_gbytes_per_s"] * ag["ms_per_step"] / 1e3,
                          "gemm_ms_per_step": ag["ms_per_step"], "mfma_frac_of_step": ag["tflops"] * ag["ms_per_step"] / ms / 2500.0,
                          "hbm_frac_of_step": ag["algorithmic_gbytes_per_s"] * ag["ms_per_step"] / ms / 8000.0,
                          "note": "the xva_gemm launches of the profiled (serialised, event-timed) iteration sum to %.0f %% of the timed iteration, which runs them on five "
                                  "streams (vocoder branch | text encoder | duration predictor | pitch predictor next to the posterior encoder / flow chain): "
                                  "a ratio above 100 %% is that overlap; the iteration is a dependent chain of ~3 000 small launches, bound by their latency, not by MFMA or HBM rate"
                                  % (100.0 * ag["ms_per_step"] / ms)}
        res["roofline"] = r
    try:
        res["parity"] = golden_parity("xvapitch", "bf16")
    except Exception as e:          # the leg's numbers stand without it; say why it is missing
        res["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if cpu_base:
        res["cpu_baseline"] = xvapitch_cpu_baseline(ac, dec, D, tokens, x_lens, y_lens, wavs, wav_lens, dvec, lids, pitch, SEG)
    return res


def xvapitch_cpu_baseline(ac, dec, D, tokens, x_lens, y_lens, wavs, wav_lens, dvec, lids, pitch, SEG, Bc=4, threads=8):
    """The CPU restatement of the same iteration (oracle/xvapitch.py + oracle/hifigan.py + oracle/mel.py: the checker of tests/, timed here as the
    reference-algorithm baseline on this host's cores) on the first Bc items of the bench batch with the same weights: linear spectrograms, generator
    pass + discriminator pass, forward + backward, fp32 torch-CPU (no optimiser: < 1 % of the CPU iteration)."""
    import torch.nn.functional as F
    from oracle import hifigan as ohg, mel as omel, xvapitch as oxv
    old_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        cfg = {"latent": 192, "lang_dim": 4, "dvec": 512, "heads": 2, "te_layers": 10, "pe_layers": 16, "flow_layers": 4, "num_flows": 4}
        cpu = lambda t: t[:Bc].detach().cpu()
        leaves = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in ac.state_dict().items()}
        dl = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
        ddl = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in D.state_dict().items()}
        tk, xl, yl, ww, wl, dv, li, pp = (cpu(t) for t in (tokens, x_lens.cpu(), y_lens, wavs, wav_lens, dvec, lids, pitch))
        Tyc, Ttc = int(yl.max()), int(xl.max())
        tk, pp = tk[:, :Ttc], pp[:, :, :Tyc]
        ww = F.pad(ww, (0, max(0, Tyc * 256 - ww.size(1))))[:, :Tyc * 256]
        eps, noise = torch.randn(Bc, 192, Tyc), torch.randn(Bc, 2, Ttc)
        ids = (torch.rand(Bc) * (yl - SEG + 1)).long()

        def cpu_iteration():
            t0 = time.perf_counter()
            yy = torch.stack([F.pad(omel.linear_m3(ww[i, :int(wl[i])].unsqueeze(0))[0], (0, Tyc - (1 + int(wl[i]) // 256))) for i in range(Bc)])
            o = oxv.acoustic_losses(leaves, tk, xl, yy, yl, dv, li, eps, noise, cfg, pitch_padded=pp)
            g = F.normalize(dv).unsqueeze(-1)
            wav_hat = ohg.vits_decoder(dl, oxv.segment(o["z"], ids, SEG), g)
            seg = oxv.segment(ww.unsqueeze(1), ids * 256, SEG * 256)
            loss_mel = F.l1_loss(omel.mel_m3(seg.squeeze(1)), omel.mel_m3(wav_hat.squeeze(1)), reduction="none").mean() * 45
            rs, fr, gs, fg = ohg.vits_disc({k: v.detach() for k, v in ddl.items()}, seg, wav_hat)
            loss = o["loss"] + loss_mel + ohg.generator_loss(gs) + ohg.feature_loss(fr, [[t.detach() for t in f] for f in fg])
            loss.backward()
            rs, fr, gs, fg = ohg.vits_disc(ddl, seg, wav_hat.detach())
            ohg.discriminator_loss(rs, gs).backward()
            return time.perf_counter() - t0
        cpu_iteration()
        s_ = min(cpu_iteration() for _ in range(2))
    finally:
        torch.set_num_threads(old_threads)
    return {"value": Bc * SEG * 256 / s_, "unit": "audio-samples/s", "cores": threads, "kind": "port", "s_per_step": s_, "host_cpu_count": os.cpu_count(),
            "sample": "first %d items of the bench batch (same weights): spectrograms + generator pass + discriminator pass, forward + backward, fp32 torch-CPU "
                      "oracle; 1 warm-up + best of 2 at %d threads" % (Bc, threads)}


def dry_run_gloo(a, rank, world):
    """--dry-run-gloo: the N-rank skeleton of main() on CPU — same shards (seed 1234 + rank), same bucket ranges (csrc xva_fp_bucket_range),
    same barrier / MAX-over-ranks / SUM-of-units arithmetic and the same JSON contract — with gloo for RCCL and `grads = rank + 1` standing in
    for forward + backward.  Every rank checks the reduced gradient (sum over ranks of rank + 1 in every trainable bucket)."""
    import torch.distributed as dist
    from xva_trainer_amd import synthetic
    from xva_trainer_amd.fastpitch import engine as E
    from xva_trainer_amd.fastpitch.dp import allreduce_flat_buckets, bucket_ranges, buckets_for_stage
    if world > 1:
        dist.init_process_group("gloo")
        assert dist.get_world_size() == world
    shard = synthetic.fastpitch_batch(a.batch, a.t_text, a.t_mel, 1234 + rank)
    frames_per_step = int(torch.as_tensor(shard["mel_lens"]).sum())
    ranges = [bucket_ranges()[i] for i in buckets_for_stage(a.stage)]
    total = max(e for _, e in bucket_ranges())
    grads = torch.zeros(total)
    expect = float(sum(r + 1 for r in range(world)))

    stamps = {}

    def step():
        grads.zero_()
        t_b = time.perf_counter()
        for b, e in ranges:
            grads[b:e] = rank + 1.0
        stamps["bucket_start_ms"] = [1e3 * (time.perf_counter() - t_b)] * len(ranges)       # the stand-in "backward" finishes every bucket at once
        t_e = time.perf_counter()
        if world > 1:
            allreduce_flat_buckets(grads, ranges)
        stamps["backward_ms"], stamps["allreduce_ms_exposed"] = 1e3 * (t_e - t_b), 1e3 * (time.perf_counter() - t_e)
        for b, e in ranges:
            if e > b and not (grads[b].item() == expect and grads[e - 1].item() == expect):
                sys.exit("bench.py --dry-run-gloo: rank %d bucket [%d, %d) reduced to %g / %g, expected %g" % (rank, b, e, grads[b], grads[e - 1], expect))

    barrier = (lambda: dist.barrier()) if world > 1 else (lambda: None)
    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    total_frames = float(frames_per_step)
    if world > 1:
        t = torch.tensor([dt, float(frames_per_step)], dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, total_frames = tmax[0].item(), t[1].item()
    if rank == 0:
        print(json.dumps({"metric": "mel-frames/sec (FastPitch1.1 train step) — DRY RUN of the multi-rank plumbing on CPU / gloo, no compute",
                          "value": total_frames * a.steps / dt, "unit": "mel-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                          "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "none",
                          "data": "synthetic", "dry_run": True,
                          "dp": dict(stamps, rccl_ranks=dist.get_world_size() if world > 1 else 1, backend=dist.get_backend() if world > 1 else None, rccl_version=None),
                          "config": {"workload": "stand-in step: bucketed gloo all-reduce of the stage-%d gradient ranges" % a.stage,
                                     "global_batch": a.batch * world, "per_gpu_frames_per_step": frames_per_step, "parallelism": "dp%d" % world,
                                     "buckets": len(ranges), "bucket_floats": int(sum(e - b for b, e in ranges))}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------------------
# The contract line.  Everything measured goes to bench_detail.json (tables, notes, methods, samples); stdout carries ONE compact JSON
# line — the contract keys + `roofline` + `cpu_baseline` + one small object per extra leg — short enough to survive a log tail
# (the 25 KB line of round 4 could not be read back by the driver).  tests/test_dp_cpu.py asserts the length bound.
MAX_LINE_BYTES = 6000


def _r(x, nd=4):
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float("%.*g" % (nd + 2, x))
    return x


def _pick(d, keys):
    return {k: _r(d[k]) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def _compact_roofline(r):
    if not isinstance(r, dict):
        return r
    o = _pick(r, ["bound", "achieved", "peak", "unit", "frac", "frac_rocprof", "traffic", "avg_launch_us", "launches_per_step",
                  "algorithmic_gflop_per_launch", "algorithmic_mbytes_per_launch"])
    if "kernel" in r:
        o["kernel"] = str(r["kernel"]).split(" (")[0][:64]
    if r.get("traffic") is not None:
        o["traffic_source"] = "offline rocprofv3 --pmc pass (profiles/), per launch"
    if r.get("frac_rocprof") is not None:
        o["frac_rocprof_source"] = "offline rocprofv3 kernel trace (profiles/)"
    return o


def _compact_cpu(c):
    return _pick(c, ["value", "unit", "cores", "host_cpu_count", "kind", "s_per_step"]) if isinstance(c, dict) else c


def _compact_parity(p):
    if not isinstance(p, dict):
        return p
    return {k: _r(v) for k, v in p.items() if k == "mode" or k == "error" or k.endswith("_rel") or k.endswith("_rel_l2")}


def _compact_leg(leg):
    if not isinstance(leg, dict):
        return leg
    if "error" in leg:
        return {"error": str(leg["error"])[:200]}
    o = _pick(leg, ["value", "value_at_tolerance", "unit", "ms_per_step", "ms_per_step_at_tolerance", "steps", "dtype"])
    if isinstance(leg.get("roofline_stack"), dict):
        o["roofline_stack"] = _pick(leg["roofline_stack"], ["bound", "frac", "achieved", "peak", "unit", "mfma_frac", "traffic_over_algorithmic"])
    if isinstance(leg.get("roofline"), dict):
        o["roofline"] = _pick(leg["roofline"], ["bound", "frac", "achieved", "peak", "unit"])
        if isinstance(leg["roofline"].get("iteration"), dict):
            o["iteration"] = _pick(leg["roofline"]["iteration"], ["launches_per_iteration", "mfma_frac_of_step", "hbm_frac_of_step"])
    if "parity" in leg:
        o["parity"] = _compact_parity(leg["parity"])
    if isinstance(leg.get("cpu_baseline"), dict):
        o["cpu_baseline"] = _pick(leg["cpu_baseline"], ["value", "unit", "cores", "kind"])
    if isinstance(leg.get("split_products"), dict):
        o["split_products"] = _pick(leg["split_products"], ["ms_per_step", "value"])
        if "parity" in leg["split_products"]:
            o["split_products"]["parity"] = _compact_parity(leg["split_products"]["parity"])
    return o


def compact_line(out):
    """The stdout line: the contract keys verbatim, `roofline` / `cpu_baseline` reduced to their numbers, one small object per extra leg."""
    line = {}
    for k in ["metric", "value", "value_at_tolerance", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_at_tolerance", "tolerance_mode", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data"]:
        if k in out:
            line[k] = out[k]
    line["config"] = dict(out.get("config", {}))
    for k in ["provenance", "shared_gpu_gloo"]:
        if k in out:
            line[k] = out[k]
    if isinstance(out.get("dp"), dict):
        line["dp"] = {k: ([_r(x, 3) for x in v] if isinstance(v, list) else _r(v)) for k, v in out["dp"].items()
                      if k in ("rccl_ranks", "rccl_version", "backend", "bucket_start_ms", "backward_ms", "allreduce_ms_exposed", "error")}
    if "parity" in out:
        line["parity"] = _compact_parity(out["parity"])
    if "roofline" in out:
        line["roofline"] = _compact_roofline(out["roofline"])
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _compact_cpu(out["cpu_baseline"])
    if isinstance(out.get("hbm_kernels"), dict):
        line["hbm_kernels"] = {k: _pick(v, ["frac", "achieved", "avg_launch_us"]) for k, v in out["hbm_kernels"].items() if isinstance(v, dict)}
    for k in ["fastpitch_f16", "hifigan", "xvapitch_c5", "fastpitch_split", "fastpitch_fp32_parity", "hifigan_fp32_parity"]:
        if k in out:
            line[k] = _compact_leg(out[k])
    if isinstance(out.get("trainer"), dict):
        t = out["trainer"]
        line["trainer"] = {"error": str(t["error"])[:200]} if "error" in t else {
            k: _pick(t[k], ["value", "unit", "trainer_over_engine", "iterations", "prefetch"]) for k in ("fastpitch", "hifigan") if isinstance(t.get(k), dict)}
    line["detail"] = out.get("detail_file", "bench_detail.json")
    s = json.dumps(line, separators=(",", ":"))
    if len(s) > MAX_LINE_BYTES:                           # never at the price of the contract: drop the extras, largest first
        for k in ["hbm_kernels", "hifigan_fp32_parity", "fastpitch_fp32_parity", "fastpitch_split", "trainer", "xvapitch_c5", "fastpitch_f16", "hifigan", "parity"]:
            line.pop(k, None)
            s = json.dumps(line, separators=(",", ":"))
            if len(s) <= MAX_LINE_BYTES:
                break
    return s


def emit(out):
    """Write the full measurement to bench_detail.json (next to bench.py, and under gpurun_out/ when that exists) and print the contract line."""
    root = os.path.dirname(os.path.abspath(__file__))
    paths = [os.path.join(root, "bench_detail.json")]
    if os.path.isdir(os.path.join(root, "gpurun_out")):
        paths.append(os.path.join(root, "gpurun_out", "bench_detail.json"))
    written = None
    for pth in paths:
        try:
            with open(pth, "w") as f:
                json.dump(out, f, indent=1)
            written = written or os.path.relpath(pth, root)
        except OSError:
            pass
    out["detail_file"] = written
    sys.stdout.flush()
    print(compact_line(out), flush=True)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a.gpus)                              # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        sys.exit("bench.py: --gpus %d does not match the launcher's WORLD_SIZE %d" % (a.gpus, world))
    if a.dry_run_gloo:
        return dry_run_gloo(a, rank, world)
    if a.share_gpu_gloo:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        sys.exit("bench.py: rank %d needs cuda:%d but only %d GPU(s) are visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if a.trainer_leg:
        print(json.dumps(trainer_leg(a, dev, a.compute)), flush=True)
        return
    if a.xvapitch_leg_only:                              # the child of xvapitch_c5_fresh_process: this leg alone, its object on stdout
        print(json.dumps(xvapitch_c5_leg(dev, roofline=not a.no_roofline, cpu_base=not a.no_cpu_baseline)), flush=True)
        return
    if world > 1:
        import torch.distributed as dist
        if a.share_gpu_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        assert dist.get_world_size() == world

    from xva_trainer_amd import _lib, synthetic
    from xva_trainer_amd.fastpitch import engine as E, params as P
    from xva_trainer_amd.fastpitch.lamb import Lamb
    from xva_trainer_amd.fastpitch.dp import GradSync

    torch.manual_seed(1234 + rank)
    eng = E.FastPitchEngine(dev, a.compute, p_dropout=a.dropout, seed=1234 + rank)
    flat = torch.zeros(eng.total, device=dev)
    P.default_init_(flat, eng.table, seed=1234)          # identical replicas on every rank
    grads = torch.zeros_like(flat)
    opt = Lamb(flat, eng.table, lr=0.1, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    stage = a.stage
    ranges = E.trainable_ranges(stage)
    active = {t[0] for t in eng.table if any(b <= t[1] < e for b, e in ranges)}
    if stage == 2:
        active = {n for n in active if not n.startswith("energy_emb")}
    batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(a.batch, a.t_text, a.t_mel, 1234 + rank), dev)
    frames_per_step = int(batch.mel_lens.sum().item())
    sync = GradSync(eng, flat, grads, world) if world > 1 else None
    total_iter = [50000]

    def step():
        total_iter[0] += 1
        it = total_iter[0]
        opt.param_groups[0]["lr"] = 0.1 * (1.0 / it ** 0.5 if it > 1000 else it / 1000 ** 1.5)   # xva_train.py:1252-1261
        grads.zero_()
        if sync is None:
            eng.fwd_loss_bwd(flat, grads, batch, stage, grad_scale=1.0)
        else:
            sync.fwd_loss_bwd(batch, stage, grad_scale=1.0)
        opt.step(grads, active, max_grad_norm=1000.0, inv_scale=eng.grad_inv_scale)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt, float(frames_per_step)], device=dev, dtype=torch.float64)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt = tmax[0].item()
        total_frames = t[1].item()
    else:
        total_frames = float(frames_per_step)
    loss = eng.slot("LOSSES", (8,)).cpu()[0].item()

    out = {
        "metric": "mel-frames/sec (FastPitch1.1 train step; BASELINE.json: mel-frames/sec/GPU (FastPitch) + audio-samples/sec/GPU (HiFi-GAN))",
        "value": total_frames * a.steps / dt, "unit": "mel-frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1000.0 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": a.compute, "data": "synthetic",
        "config": {"workload": "FastPitch1.1 stage-%d full train step (fwd+loss+bwd%s+clip+fused LAMB), batch %d/GPU, %d tokens x %d mel frames, dropout p=%g, %s activations"
                               % (stage, "+RCCL grad all-reduce" if world > 1 else "", a.batch, a.t_text, a.t_mel, a.dropout,
                                  {"bf16": "bf16", "fp32": "fp32", "f16": "fp32 residual stream + fp16 operand"}[a.compute]),
                   "global_batch": a.batch * world, "per_gpu_frames_per_step": frames_per_step, "parallelism": "dp%d" % world,
                   "final_loss": loss},
    }
    out["provenance"] = {"git_head": git_head(), "csrc": csrc_fingerprint()}
    if a.share_gpu_gloo:
        out["shared_gpu_gloo"] = True
    if world > 1:
        # the exchange explains itself: who took part, and one extra step with the buckets' timeline stamped
        import torch.distributed as dist
        dp = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "rccl_version": None}
        try:
            dp["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            pass
        try:
            sync.diag = True
            step()
            dp.update(sync.diagnostics() or {})
        except Exception as e:
            dp["error"] = "%s: %s" % (type(e).__name__, e)
        out["dp"] = dp
    if rank == 0 and world == 1:
        try:
            out["parity"] = golden_parity("fastpitch", a.compute)
        except Exception as e:                               # an extra measurement: never at the price of the contract line
            out["parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and not a.no_roofline:
        def run_profiled():
            grads.zero_()
            eng.fwd_loss_bwd(flat, grads, batch, stage)
        peak = 157.3 if a.compute == "fp32" else 2500.0
        out["roofline"] = gemm_roofline(run_profiled, 3, "mfma", peak,
                                        "FastPitch fwd+bwd: %d extra profiled passes after the timed region" % 3,
                                        pmc_csv=latest_profile("fastpitch_pmc_hbm_bytes.csv") if a.compute == "bf16" else None)
        if a.compute == "bf16" and "256x256" in out["roofline"]["kernel"]:
            # the same family's fraction from the committed rocprofv3 kernel trace (no event-pair overhead in the duration)
            fr, why = rocprof_frac(r"xva_gemm_glds8_kernel<\d, 256, 256,|xva_gemm_glds8w_kernel<", out["roofline"]["algorithmic_gflop_per_launch"] * 1e9,
                                   latest_profile("fastpitch_only_serial_lanes_kernel_stats.csv"))
            out["roofline"]["frac_rocprof"] = fr["frac"] if fr else None
            out["roofline"]["frac_rocprof_detail"] = fr if fr else why
    if rank == 0 and not a.no_roofline:
        out["hbm_kernels"] = hbm_kernel_rooflines(dev, opt, grads, active, a.compute)
    if rank == 0 and world == 1 and a.compute == "bf16" and not a.no_f16:
        try:
            out["fastpitch_f16"] = fastpitch_f16_leg(a, dev)
            par = out["fastpitch_f16"].get("parity", {})
            # the throughput of the cheapest mode whose OUTPUTS meet north_star's 1e-3 on the reference-recorded case, next to `value` (the bf16 mode BASELINE
            # configs[1] names, whose outputs sit at 1e-2): null if the in-run parity measurement does not confirm it
            ok = all(isinstance(par.get(k), float) and par[k] <= 1e-3 for k in ("mel_rel", "pitch_pred_rel", "loss_rel"))
            out["value_at_tolerance"] = out["fastpitch_f16"]["value"] if ok else None
            out["ms_per_step_at_tolerance"] = out["fastpitch_f16"]["ms_per_step"] if ok else None
            out["tolerance_mode"] = "fastpitch_f16 (fp16 operands, fp32 residual stream): outputs and losses within 1e-3 of the reference"
        except Exception as e:                               # an extra measurement: never at the price of the contract line
            out["fastpitch_f16"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and a.compute == "bf16" and not a.no_fp32_parity:
        try:
            out["fastpitch_fp32_parity"] = fastpitch_fp32_leg(a, dev)
            sp = out["fastpitch_fp32_parity"].get("split_products")
            if isinstance(sp, dict):     # the mode that meets north_star's 1e-3 on outputs and losses at a third of the exact mode's time: a leg of its own on the line
                out["fastpitch_split"] = {"metric": "mel-frames/sec (FastPitch1.1 train step, fp32 storage + split-bf16 products)", "value": sp["value"], "unit": sp["unit"],
                                          "ms_per_step": sp["ms_per_step"], "steps": sp["steps"], "dtype": "fp32 storage, bf16x3 products", "parity": sp.get("parity")}
        except Exception as e:                               # an extra measurement: never at the price of the contract line
            out["fastpitch_fp32_parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if not a.no_hifigan:
        del opt, grads
        eng._ws = None
        torch.cuda.empty_cache()
        if world > 1:
            # The HiFi-GAN leg is an EXTRA object of the line; in a multi-rank run it contains collectives of its own.  It must never cost the
            # contract line: if it has not finished after --hg-timeout seconds (a rank that died, a collective that never completes), rank 0
            # prints the line it already has and every rank leaves.
            import threading
            hg_done = threading.Event()

            def watchdog():
                if not hg_done.wait(a.hg_timeout):
                    if rank == 0:
                        out["hifigan"] = {"error": "the multi-rank HiFi-GAN leg did not finish within %d s: line printed without it" % a.hg_timeout}
                        emit(out)
                    os._exit(0)
            threading.Thread(target=watchdog, daemon=True).start()
        try:
            hg = hifigan_leg(a, dev, rank, world)
            if rank == 0 and world == 1 and not a.no_cpu_baseline:
                hg["cpu_baseline"] = hifigan_cpu_baseline()
        except Exception as e:
            if world == 1:
                raise
            hg = {"error": "%s: %s" % (type(e).__name__, e)}
        if world > 1:
            hg_done.set()
        out["hifigan"] = hg
        if rank == 0 and world == 1 and a.compute == "bf16" and not a.no_fp32_parity:
            try:
                out["hifigan_fp32_parity"] = hifigan_fp32_leg(a, dev)
                sp = out["hifigan_fp32_parity"].get("split_products", {})
                par = sp.get("parity", {})
                if isinstance(out.get("hifigan"), dict) and isinstance(par.get("wave_rel"), float):
                    ok = par["wave_rel"] <= 1e-3 and par.get("loss_rel", 1.0) <= 2e-3
                    out["hifigan"]["value_at_tolerance"] = sp["value"] if ok else None
                    out["hifigan"]["ms_per_step_at_tolerance"] = sp["ms_per_step"] if ok else None
                    out["hifigan"]["tolerance_mode"] = ("fp32 storage + split-bf16 products (three MFMA passes): no single-pass 16-bit format keeps the 78-layer generator's waveform "
                                                        "within 1e-3 (profiles/r06_hifigan_precision_probe.txt: bf16 1.3e-2, fp16 1.7e-3)")
            except Exception as e:                           # an extra measurement: never at the price of the contract line
                out["hifigan_fp32_parity"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not a.no_trainer_leg:
        try:
            torch.cuda.empty_cache()
            out["trainer"] = trainer_leg(a, dev, a.compute)
            if isinstance(out.get("hifigan"), dict) and out["hifigan"].get("value") and out["trainer"].get("hifigan", {}).get("value"):
                out["trainer"]["hifigan"]["trainer_over_engine"] = out["trainer"]["hifigan"]["value"] / out["hifigan"]["value"]
        except Exception as e:                               # an extra measurement: never at the price of the contract line
            out["trainer"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not a.no_xvapitch:
        try:
            torch.cuda.empty_cache()
            out["xvapitch_c5"] = xvapitch_c5_fresh_process(a, dev)
        except Exception as e:                               # an extra measurement: never at the price of the contract line
            out["xvapitch_c5"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(stage)
    if rank == 0:
        emit(out)
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
