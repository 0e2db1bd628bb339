"""A / B of the xVAPitch C5 iteration inside ONE process (box-to-box and run-to-run noise is +-1.5 ms; the engines' switches are module attributes):
python tools/c5_ab.py wn=0,1 tr=1 dds=1 [rounds=4 iters=20]   ->  median ms per iteration of every combination, interleaved round by round."""
import itertools, os, runpy, statistics, sys, time
opts = dict(a.split("=") for a in sys.argv[1:])
rounds, iters = int(opts.pop("rounds", 4)), int(opts.pop("iters", 20))
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
import torch
from xva_trainer_amd.xvapitch import sdp, transformer, wn
SW = {"wn": (wn, "_WN_ENGINE"), "tr": (transformer, "_ENGINE"), "dds": (sdp, "_DDS_ENGINE"), "cf": (sdp, "_CF_FUSED")}
names = sorted(opts)
combos = list(itertools.product(*[[int(v) for v in opts[n].split(",")] for n in names]))
step, D = g["step"], g["D"]
def iteration():
    step.gen.zero_grad(); D.zero_grad()
    o = step.generator_pass(g["tokens"], g["x_lens"], g["y"], g["y_lens"], g["wav"], g["dvec"], g["lids"], pitch_padded=g["pitch"], eager_disc=True)
    o["loss"].backward()
    step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
    step.optimizer_step(lr=1e-6, lr_disc=1e-6)
res = {c: [] for c in combos}
for r in range(rounds):
    for c in combos:
        for n, v in zip(names, c):
            setattr(SW[n][0], SW[n][1], v)
        for _ in range(3):
            iteration()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(iters):
            iteration()
        torch.cuda.synchronize()
        res[c].append((time.perf_counter() - t0) / iters * 1e3)
for c in combos:
    print("  ".join("%s=%d" % (n, v) for n, v in zip(names, c)), " median %.2f ms  (%s)" % (statistics.median(res[c]), " ".join("%.1f" % v for v in res[c])))
