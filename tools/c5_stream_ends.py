"""When does each stream of the xVAPitch C5 iteration run dry?  Events recorded on every stream after the forward pass has been issued and after the backward
pass has been issued, timed against an event at the start of the iteration: the stream that ends last is the critical path.  python tools/c5_stream_ends.py"""
import os, runpy, sys, time
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
import torch
step = g["step"]
gp, ac = step.gen, step.gen.acoustic
dev = torch.device("cuda", torch.cuda.current_device())
names = ["main", "vocoder branch", "text encoder", "duration predictor", "pitch predictor"]
def streams():
    return [torch.cuda.current_stream(dev), gp.branch_stream(dev)] + list(ac._streams(dev))
acc = {}
N = 5
for it in range(N + 2):
    step.gen.zero_grad(); g["D"].zero_grad()
    torch.cuda.synchronize()
    t0h = time.perf_counter()
    e0 = torch.cuda.Event(enable_timing=True); e0.record()
    o = step.generator_pass(g["tokens"], g["x_lens"], g["y"], g["y_lens"], g["wav"], g["dvec"], g["lids"], pitch_padded=g["pitch"], eager_disc=True)
    t1h = time.perf_counter()
    ef = []
    for s in streams():
        e = torch.cuda.Event(enable_timing=True); e.record(s); ef.append(e)
    o["loss"].backward()
    t2h = time.perf_counter()
    eb = []
    for s in streams():
        e = torch.cuda.Event(enable_timing=True); e.record(s); eb.append(e)
    torch.cuda.synchronize()
    if it >= 2:
        for n, a, b in zip(names, ef, eb):
            acc.setdefault(n, [0.0, 0.0]); acc[n][0] += e0.elapsed_time(a) / N; acc[n][1] += e0.elapsed_time(b) / N
        acc.setdefault("host", [0.0, 0.0]); acc["host"][0] += (t1h - t0h) * 1e3 / N; acc["host"][1] += (t2h - t0h) * 1e3 / N
print("ms from the start of the iteration until the stream has finished what was issued up to the end of the forward / of the backward pass (host: issue done)")
for n in ["host"] + names:
    print("  %-20s forward %6.2f   backward %6.2f" % (n, acc[n][0], acc[n][1]))
