"""Host time per autograd.Function of the xVAPitch C5 iteration, forward and backward (the backward runs on autograd's thread, which cProfile does not see):
every Function subclass of xva-trainer_amd/xvapitch is wrapped with a wall-clock timer.  python tools/c5_backward_profile.py"""
import collections, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
acc = collections.defaultdict(lambda: [0, 0.0])
mods = [importlib.import_module("xva_trainer_amd.xvapitch." + m) for m in ("ops", "wn", "transformer", "sdp", "acoustic", "decoder", "discriminator", "generator_pass", "train_step")]
def wrap(cls, name):
    orig = getattr(cls, name)
    def f(*a, **k):
        t0 = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            e = acc[(cls.__name__, name)]; e[0] += 1; e[1] += time.perf_counter() - t0
    setattr(cls, name, staticmethod(f))
for m in mods:
    for n, o in list(vars(m).items()):
        if isinstance(o, type) and issubclass(o, torch.autograd.Function) and o is not torch.autograd.Function and o.__module__ == m.__name__:
            wrap(o, "forward"); wrap(o, "backward")
import runpy
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
acc.clear()
N = 3
t0 = time.perf_counter()
for _ in range(N):
    g["iteration"]()
tot = (time.perf_counter() - t0) / N
print("iteration %.1f ms; host ms per iteration inside Function.forward / backward (nested calls counted in both):" % (tot * 1e3))
for (c, n), (k, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.2f ms  %5.0f calls  %s.%s" % (t / N * 1e3, k / N, c, n))
