"""Which Python lines of the xVAPitch C5 iteration issue torch fill / copy launches (torch.zeros, zeros_like, .zero_(), .contiguous() of a
non-contiguous tensor, .float() casts, torch.cat): call counts per source line over ONE iteration.  Diagnostic for the launch count."""
import collections, os, runpy, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

counts = collections.Counter()
armed = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if "xva-trainer_amd" in fr.filename or "xva_trainer_amd" in fr.filename:
            return "%s:%d" % (os.path.basename(fr.filename), fr.lineno)
    return "?"


def wrap(owner, name, tag, cond=None):
    orig = getattr(owner, name)

    def f(*a, **k):
        if armed[0] and (cond is None or cond(*a, **k)):
            counts[(tag, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


wrap(torch, "zeros", "zeros")
wrap(torch, "zeros_like", "zeros_like")
wrap(torch, "cat", "cat")
wrap(torch, "stack", "stack")
wrap(torch.Tensor, "zero_", "zero_")
wrap(torch.Tensor, "contiguous", "contiguous(copy)", lambda t, *a, **k: not t.is_contiguous())
wrap(torch.Tensor, "clone", "clone")
wrap(torch.Tensor, "copy_", "copy_")
wrap(torch.Tensor, "float", "float(cast)", lambda t, *a, **k: t.dtype != torch.float32)
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
armed[0] = True
g["iteration"]()
armed[0] = False
tot = collections.Counter()
for (tag, s), n in counts.items():
    tot[tag] += n
print(dict(tot))
for (tag, s), n in counts.most_common(60):
    print("%5d  %-18s %s" % (n, tag, s))
