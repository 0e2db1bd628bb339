"""Who launches what in the xVAPitch C5 iteration: torch.profiler (CPU + device) over one iteration; every device kernel / memcpy / memset is attributed to
the outermost autograd Function (or aten operator when outside one) that was on the host stack when it was launched.
python tools/c5_launch_map.py [detail]   (detail: also the per-(owner, kernel) table)"""
import collections, os, runpy, sys
detail = len(sys.argv) > 1
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
import torch
from torch.profiler import profile, ProfilerActivity
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    g["iteration"]()
    torch.cuda.synchronize()
own = collections.Counter(); pair = collections.Counter(); kern = collections.Counter()
def owner(ev):
    top, inner = None, ev.name
    p = ev
    while p is not None:
        n = p.name
        if not n.startswith(("hip", "cuda")) and not n.startswith("autograd::engine"):
            top = n
        p = p.cpu_parent
    return top or ev.name
for ev in prof.events():
    ks = getattr(ev, "kernels", None)
    if not ks or ev.device_type != torch.autograd.DeviceType.CPU:
        continue
    if ev.cpu_children and any(getattr(c, "kernels", None) for c in ev.cpu_children):
        continue                                   # count a launch once: at the innermost host event that owns it
    o = owner(ev)
    for k in ks:
        kn = k.name.split("(")[0].replace("void ", "")[:70]
        own[o] += 1; pair[(o, kn)] += 1; kern[kn] += 1
print("device launches in one iteration: %d" % sum(own.values()))
for o, n in own.most_common(40):
    print("%6d  %s" % (n, o))
if detail:
    print("-- (owner, kernel)")
    for (o, k), n in pair.most_common(150):
        print("%6d  %-28s %s" % (n, o[:28], k))
