"""Which Python lines of the xVAPitch C5 iteration cause the torch glue launches (aten::copy_ / fill_ / add / cat / ...), forward AND backward:
torch.profiler with stacks over ONE iteration, aggregated by (aten op, innermost frame inside xva-trainer_amd).  python tools/c5_glue_sites.py"""
import collections, os, runpy, sys
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
import torch
from torch.profiler import profile, ProfilerActivity
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    g["iteration"]()
torch.cuda.synchronize()
OPS = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::cat", "aten::mul", "aten::clone", "aten::contiguous", "aten::sum", "aten::flip",
       "aten::index", "aten::stack", "aten::sub", "aten::exp", "aten::neg", "aten::div", "aten::where", "aten::masked_fill", "aten::_to_copy")
cnt = collections.Counter()
leaf = collections.Counter()
for ev in prof.events():
    if ev.name not in OPS:
        continue
    site = "?"
    for fr in ev.stack:
        if "xva-trainer_amd" in fr or "xva_trainer_amd" in fr:
            site = fr.split("xva-trainer_amd/")[-1].split("xva_trainer_amd/")[-1]
            break
    if site == "?" and ev.stack:
        site = "autograd:" + ev.stack[0][-60:]
    cnt[(ev.name, site)] += 1
    leaf[ev.name] += 1
print(dict(leaf))
for (op, site), n in cnt.most_common(70):
    print("%5d  %-18s %s" % (n, op, site))
