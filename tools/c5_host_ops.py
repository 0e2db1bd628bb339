"""Host-side cost of the xVAPitch C5 iteration by operator: torch.profiler (CPU activity only) over three iterations, top operators by self CPU time
and the autograd nodes by total CPU time.  python tools/c5_host_ops.py"""
import os, runpy, sys
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")
import torch
from torch.profiler import profile, ProfilerActivity
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU]) as prof:
    for _ in range(N):
        g["iteration"]()
torch.cuda.synchronize()
ka = prof.key_averages()
tot = sum(e.self_cpu_time_total for e in ka) / N / 1e3
print("sum of self CPU time per iteration %.1f ms over %d operator calls" % (tot, sum(e.count for e in ka) / N))
print("-- by self CPU time")
for e in sorted(ka, key=lambda e: -e.self_cpu_time_total)[:40]:
    print("%8.2f ms self %8.2f ms total %6.0f calls  %s" % (e.self_cpu_time_total / N / 1e3, e.cpu_time_total / N / 1e3, e.count / N, e.key[:90]))
