"""Does the gradient exchange of a FastPitch DP step start UNDER backward on this box's stream -> hardware-queue mapping?  One rank, the real GradSync
code path (bucket events recorded by xva_fp_backward_ex, the side stream waiting on them), with torch.distributed.all_reduce replaced by a marker kernel
on the stream it is called on (a 1-rank process group has no exchange to time): under `rocprofv3 --kernel-trace` the marker kernels' start times tell
when each bucket's exchange could begin.

    rocprofv3 --kernel-trace -d /tmp/t_dp -o d -- python tools/dp_overlap_probe.py ; python tools/trace_dump.py <db> out.csv ; python tools/dp_overlap_probe.py --report out.csv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    # marker = torch's sign kernel on an int8 scratch tensor (unique in this workload); steps are delimited by lamb_pass1_kernel
    lamb = [i for i, r in enumerate(rows) if r["name"].startswith("lamb_pass1")]
    for s in range(max(0, len(lamb) - 3), len(lamb) - 1):
        seg = rows[lamb[s] + 1:lamb[s + 1]]
        t0 = int(seg[0]["start_ns"])
        bwd_end = max(int(r["end_ns"]) for r in seg if "xva_gemm" in r["name"] or "layernorm_bwd" in r["name"])
        marks = [(int(r["start_ns"]) - t0) / 1e3 for r in seg if "sign_kernel" in r["name"]]
        names = sorted({r["name"][:80] for r in seg if "sign_kernel" in r["name"]})
        print("step %d (t = 0: first kernel after the previous LAMB step): backward ends at %.0f us; %d exchange markers (%s) start at %s us"
              % (s, (bwd_end - t0) / 1e3, len(marks), "; ".join(names), [round(m) for m in marks]))
    sys.exit(0)

import torch
import torch.distributed as dist
from xva_trainer_amd import synthetic
from xva_trainer_amd.fastpitch import engine as E, params as P
from xva_trainer_amd.fastpitch.lamb import Lamb
from xva_trainer_amd.fastpitch import dp

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


class _Work:
    def __init__(self, ev): self.ev = ev
    def wait(self): torch.cuda.current_stream().wait_event(self.ev)


# the marker must not disturb the numbers: operate on a scratch tensor instead of the bucket
_scratch = torch.zeros(4096, dtype=torch.int8, device=dev)


def fake_all_reduce(t, group=None, async_op=False, op=None):
    if t.numel() > 1000:
        _scratch.sign_()
    ev = torch.cuda.Event(); ev.record()
    return _Work(ev)


dist.all_reduce = fake_all_reduce
dp.dist.all_reduce = fake_all_reduce
eng = E.FastPitchEngine(dev, "bf16", p_dropout=0.1, seed=1)
flat = torch.zeros(eng.total, device=dev); P.default_init_(flat, eng.table, seed=1234)
grads = torch.zeros_like(flat)
opt = Lamb(flat, eng.table, lr=1e-4, betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
ranges = E.trainable_ranges(3)
active = {t[0] for t in eng.table if any(b <= t[1] < e for b, e in ranges)}
batch = E.DeviceBatch.from_dict(synthetic.fastpitch_batch(32, 150, 860, 1234), dev)
sync = dp.GradSync(eng, flat, grads, 2)
for _ in range(6):
    grads.zero_()
    sync.fwd_loss_bwd(batch, 3, grad_scale=1.0)
    opt.step(grads, active, max_grad_norm=1000.0)
torch.cuda.synchronize()
print("done")
