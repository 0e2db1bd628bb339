"""The HiFi-GAN twin of tools/dp_overlap_probe.py: one rank, the real BucketSync path (begin() -> engine bucket callbacks -> reduce()), all_reduce replaced
by a marker kernel on the stream it is called on.  Under `rocprofv3 --kernel-trace`: when can each of the 8 discriminator / 6 generator buckets' exchanges start?

    rocprofv3 --kernel-trace -d /tmp/t -o d -- python tools/dp_overlap_probe_hg.py ; python tools/trace_dump.py <db> out.csv ; python tools/dp_overlap_probe_hg.py --report out.csv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    adam = [i for i, r in enumerate(rows) if r["name"].startswith("adamw_kernel")]
    # an iteration = [after the generator's AdamW of the previous one, this one's generator AdamW]; the discriminator's AdamW splits it
    for k in range(len(adam) - 4, len(adam) - 1, 2):
        seg = rows[adam[k - 1] + 1:adam[k + 1] + 1]
        t0 = int(seg[0]["start_ns"])
        us = lambda r: round((int(r["start_ns"]) - t0) / 1e3)
        ad = [us(r) for r in seg if r["name"].startswith("adamw_kernel")]
        marks = [us(r) for r in seg if "sign_kernel" in r["name"]]
        print("iteration: AdamW(D) at %s us, AdamW(G) at %s us; exchange markers start at %s us" % (ad[0] if ad else "?", ad[-1] if ad else "?", marks))
    sys.exit(0)

import socket
import torch
import torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)
from xva_trainer_amd.hifigan import engine as E
from xva_trainer_amd.hifigan.step import BucketSync, HifiganStep
import bench

dev = torch.device("cuda", 0); torch.cuda.set_device(0)
_scratch = torch.zeros(4096, dtype=torch.int8, device=dev)


class _Work:
    def __init__(self, ev): self.ev = ev
    def wait(self): torch.cuda.current_stream().wait_event(self.ev)


def fake_all_reduce(t, group=None, async_op=False, op=None):
    _scratch.sign_()
    ev = torch.cuda.Event(); ev.record()
    return _Work(ev)


st = HifiganStep(dev, "bf16")
bench.init_hifigan_weights(st)
st.sync_d, st.sync_g = BucketSync(E.D, st.grads_d), BucketSync(E.G, st.grads_g)
torch.distributed.all_reduce = fake_all_reduce
x, y, y_mel = bench.hifigan_inputs(64, 0, dev, 8192)
for _ in range(5):
    st.train_step(x, y, y_mel)
torch.cuda.synchronize()
print("done")
