"""The xVAPitch (C5) twin of tools/dp_overlap_probe.py: one rank, the real BucketedSync path of the trainer's iteration — attach() before the generator
backward (tensor hooks on the decoder's input and the flow's input), start_generator() after it, start_discriminator() after the discriminator pass —
with all_reduce replaced by a marker kernel on the stream it is called on (since the five-stream iteration: the discriminator pass runs inside the generator pass
and start_discriminator() comes right after the forward pass, as in the trainer).  Under `rocprofv3 --kernel-trace`: when can each bucket's exchange start,
relative to the generator backward?

    rocprofv3 --kernel-trace -d /tmp/t -o d -- python tools/dp_overlap_probe_c5.py ; python tools/trace_dump.py <db> out.csv ; python tools/dp_overlap_probe_c5.py --report out.csv"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    adam = [i for i, r in enumerate(rows) if r["name"].startswith("adamw_kernel")]
    # an iteration ends with the optimiser's adamw launches (3 per iteration: acoustic arena, decoder, discriminator); take the last two whole ones
    ends = [adam[i] for i in range(len(adam)) if i + 1 == len(adam) or adam[i + 1] - adam[i] > 50]
    for k in range(max(1, len(ends) - 2), len(ends)):
        seg = rows[ends[k - 1] + 1:ends[k] + 1]
        t0 = int(seg[0]["start_ns"])
        us = lambda r: round((int(r["start_ns"]) - t0) / 1e3)
        marks = [us(r) for r in seg if "sign_kernel" in r["name"] and "<16" in r["name"]]         # the int8 marker (16 elements per vector)
        bounds = [us(r) for r in seg if "sign_kernel" in r["name"] and "<16" not in r["name"]]   # the fp32 one the host enqueues: forward (+ discriminator pass) done | generator backward done | start_generator() done
        ad = [us(r) for r in seg if r["name"].startswith("adamw_kernel")]
        print("iteration of %d us: generator forward ends at %s us, generator backward at %s us, the last generator bucket is issued at %s us, AdamW at %s us; "
              "exchange markers start at %s us (first: the discriminator's bucket)" % (round((int(seg[-1]["end_ns"]) - t0) / 1e3), *(bounds + ["?"] * 3)[:3], ad, marks))
    sys.exit(0)

import socket
import torch
import torch.distributed as dist
with socket.socket() as s:
    s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
dist.init_process_group("gloo", rank=0, world_size=1, init_method="tcp://127.0.0.1:%d" % port)
sys.argv = [sys.argv[0], "16", "100", "400", "bf16", "bf16"]
import runpy
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_step_time.py"), run_name="c5")      # builds the model + batch, runs 7 plain iterations
from xva_trainer_amd.xvapitch.train_step import BucketedSync

dev = torch.device("cuda", 0)
_scratch = torch.zeros(4096, dtype=torch.int8, device=dev)
_bound = torch.zeros(4096, dtype=torch.float32, device=dev)


def fake_all_reduce(t, group=None, async_op=False, op=None):
    _scratch.sign_()


dist.all_reduce = fake_all_reduce
step = g["step"]
sync = BucketedSync(step)
for _ in range(4):
    step.gen.zero_grad(); g["D"].zero_grad()
    # the trainer's order (xvapitch/xva_train.py iteration): the discriminator pass runs INSIDE the generator pass on the vocoder branch's stream, its bucket
    # goes out before the generator backward starts
    o = step.generator_pass(g["tokens"], g["x_lens"], g["y"], g["y_lens"], g["wav"], g["dvec"], g["lids"], pitch_padded=g["pitch"], eager_disc=True)
    _bound.sign_()                                       # forward (+ discriminator pass) issued; main has joined the branch
    ld = step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
    sync.start_discriminator()
    sync.attach(o)
    o["loss"].backward()
    _bound.sign_()
    sync.start_generator()
    _bound.sign_()
    sync.finish("gen"); sync.finish("disc")
    step.optimizer_step(lr=1e-6, lr_disc=1e-6)
torch.cuda.synchronize()
print("done")
