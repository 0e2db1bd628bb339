"""Does a consumer GEMM find its producer's output in the 256 MiB Infinity Cache?  

FastPitch decoder FFN at the bench geometry: conv1 forward writes the 84.7 MB intermediate h (27 584 x 1 536 bf16), conv2 forward reads it as its A
operand.  conv2 is timed (a) warm (the same launch repeated), (b) straight after conv1 wrote h, caches flushed BEFORE conv1, (c) after a 512 MB
flush, (d) as (b) with a 100 MB / 200 MB unrelated stream between producer and consumer (what the other stream lanes put in between).
If (b) ~ (a) the writes allocate in the Infinity Cache and the in-step "cold" penalty comes from what runs in between; if (b) ~ (c) they do not,
and no tile order will help."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L
dt = torch.bfloat16
R = 32 * 862
x = torch.randn(R + 2, 384, device="cuda").to(dt); h = torch.zeros(R + 2, 1536, device="cuda", dtype=dt)
W1 = (torch.randn(1536, 1152, device="cuda") * 0.03).to(dt); W2 = (torch.randn(384, 4608, device="cuda") * 0.03).to(dt)
o2 = torch.zeros(R, 384, device="cuda", dtype=dt)
big = torch.zeros(512 << 20, device="cuda", dtype=torch.uint8)
mid = torch.zeros(200 << 20, device="cuda", dtype=torch.uint8)
c1 = lambda: L.gemm(x[1:], W1, h[1:], R, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=-384, relu=True)
c2 = lambda: L.gemm(h[1:], W2, o2, R, 384, 4608, 1536, 4608, 384, compute=1, a_offset=-1536)


def run(mode, iters=12):
    ts1, ts2 = [], []
    for it in range(iters + 2):
        if mode != "warm":
            big.add_(1)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        ev[0].record()
        if mode in ("after_producer", "after_producer_100MB", "after_producer_200MB"):
            c1()
        ev[1].record()
        if mode == "after_producer_100MB":
            mid[:50 << 20].add_(1)
        if mode == "after_producer_200MB":
            mid[:100 << 20].add_(1)
        ev[2].record()
        c2()
        ev[3].record()
        torch.cuda.synchronize()
        if it >= 2:
            ts1.append(ev[0].elapsed_time(ev[1])); ts2.append(ev[2].elapsed_time(ev[3]))
    return sum(ts1) / len(ts1) * 1e3, sum(ts2) / len(ts2) * 1e3


for mode in ("warm", "cold", "after_producer", "after_producer_100MB", "after_producer_200MB", "warm"):
    t1, t2 = run(mode)
    print("%-22s conv1 fwd %7.1f us   conv2 fwd %7.1f us" % (mode, t1, t2))
