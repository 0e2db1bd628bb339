"""Would two half-batch chains on two streams beat one full-batch chain?  FastPitch decoder FFN forward (conv1 + conv2) and backward-data on
M = 27584 rows against the same work as two independent 13792-row chains on two streams.  python tools/half_batch_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xva_trainer_amd import _lib as L

dt = torch.bfloat16
R = 32 * 862
H = R // 2
x = torch.randn(R + 2, 384, device="cuda").to(dt); h = torch.zeros(R + 2, 1536, device="cuda", dtype=dt)
W1 = torch.randn(1536, 1152, device="cuda").to(dt) * 0.03; W2 = torch.randn(384, 4608, device="cuda").to(dt) * 0.03
o2 = torch.zeros(R + 2, 384, device="cuda", dtype=dt)
b1 = torch.randn(1536, device="cuda"); b2 = torch.randn(384, device="cuda")
lens = torch.full((32,), 860, device="cuda", dtype=torch.int32)


def chain(r0, rows, nlayers=6):
    """conv1 (+bias+relu+mask) -> conv2 (+bias+dropout+residual+mask), nlayers times, on rows [r0, r0 + rows)"""
    for _ in range(nlayers):
        L.gemm(x, W1, h, rows, 1536, 1152, 384, 1152, 1536, compute=1, a_offset=(r0 + 1) * 384 - 384, c_offset=(r0 + 1) * 1536, bias=b1, relu=True,
               mask_mode=L.MASK_LEN, lens=lens[r0 // 862:], Tp=862, mask_pad=1, mask_len=860)
        L.gemm(h, W2, o2, rows, 384, 4608, 1536, 4608, 384, compute=1, a_offset=(r0 + 1) * 1536 - 1536, c_offset=(r0 + 1) * 384, bias=b2, R=x[r0 + 1:], ldr=384,
               mask_mode=L.MASK_LEN, lens=lens[r0 // 862:], Tp=862, mask_pad=1, mask_len=860, drop_p=0.1, drop_seed=5, drop_stream=3)


def timed(fn, iters=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def two_halves():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        chain(0, H)
    with torch.cuda.stream(s2):
        chain(H, R - H)
    cur.wait_stream(s1); cur.wait_stream(s2)


print("one chain of 27584 rows : %.3f ms" % timed(lambda: chain(0, R)))
print("two chains of 13792 rows: %.3f ms" % timed(two_halves))
