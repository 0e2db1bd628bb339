import torch, sys
sys.path.insert(0, ".")
import bench
from xva_trainer_amd import _lib
L = _lib.lib
import ctypes as C
rows, Cc = 27584, 384
dt = torch.bfloat16
x = torch.randn(rows, Cc, device="cuda").to(dt); dy = torch.randn(rows, Cc, device="cuda").to(dt)
mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda"); gamma = torch.ones(Cc, device="cuda")
dx = torch.empty_like(x); dxm = torch.empty_like(x); dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
lens = torch.full((32,), 860, device="cuda", dtype=torch.int32)
def run(drop):
    L.xva_fp_layernorm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dx), _lib.ptr(dxm) if drop else None, 1, _lib.ptr(dg), _lib.ptr(db),
        C.c_int64(rows), Cc, 2, _lib.ptr(lens), 862, 0, C.c_float(0.0), C.c_uint64(0), 0, C.c_float(0.1 if drop else 0.0), C.c_uint64(5), 3, None, None, _lib.stream_ptr())
for drop in (1, 0):
    print("drop", drop, "us", bench.timed_us(lambda: run(drop), iters=50, warm=5))
