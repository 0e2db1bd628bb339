"""LayerNorm forward / backward of the FastPitch decoder rows (27 584 x 384 bf16), four-rows-per-wave kernels (XVA_FP_LN4=1) against one-row-per-wave (0)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from xva_trainer_amd import _lib
L = _lib.lib
rows, Cc = 27584, 384
x = torch.randn(rows, Cc, device="cuda").bfloat16(); dy = torch.randn(rows, Cc, device="cuda").bfloat16()
mean = torch.zeros(rows, device="cuda"); rstd = torch.ones(rows, device="cuda"); gamma = torch.ones(Cc, device="cuda"); beta = torch.zeros(Cc, device="cuda")
y = torch.empty_like(x); dx = torch.empty_like(x); dxm = torch.empty_like(x); dg = torch.zeros(Cc, device="cuda"); db = torch.zeros(Cc, device="cuda")
lens = torch.full((32,), 860, device="cuda", dtype=torch.int32)
def fwd():
    L.xva_fp_layernorm_fwd(_lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), 1, _lib.ptr(mean), _lib.ptr(rstd), C.c_int64(rows), Cc, 2, _lib.ptr(lens), 862,
                           C.c_float(0.0), C.c_uint64(0), 0, _lib.stream_ptr())
def bwd(drop, tail=True):
    L.xva_fp_layernorm_bwd(_lib.ptr(dy), _lib.ptr(x), _lib.ptr(mean), _lib.ptr(rstd), _lib.ptr(gamma), _lib.ptr(dx), _lib.ptr(dxm) if drop else None, 1, _lib.ptr(dg) if tail else None, _lib.ptr(db) if tail else None,
        C.c_int64(rows), Cc, 2, _lib.ptr(lens), 862, 0, C.c_float(0.0), C.c_uint64(0), 0, C.c_float(0.1 if drop else 0.0), C.c_uint64(5), 3, None, None, _lib.stream_ptr())
for mode in (0, 1):
    L.xva_fp_set_ln4(mode)
    print("LN4=%d  fwd %.1f us   bwd %.1f us   bwd+dropout copy %.1f us   (event-pair timing of back-to-back launches: ~5 us of launch cost in each)" %
          (mode, bench.timed_us(fwd, iters=50, warm=5), bench.timed_us(lambda: bwd(0), iters=50, warm=5), bench.timed_us(lambda: bwd(1), iters=50, warm=5)))
print("bwd without the dgamma / dbeta tail (no cross-wave sum, no atomics; timing only): %.1f us" % bench.timed_us(lambda: bwd(0, False), iters=50, warm=5))
