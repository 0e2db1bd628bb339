import sys, ctypes as C, torch
sys.path.insert(0, '/root/repo')
from xva_trainer_amd import _lib
L = _lib.lib
B, T = 32, 860; Tp = T + 2; rows = B * Tp
g = torch.Generator().manual_seed(1)
av = torch.randn(rows, 64, generator=g).cuda().half(); x = torch.randn(rows, 384, generator=g).cuda()
W = (torch.randn(384, 64, generator=g) * 0.2).cuda().half()
gamma = torch.ones(384).cuda(); beta = torch.zeros(384).cuda(); lens = torch.full((B,), T).int().cuda()
s = torch.zeros(rows, 384, device="cuda"); y = torch.zeros_like(s); h = torch.zeros(rows, 384, device="cuda", dtype=torch.float16)
m = torch.zeros(rows, device="cuda"); r = torch.zeros(rows, device="cuda")
big = torch.zeros(512 << 20, device="cuda", dtype=torch.uint8)
def unfused(p):
    _lib.gemm(av, W, s, rows, 384, 64, 64, 64, 384, layout=_lib.GEMM_NT, compute=1, R=x, ldr=384, drop_p=p, drop_seed=5, drop_stream=1)
    L.xva_fp_layernorm_fwd_pair(_lib.ptr(s), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(y), _lib.ptr(h), C.c_int64(0), _lib.ptr(m), _lib.ptr(r), C.c_int64(rows), 384, 2, _lib.ptr(lens), Tp, _lib.stream_ptr())
def fused(p):
    L.xva_fp_onet_ln_fwd_f16(_lib.ptr(av), _lib.ptr(W), _lib.ptr(x), _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(s), _lib.ptr(y), _lib.ptr(h), _lib.ptr(m), _lib.ptr(r), C.c_int64(rows), 2, _lib.ptr(lens), Tp, C.c_float(p), C.c_uint64(5), 1, _lib.stream_ptr())
def bench(fn, p, cold):
    ts = []
    for i in range(12):
        if cold: big.add_(1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(p); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1) * 1e3)
    return sum(ts) / len(ts)
for p in (0.0, 0.1):
    print("p_drop %.1f  unfused warm %.1f cold %.1f us   fused warm %.1f cold %.1f us" % (p, bench(unfused, p, False), bench(unfused, p, True), bench(fused, p, False), bench(fused, p, True)))
