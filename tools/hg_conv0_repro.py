"""stress: the direct conv0 kernel (xva_hg_cin1_fwd) on side streams next to other work; is its output reproducible?"""
import ctypes as C, sys, os
sys.path.insert(0, "/root/repo")
import torch
from xva_trainer_amd import _lib
lib = _lib.lib
vp, i32 = C.c_void_p, C.c_int
lib.xva_hg_cin1_fwd.restype = i32
lib.xva_hg_cin1_fwd.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, i32, C.c_float, vp]
lib.xva_hg_cin1_out_len.restype = i32
dt = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # 1 = bf16, 0 = fp32
load = int(sys.argv[2]) if len(sys.argv) > 2 else 1        # concurrent matmul load on other streams
torch.manual_seed(0)
cfgs = [(4, 8192, 7, 5, 3, 2, 32), (4, 8192, 3, 5, 3, 2, 32), (4, 8192, 1, 15, 1, 7, 128)]
streams = [torch.cuda.Stream() for _ in range(4)]
A = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for (nb, Tw, p, k, s, P, Cout) in cfgs:
    wav = torch.randn(nb, Tw, device="cuda")
    W = torch.randn(Cout, k, device="cuda") * 0.3
    bias = torch.randn(Cout, device="cuda") * 0.1
    Tout = lib.xva_hg_cin1_out_len(Tw, p, k, s, P)
    Hp, padF = Tout + 8, 4
    tdt = torch.bfloat16 if dt == 1 else torch.float32
    outs = [torch.zeros(nb * p, Hp, Cout, device="cuda", dtype=tdt) for _ in streams]
    torch.cuda.synchronize()
    ref = None
    bad = 0
    for rep in range(60):
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for si, st in enumerate(streams):
            with torch.cuda.stream(st):
                if load:
                    for _ in range(2):
                        torch.mm(A, A)
                _lib.check(lib.xva_hg_cin1_fwd(_lib.ptr(wav), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(outs[si]), dt, nb, Tw, p, k, s, P, Cout, Hp, padF, 0.1,
                                               C.c_void_p(st.cuda_stream)), "cin1")
                if load:
                    torch.mm(A, A)
        torch.cuda.synchronize()
        if ref is None:
            ref = outs[0].clone()
        for si, o in enumerate(outs):
            if not torch.equal(o, ref):
                bad += 1
                d = (o != ref).nonzero()
                if bad <= 3:
                    print("   rep", rep, "stream", si, "differs in", d.size(0), "elements; items", sorted(set(d[:, 0].tolist()))[:6], "rows", int(d[:, 1].min()), int(d[:, 1].max()),
                          "channels", sorted(set(d[:, 2].tolist()))[:16], "max abs diff", float((o.float() - ref.float()).abs().max()))
    print("cfg", (nb, Tw, p, k, s, P, Cout), "dt", dt, "load", load, ": mismatching outputs", bad, "of", 60 * len(streams))
