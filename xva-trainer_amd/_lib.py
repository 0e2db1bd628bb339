"""ctypes binding of libxvahip.so (C ABI in include/xva_hip.h).

Fails loudly: there is no fallback implementation of any op in this package.
`import torch` happens first so that the HIP runtime already mapped by PyTorch
(soname libamdhip64.so.7) is the one the kernel library binds to — streams and device
pointers are then interchangeable between the two.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XVA_LIB_PATH", os.path.join(_HERE, "csrc", "libxvahip.so"))   # override: A/B builds of the kernel library


class XvaError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise XvaError(
        "libxvahip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
        "— there is no CPU/eager fallback for the hot path." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmParams(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("C", vp),
        ("M", i32), ("N", i32), ("K", i32),
        ("lda", i64), ("ldb", i64), ("ldc", i64),
        ("batch", i32),
        ("sA", i64), ("sB", i64), ("sC", i64),
        ("batch2", i32),
        ("sA2", i64), ("sB2", i64), ("sC2", i64),
        ("a_seglen", i32), ("a_segadj", i64),
        ("seglen", i32), ("seg0", i64), ("segstride", i64),
        ("a_lrelu", i32), ("b_lrelu", i32), ("a_slope", f32), ("b_slope", f32),
        ("alpha", f32), ("beta", f32),
        ("bias", vp), ("sbias2", i64),
        ("act", i32), ("act_slope", f32),
        ("R", vp), ("ldr", i64), ("sR", i64), ("sR2", i64), ("r_dtype", i32),
        ("G", vp), ("ldg", i64), ("sG", i64), ("sG2", i64), ("g_dtype", i32), ("gate_slope", f32),
        ("mask_mode", i32),
        ("lens", vp),
        ("Tp", i32), ("mask_pad", i32), ("mask_len", i32), ("mask_mul", i32), ("mask_add", i32),
        ("accumulate", i32),
        ("splitk", i32),
        ("compute", i32),
        ("layout", i32),
        ("a_dtype", i32), ("b_dtype", i32), ("c_dtype", i32), ("c_trans", i32), ("kb_len", i32), ("kb_sA", i64), ("kb_sB", i64),
        ("drop_p", f32), ("drop_seed", C.c_uint64), ("drop_stream", C.c_uint32),
        ("sk_ws", C.c_void_p), ("sk_ws_bytes", i64),
        ("C2", vp), ("c2_slope", f32),
        ("a_rowpitch", i64),
        ("F", vp), ("fm_c", f32),
        ("planes", i32), ("a_plane", i64), ("b_plane", i64), ("c_plane", i64),
        ("colsum_out", vp),
    ]


class MelConfig(C.Structure):
    _fields_ = [("n_fft", i32), ("hop", i32), ("n_mel", i32), ("pad", i32),
                ("mag_eps_add", f32), ("mag_clamp_min", f32), ("log_clamp", f32)]


GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
MASK_NONE, MASK_PAD, MASK_LEN = 0, 1, 2

lib.xva_last_error.restype = C.c_char_p
lib.xva_abi_version.restype = i32
lib.xva_target_arch.restype = C.c_char_p
lib.xva_gemm.restype = i32
lib.xva_gemm.argtypes = [C.POINTER(GemmParams), vp]
lib.xva_mel_num_frames.restype = i32
lib.xva_mel_num_frames.argtypes = [C.POINTER(MelConfig), i32]
lib.xva_mel_workspace_bytes.restype = i64
lib.xva_mel_workspace_bytes.argtypes = [C.POINTER(MelConfig), i32, i32]
lib.xva_mel_spectrogram.restype = i32
lib.xva_mel_spectrogram.argtypes = [C.POINTER(MelConfig), vp, i32, i32, i64, vp, vp, vp, vp, i64, vp]


def check(rc, what=""):
    if rc != 0:
        raise XvaError("%s failed (rc=%d): %s" % (what or "libxvahip call", rc, lib.xva_last_error().decode()))


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def raw_stream(device=None):
    """Raw hipStream_t (an int) of torch's current stream on `device` (default: the current device).  torch.cuda.current_stream() builds a Stream
    object through three Python layers (measured 8.7 us per call; the Python-sequenced xVAPitch iteration asks ~1 300 times: 11 ms of its 48);
    torch._C._cuda_getCurrentRawStream is the same query as one C call."""
    if _raw_stream is None or _get_device is None:
        return torch.cuda.current_stream(device).cuda_stream
    if device is None:
        idx = _get_device()
    elif isinstance(device, int):
        idx = device
    else:
        idx = torch.device(device).index
        if idx is None:
            idx = _get_device()
    return _raw_stream(idx)


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream (0 = default stream)."""
    return C.c_void_p(raw_stream(device))


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise XvaError("libxvahip ops need device tensors; got a %s tensor (no CPU fallback exists)" % t.device)


ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, ACT_LOGCLAMP = 0, 1, 2, 3, 4


def _dt(t):
    if t.dtype == torch.float32:
        return 0
    if t.dtype == torch.bfloat16:
        return 1
    if t.dtype == torch.float16:
        return 2          # XVA_F16: direct-to-LDS kernels only (include/xva_gemm.h)
    raise XvaError("unsupported dtype %s" % t.dtype)


def _fill_gemm(A, B, Cm, M, N, K, lda, ldb, ldc, layout=GEMM_NT, compute=0, batch=1, sA=0, sB=0, sC=0, alpha=1.0, beta=1.0,
               bias=None, relu=False, act=ACT_NONE, act_slope=0.0, R=None, ldr=0, sR=0, G=None, ldg=0, sG=0, gate_slope=0.0,
               mask_mode=MASK_NONE, lens=None, Tp=0, mask_pad=1, mask_len=0, mask_mul=1, mask_add=0, accumulate=False, splitk=1, seglen=0, seg0=0, segstride=0,
               a_seglen=0, a_segadj=0, a_lrelu=None, b_lrelu=None, a_offset=0, b_offset=0, c_offset=0, batch2=1, sA2=0, sB2=0, sC2=0, sR2=0, sG2=0,
               sk_ws=None, **extra):
    require_cuda(A, B, Cm, bias, R, G, lens)
    p = GemmParams()
    p.A = A.data_ptr() + A.element_size() * a_offset
    p.B = B.data_ptr() + B.element_size() * b_offset
    p.C = Cm.data_ptr() + Cm.element_size() * c_offset
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldb, p.ldc = lda, ldb, ldc
    p.batch, p.sA, p.sB, p.sC = batch, sA, sB, sC
    p.batch2, p.sA2, p.sB2, p.sC2 = batch2, sA2, sB2, sC2
    p.a_seglen, p.a_segadj = a_seglen, a_segadj
    p.seglen, p.seg0, p.segstride = seglen, seg0, segstride
    p.a_lrelu, p.a_slope = (1, a_lrelu) if a_lrelu is not None else (0, 0.0)
    p.b_lrelu, p.b_slope = (1, b_lrelu) if b_lrelu is not None else (0, 0.0)
    p.alpha, p.beta = alpha, beta
    p.bias = bias.data_ptr() if bias is not None else None
    p.act = ACT_RELU if relu else act
    p.act_slope = act_slope
    p.R = R.data_ptr() if R is not None else None
    p.ldr, p.sR, p.sR2 = ldr, sR, sR2
    p.r_dtype = _dt(R) if R is not None else 0
    p.G = G.data_ptr() if G is not None else None
    p.ldg, p.sG, p.sG2 = ldg, sG, sG2
    p.g_dtype = _dt(G) if G is not None else 0
    p.gate_slope = gate_slope
    p.mask_mode = mask_mode
    p.lens = lens.data_ptr() if lens is not None else None
    p.Tp, p.mask_pad, p.mask_len, p.mask_mul, p.mask_add = Tp, mask_pad, mask_len, mask_mul, mask_add
    p.accumulate = int(accumulate)
    p.splitk = splitk
    p.compute = compute
    p.layout = layout
    p.a_dtype, p.b_dtype, p.c_dtype = _dt(A), _dt(B), _dt(Cm)
    if sk_ws is not None:
        p.sk_ws, p.sk_ws_bytes = sk_ws.data_ptr(), sk_ws.numel() * sk_ws.element_size()
    for k, v in extra.items():          # any further xva_gemm_params field by name (c_trans, kb_len, drop_p, ...)
        setattr(p, k, v)
    return p


_SK_SCRATCH = {}
PARAM_EPOCH = [0]        # bumped by whoever moves parameter / gradient tensors to new storage (train_step.FlatGroupAdamW._flatten): cached pointer tables check it


def sk_scratch(device, nbytes=64 << 20):
    """Split-K slab scratch of the Python-sequenced GEMM call sites (xvapitch/wn.py, ...): one buffer per (device, stream) — launches of one
    stream are ordered, so the slabs of a product are reduced before the next product on that stream overwrites them.  Without a scratch
    xva_gemm's split-K falls back to fp32 atomics (measured 4x slower on the WaveNet weight gradients)."""
    key = (torch.device(device).index or 0, raw_stream(device))
    t = _SK_SCRATCH.get(key)
    if t is None:                          # never re-allocated: prepared call sites keep its address
        t = _SK_SCRATCH[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
    return t


def gemm(A, B, Cm, M, N, K, lda, ldb, ldc, **kw):
    """Thin test/utility wrapper over xva_gemm (keywords: see _fill_gemm). `a_offset` / `b_offset` / `c_offset` (elements) shift the base
    pointers (negative for the overlapping-row conv forms).  Storage dtypes are taken from the tensors."""
    check(lib.xva_gemm(C.byref(_fill_gemm(A, B, Cm, M, N, K, lda, ldb, ldc, **kw)), stream_ptr()), "xva_gemm")


class PreparedGemm:
    """One xva_gemm call site with everything but the operand addresses fixed: the parameter block is filled once (from exemplar tensors) and
    run() patches the pointers — the Python-sequenced paths (xvapitch/wn.py, transformer.py) issue ~1 000 GEMMs per iteration and filling the
    60-field block through ctypes dominated their host time."""
    __slots__ = ("p", "ref", "oa", "ob", "oc")

    def __init__(self, A, B, Cm, *args, a_offset=0, b_offset=0, c_offset=0, **kw):
        self.p = _fill_gemm(A, B, Cm, *args, a_offset=a_offset, b_offset=b_offset, c_offset=c_offset, **kw)
        self.ref = C.byref(self.p)
        self.oa, self.ob, self.oc = A.element_size() * a_offset, B.element_size() * b_offset, Cm.element_size() * c_offset

    def run(self, A, B, Cm, bias=None, R=None, G=None):
        """same dtypes / shapes as the exemplars; bias / R / G only if the call was prepared with them"""
        p = self.p
        p.A, p.B, p.C = A.data_ptr() + self.oa, B.data_ptr() + self.ob, Cm.data_ptr() + self.oc
        if bias is not None:
            p.bias = bias.data_ptr()
        if R is not None:
            p.R = R.data_ptr()
        if G is not None:
            p.G = G.data_ptr()
        if lib.xva_gemm(self.ref, stream_ptr()) != 0:
            check(-1, "xva_gemm")
