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
LIB_PATH = os.path.join(_HERE, "csrc", "libxvahip.so")


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
        ("seglen", i32),
        ("seg0", i64), ("segstride", i64),
        ("alpha", f32),
        ("bias", vp),
        ("relu", i32),
        ("log_clamp", f32),
        ("R", vp), ("ldr", i64), ("sR", i64),
        ("G", vp), ("ldg", i64), ("sG", i64),
        ("mask_mode", i32),
        ("lens", vp),
        ("Tp", i32),
        ("accumulate", i32),
        ("splitk", i32),
        ("compute", i32),
        ("layout", i32),
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


def stream_ptr(device=None):
    """Raw hipStream_t of torch's current stream (0 = default stream)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise XvaError("libxvahip ops need device tensors; got a %s tensor (no CPU fallback exists)" % t.device)


def gemm(A, B, Cm, M, N, K, lda, ldb, ldc, layout=GEMM_NT, compute=0, batch=1, sA=0, sB=0, sC=0, alpha=1.0,
         bias=None, relu=False, log_clamp=0.0, R=None, ldr=0, sR=0, G=None, ldg=0, sG=0, mask_mode=MASK_NONE,
         lens=None, Tp=0, accumulate=False, splitk=1, seglen=0, seg0=0, segstride=0, a_offset=0):
    """Thin test/utility wrapper over xva_gemm. `a_offset` (elements) shifts the A base pointer
    (negative for the overlapping-row conv form)."""
    require_cuda(A, B, Cm, bias, R, G, lens)
    p = GemmParams()
    p.A = A.data_ptr() + 4 * a_offset
    p.B = B.data_ptr()
    p.C = Cm.data_ptr()
    p.M, p.N, p.K = M, N, K
    p.lda, p.ldb, p.ldc = lda, ldb, ldc
    p.batch, p.sA, p.sB, p.sC = batch, sA, sB, sC
    p.seglen, p.seg0, p.segstride = seglen, seg0, segstride
    p.alpha = alpha
    p.bias = bias.data_ptr() if bias is not None else None
    p.relu = int(relu)
    p.log_clamp = log_clamp
    p.R = R.data_ptr() if R is not None else None
    p.ldr, p.sR = ldr, sR
    p.G = G.data_ptr() if G is not None else None
    p.ldg, p.sG = ldg, sG
    p.mask_mode = mask_mode
    p.lens = lens.data_ptr() if lens is not None else None
    p.Tp = Tp
    p.accumulate = int(accumulate)
    p.splitk = splitk
    p.compute = compute
    p.layout = layout
    check(lib.xva_gemm(C.byref(p), stream_ptr()), "xva_gemm")
