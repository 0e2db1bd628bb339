"""HifiganEngine — host driver of the C-ABI HiFi-GAN engine (xva_hg_* in include/xva_hip.h).

Owns no numerics: allocates the two flat parameter buffers (generator; mpd + msd), their gradient buffers and the
workspace as torch tensors, converts reference state_dicts (python/hifigan/models.py; checkpoint keys of
python/hifigan/xva_train.py:570-601) to / from the flat buffers, and issues the engine calls on torch's current stream.
"""
import ctypes as C

import torch

from .. import _lib

lib = _lib.lib
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class HgDims(C.Structure):
    _fields_ = [("B", i32), ("seg", i32), ("dt", i32)]


lib.xva_hg_param_floats.restype = i64
lib.xva_hg_param_floats.argtypes = [i32]
lib.xva_hg_trainable_floats.restype = i64
lib.xva_hg_trainable_floats.argtypes = [i32]
lib.xva_hg_num_tensors.restype = i32
lib.xva_hg_num_tensors.argtypes = [i32]
lib.xva_hg_tensor_info.restype = i32
lib.xva_hg_tensor_info.argtypes = [i32, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i64 * 4), C.POINTER(i32)]
lib.xva_hg_workspace_bytes.restype = i64
lib.xva_hg_workspace_bytes.argtypes = [C.POINTER(HgDims)]
lib.xva_hg_generator_forward.restype = i32
lib.xva_hg_generator_forward.argtypes = [C.POINTER(HgDims), vp, vp, vp, i64, vp, vp]
lib.xva_hg_generator_backward_ex.restype = i32
lib.xva_hg_generator_backward_ex.argtypes = [C.POINTER(HgDims), vp, vp, vp, vp, i64, C.POINTER(vp), vp]
lib.xva_hg_num_buckets.restype = i32
lib.xva_hg_num_buckets.argtypes = [i32]
lib.xva_hg_bucket_range.restype = i32
lib.xva_hg_bucket_range.argtypes = [i32, i32, C.POINTER(i64), C.POINTER(i64)]

DT = {"fp32": 0, "bf16": 1, 0: 0, 1: 1}
G, D = 0, 1


def tensor_table(which):
    out = []
    buf = C.create_string_buffer(160)
    for i in range(lib.xva_hg_num_tensors(which)):
        off, n, nd, kind = i64(), i64(), i32(), i32()
        shape = (i64 * 4)()
        _lib.check(lib.xva_hg_tensor_info(which, i, buf, 160, C.byref(off), C.byref(n), C.byref(nd), C.byref(shape), C.byref(kind)))
        out.append((buf.value.decode(), off.value, n.value, tuple(shape[k] for k in range(nd.value)), kind.value))
    return out


def to_flat(sd, table, flat, prefix=""):
    with torch.no_grad():
        for name, off, n, shape, kind in table:
            if not name.startswith(prefix):
                continue
            t = sd[name[len(prefix):]]
            if tuple(t.shape) != tuple(shape):
                raise ValueError("%s: checkpoint shape %s != %s" % (name, tuple(t.shape), shape))
            flat[off:off + n].copy_(t.reshape(-1).to(device=flat.device, dtype=torch.float32))


def from_flat(flat, table, prefix=""):
    out = {}
    for name, off, n, shape, kind in table:
        if name.startswith(prefix):
            out[name[len(prefix):]] = flat[off:off + n].view(shape).clone()
    return out


class HifiganEngine:
    def __init__(self, device, compute="bf16"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.XvaError("HifiganEngine needs a GPU device; the hot path has no CPU implementation")
        self.dt = DT[compute]
        self.table = {G: tensor_table(G), D: tensor_table(D)}
        self.total = {G: int(lib.xva_hg_param_floats(G)), D: int(lib.xva_hg_param_floats(D))}
        self.trainable = {G: int(lib.xva_hg_trainable_floats(G)), D: int(lib.xva_hg_trainable_floats(D))}
        self._ws, self._key, self._dims = None, None, None

    def _prepare(self, B, seg):
        key = (B, seg, self.dt)
        if key != self._key:
            d = HgDims(B, seg, self.dt)
            need = int(lib.xva_hg_workspace_bytes(C.byref(d)))
            if need < 0:
                raise _lib.XvaError("xva_hg_workspace_bytes: " + lib.xva_last_error().decode())
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.zeros(need, device=self.device, dtype=torch.uint8)   # zero ONCE: pad / guard rows
            else:
                self._ws.zero_()
            self._dims, self._key = d, key
            self._ws_gen = getattr(self, "_ws_gen", 0) + 1     # the prepared effective weights / norms in the workspace are gone: disc_forward must not skip
            self._d_eff_key = None                             # its reparametrisation pass (A -> B -> A geometries would re-match the old key)
        return self._dims

    def generator_forward(self, flat_g, mel):
        """mel: (B, 80, T) fp32 -> waveform (B, T * 256) fp32."""
        _lib.require_cuda(flat_g, mel)
        B, _, T = mel.shape
        d = self._prepare(B, T * 256)
        mel = mel.float().contiguous()
        wav = torch.empty(B, T * 256, device=self.device, dtype=torch.float32)
        _lib.check(lib.xva_hg_generator_forward(C.byref(d), _lib.ptr(flat_g), _lib.ptr(mel), _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(wav),
                                                _lib.stream_ptr()), "xva_hg_generator_forward")
        return wav

    def generator_backward(self, flat_g, grads_g, d_wav, events=None):
        """events: ctypes array of xva_hg_num_buckets(G) event handles recorded as each gradient bucket completes (DP overlap)."""
        d = self._dims
        d_wav = d_wav.float().contiguous()
        _lib.check(lib.xva_hg_generator_backward_ex(C.byref(d), _lib.ptr(flat_g), _lib.ptr(grads_g), _lib.ptr(d_wav), _lib.ptr(self._ws),
                                                    self._ws.numel(), events, _lib.stream_ptr()), "xva_hg_generator_backward_ex")


SLOT_KINDS = {"mel": 0, "h0": 1, "u": 2, "ua": 3, "xt1": 4, "xr": 5, "xra": 6, "xs": 7, "y": 8, "mpd": 9, "msd": 10}
lib.xva_hg_slot.restype = i32
lib.xva_hg_slot.argtypes = [C.POINTER(HgDims), i32, i32, i32, i32, C.POINTER(i64), C.POINTER(i32)]


def _slot(self, kind, i0=0, i1=0, i2=0):
    """View of an activation tensor the last forward stored: (nseq, T, C) in the activation dtype, structural pad rows cut off
    (test / diagnostics accessor, include/xva_hip.h:xva_hg_slot)."""
    off, geom = i64(), (i32 * 5)()
    _lib.check(lib.xva_hg_slot(C.byref(self._dims), SLOT_KINDS[kind], int(i0), int(i1), int(i2), C.byref(off), geom), "xva_hg_slot")
    nseq, T, Cc, padF, padB = (int(v) for v in geom)
    tdt = torch.bfloat16 if self.dt == DT["bf16"] else torch.float32
    es = 2 if tdt == torch.bfloat16 else 4
    Hp = padF + T + padB
    flat = self._ws[off.value:off.value + nseq * Hp * Cc * es].view(tdt)
    return flat.view(nseq, Hp, Cc)[:, padF:padF + T]


HifiganEngine.slot = _slot


def bucket_ranges(which):
    """[begin, end) float ranges of the gradient buckets of flat buffer `which`, in backward-completion order."""
    out = []
    for i in range(lib.xva_hg_num_buckets(which)):
        b, e = i64(), i64()
        _lib.check(lib.xva_hg_bucket_range(which, i, C.byref(b), C.byref(e)), "xva_hg_bucket_range")
        out.append((b.value, e.value))
    return out


lib.xva_hg_disc_forward_ex.restype = i32
lib.xva_hg_disc_forward_ex.argtypes = [C.POINTER(HgDims), vp, vp, vp, vp, i64, vp, i32, vp]
lib.xva_hg_disc_backward_d_ex.restype = i32
lib.xva_hg_disc_backward_d_ex.argtypes = [C.POINTER(HgDims), vp, vp, vp, vp, vp, i64, C.POINTER(vp), vp]
lib.xva_hg_disc_backward_g.restype = i32
lib.xva_hg_disc_backward_g.argtypes = [C.POINTER(HgDims), vp, vp, vp, vp, vp, i64, vp]


def _disc_forward(self, flat_d, y_real, y_fake, losses="all", weights_token=None):
    """MPD + MSD on (real, fake) (B, seg) fp32.  Returns a 4-float device tensor {loss_disc, loss_gen, loss_fm, -}.
    losses: "all", "d" (discriminator loss only: the D step) or "g" (generator + feature-matching losses: the G step).
    weights_token: anything that changes whenever the caller's optimizer / checkpoint loader writes flat_d (None: no promise).  When it — and the buffer, its
    torch version counter and the workspace — are those of the previous forward, the weight reparametrisation pass is skipped (xva_hg_disc_forward_ex bit 2)."""
    _lib.require_cuda(flat_d, y_real, y_fake)
    d = self._prepare(y_real.size(0), y_real.size(1))
    self._yr, self._yg = y_real.float().contiguous(), y_fake.float().contiguous()
    mask = {"all": 3, "d": 1, "g": 2}[losses]
    key = None if weights_token is None else (weights_token, flat_d.data_ptr(), flat_d._version, self._ws.data_ptr(), self._key, self._ws_gen)
    if key is not None and key == getattr(self, "_d_eff_key", None):
        mask |= 4
    self._d_eff_key = key
    out = torch.zeros(4, device=self.device)
    _lib.check(lib.xva_hg_disc_forward_ex(C.byref(d), _lib.ptr(flat_d), _lib.ptr(self._yr), _lib.ptr(self._yg), _lib.ptr(self._ws), self._ws.numel(),
                                          _lib.ptr(out), mask, _lib.stream_ptr()), "xva_hg_disc_forward_ex")
    return out


def _disc_backward_d(self, flat_d, grads_d, events=None):
    _lib.check(lib.xva_hg_disc_backward_d_ex(C.byref(self._dims), _lib.ptr(flat_d), _lib.ptr(grads_d), _lib.ptr(self._yr), _lib.ptr(self._yg),
                                             _lib.ptr(self._ws), self._ws.numel(), events, _lib.stream_ptr()), "xva_hg_disc_backward_d_ex")


def _disc_backward_g(self, flat_d):
    d_wav = torch.empty_like(self._yg)
    _lib.check(lib.xva_hg_disc_backward_g(C.byref(self._dims), _lib.ptr(flat_d), _lib.ptr(self._yr), _lib.ptr(self._yg), _lib.ptr(d_wav),
                                          _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()), "xva_hg_disc_backward_g")
    return d_wav


HifiganEngine.disc_forward = _disc_forward
HifiganEngine.disc_backward_d = _disc_backward_d
HifiganEngine.disc_backward_g = _disc_backward_g
