"""xVAPitch's waveform decoder on libxvahip: `HifiganGenerator` (python/xvapitch/hifigan.py:156-262) as python/xvapitch/model.py:134-149 builds it
— the HiFi-GAN v1 generator of csrc/hifigan_engine.hip with a latent input (192 channels; 256 for `big`), conv_pre / conv_post without weight
norm, conv_post without bias, and cond_layer(g) added to conv_pre's output.  One C call forward (xva_vits_dec_forward), one backward
(xva_vits_dec_backward: parameter gradients into the flat buffer, d z returned to autograd).

    dec = VitsDecoder(in_channels=192, cond_channels=512, compute="fp32" | "bf16")
    dec.load_state_dict(reference_sd)                 # keys / shapes of HifiganGenerator.state_dict()
    wav = dec(z_slice, g)                             # (B, in, T) , (B, cond, 1) -> (B, 1, T * 256); differentiable w.r.t. z_slice
    dec.grads()                                       # {reference key: gradient view}
"""
import ctypes as C

import torch

from .. import _lib

lib = _lib.lib
i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p


class _Dims(C.Structure):
    _fields_ = [("B", i32), ("seg", i32), ("dt", i32), ("in_channels", i32), ("cond_channels", i32)]


lib.xva_vits_dec_param_floats.restype = i64
lib.xva_vits_dec_param_floats.argtypes = [C.POINTER(_Dims)]
lib.xva_vits_dec_num_tensors.restype = i32
lib.xva_vits_dec_num_tensors.argtypes = [C.POINTER(_Dims)]
lib.xva_vits_dec_tensor_info.restype = i32
lib.xva_vits_dec_tensor_info.argtypes = [C.POINTER(_Dims), i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i64 * 4)]
lib.xva_vits_dec_workspace_bytes.restype = i64
lib.xva_vits_dec_workspace_bytes.argtypes = [C.POINTER(_Dims)]
lib.xva_vits_dec_forward.restype = i32
lib.xva_vits_dec_forward.argtypes = [C.POINTER(_Dims), vp, vp, vp, vp, i64, vp, vp]
lib.xva_vits_dec_backward.restype = i32
lib.xva_vits_dec_backward.argtypes = [C.POINTER(_Dims), vp, vp, vp, vp, vp, vp, i64, vp]


class VitsDecoder:
    def __init__(self, in_channels=192, cond_channels=512, compute="fp32", device="cuda"):
        self.Cin, self.Cc = int(in_channels), int(cond_channels)
        self.dt = 1 if compute == "bf16" else 0
        self.device = torch.device(device)
        d = self._dims(1, 8192)
        n = lib.xva_vits_dec_param_floats(C.byref(d))
        if n <= 0:
            raise ValueError("VitsDecoder: " + lib.xva_last_error().decode())
        self.table = {}
        name = C.create_string_buffer(256)
        off, numel, ndim, shape = i64(), i64(), i32(), (i64 * 4)()
        for i in range(lib.xva_vits_dec_num_tensors(C.byref(d))):
            _lib.check(lib.xva_vits_dec_tensor_info(C.byref(d), i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), C.byref(shape)), "xva_vits_dec_tensor_info")
            self.table[name.value.decode()] = (off.value, numel.value, tuple(shape[k] for k in range(ndim.value)))
        self.params = torch.zeros(n, device=self.device)
        self.grad = torch.zeros(n, device=self.device)
        self._ws, self._ws_key = None, None

    def _dims(self, B, seg):
        return _Dims(B, seg, self.dt, self.Cin, self.Cc)

    def _view(self, flat, k):
        off, numel, shape = self.table[k]
        return flat[off:off + numel].view(shape)

    def state_dict(self):
        return {k: self._view(self.params, k).clone() for k in self.table}

    def load_state_dict(self, sd):
        if set(sd) != set(self.table):
            raise KeyError("VitsDecoder.load_state_dict: key mismatch %s" % sorted(set(sd) ^ set(self.table))[:6])
        for k, t in sd.items():
            v = self._view(self.params, k)
            if tuple(t.shape) != tuple(v.shape):
                raise ValueError("%s: shape %s != %s" % (k, tuple(t.shape), tuple(v.shape)))
            v.copy_(t.to(device=self.device, dtype=torch.float32))

    def grads(self):
        return {k: self._view(self.grad, k) for k in self.table}

    def zero_grad(self):
        self.grad.zero_()

    def workspace(self, B, seg):
        key = (B, seg)
        if self._ws_key != key:
            d = self._dims(B, seg)
            n = lib.xva_vits_dec_workspace_bytes(C.byref(d))
            if n <= 0:
                raise ValueError("VitsDecoder: " + lib.xva_last_error().decode())
            self._ws = torch.zeros(n, dtype=torch.uint8, device=self.device)          # zero-filled once (pad rows of every sequence)
            self._ws_key = key
        return self._ws

    def __call__(self, z, g=None):
        """z (B, in_channels, T) ; g (B, cond_channels, 1) or None -> (B, 1, T * 256)"""
        if self.Cc and g is None:
            raise ValueError("VitsDecoder: built with cond_channels=%d, g is required" % self.Cc)
        hook = torch.zeros(1, device=z.device, requires_grad=True)     # parameter gradients accumulate in backward: it must run even for a data-only z
        return _DecFn.apply(z, g, self, hook)


class _DecFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, g, dec, hook):
        _lib.require_cuda(z)
        B, Cin, T = z.shape
        if Cin != dec.Cin:
            raise ValueError("VitsDecoder: z has %d channels, built for %d" % (Cin, dec.Cin))
        seg = T * 256
        d = dec._dims(B, seg)
        ws = dec.workspace(B, seg)
        zc = z.detach().float().contiguous()
        gc = g.detach().float().reshape(B, dec.Cc).contiguous() if dec.Cc else None
        wav = torch.empty(B, seg, device=z.device)
        _lib.check(lib.xva_vits_dec_forward(C.byref(d), _lib.ptr(dec.params), _lib.ptr(zc), _lib.ptr(gc), _lib.ptr(ws), ws.numel(), _lib.ptr(wav),
                                            _lib.stream_ptr()), "xva_vits_dec_forward")
        ctx.dec, ctx.dims, ctx.gc = dec, (B, Cin, T), gc
        return wav.unsqueeze(1)

    @staticmethod
    def backward(ctx, d_wav):
        dec = ctx.dec
        B, Cin, T = ctx.dims
        d = dec._dims(B, T * 256)
        ws = dec.workspace(B, T * 256)
        dw = d_wav.float().reshape(B, T * 256).contiguous()
        dz = torch.empty(B, Cin, T, device=dw.device)
        _lib.check(lib.xva_vits_dec_backward(C.byref(d), _lib.ptr(dec.params), _lib.ptr(dec.grad), _lib.ptr(ctx.gc), _lib.ptr(dw), _lib.ptr(dz), _lib.ptr(ws),
                                             ws.numel(), _lib.stream_ptr()), "xva_vits_dec_backward")
        return dz, None, None, None
