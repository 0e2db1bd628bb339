"""xVAPitch's VitsDiscriminator on libxvahip — python/xvapitch/model.py:1590-1640 (nets.0 the scale discriminator of :1548-1587, nets.1-5 the period
discriminators of python/xvapitch/hifigan.py:301-367) with the LSGAN / feature losses of python/xvapitch/losses.py:64-84,331-343, as the two passes
the trainer runs (python/xvapitch/model.py:313-315 generator side, :366-384 discriminator side):

    D = VitsDiscriminator(compute="fp32" | "bf16"); D.load_state_dict(reference_sd)
    loss_disc = D.d_pass(y, y_hat)                    # discriminator_loss(D(y), D(y_hat)); parameter gradients accumulate in D.grads()
    loss_gen, loss_feat, d_wav = D.g_pass(y, y_hat)   # generator_loss + feature_loss x 2 and their gradient w.r.t. y_hat

y, y_hat: (B, seg) or (B, 1, seg) fp32 device tensors, seg a positive multiple of 256.  One C call per forward / backward
(xva_vits_disc_forward / _backward_d / _backward_g); no CPU fallback."""
import ctypes as C

import torch

from .. import _lib

lib = _lib.lib
i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p


class _Dims(C.Structure):
    _fields_ = [("B", i32), ("seg", i32), ("dt", i32)]


lib.xva_vits_disc_param_floats.restype = i64
lib.xva_vits_disc_num_tensors.restype = i32
lib.xva_vits_disc_tensor_info.restype = i32
lib.xva_vits_disc_tensor_info.argtypes = [i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i64 * 4)]
lib.xva_vits_disc_workspace_bytes.restype = i64
lib.xva_vits_disc_workspace_bytes.argtypes = [C.POINTER(_Dims)]
lib.xva_vits_disc_forward.restype = i32
lib.xva_vits_disc_forward.argtypes = [C.POINTER(_Dims), vp, vp, vp, vp, i64, vp, i32, vp]
lib.xva_vits_disc_backward_d.restype = i32
lib.xva_vits_disc_backward_d.argtypes = [C.POINTER(_Dims), vp, vp, vp, vp, vp, i64, vp]
lib.xva_vits_disc_backward_g.restype = i32
lib.xva_vits_disc_backward_g.argtypes = [C.POINTER(_Dims), vp, vp, vp, vp, i32, vp, i64, vp]


class VitsDiscriminator:
    def __init__(self, compute="fp32", device="cuda"):
        self.dt = 1 if compute == "bf16" else 0
        self.device = torch.device(device)
        self.table = {}
        name = C.create_string_buffer(256)
        off, numel, ndim, shape = i64(), i64(), i32(), (i64 * 4)()
        for i in range(lib.xva_vits_disc_num_tensors()):
            _lib.check(lib.xva_vits_disc_tensor_info(i, name, 256, C.byref(off), C.byref(numel), C.byref(ndim), C.byref(shape)), "xva_vits_disc_tensor_info")
            self.table[name.value.decode()] = (off.value, numel.value, tuple(shape[k] for k in range(ndim.value)))
        n = lib.xva_vits_disc_param_floats()
        self.params = torch.zeros(n, device=self.device)
        self.grad = torch.zeros(n, device=self.device)
        self._ws, self._ws_key = None, None
        self._losses = torch.zeros(4, device=self.device)

    def _view(self, flat, k):
        off, numel, shape = self.table[k]
        return flat[off:off + numel].view(shape)

    def state_dict(self):
        return {k: self._view(self.params, k).clone() for k in self.table}

    def load_state_dict(self, sd):
        if set(sd) != set(self.table):
            raise KeyError("VitsDiscriminator.load_state_dict: key mismatch %s" % sorted(set(sd) ^ set(self.table))[:6])
        for k, t in sd.items():
            v = self._view(self.params, k)
            if tuple(t.shape) != tuple(v.shape):
                raise ValueError("%s: shape %s != %s" % (k, tuple(t.shape), tuple(v.shape)))
            v.copy_(t.to(device=self.device, dtype=torch.float32))

    def grads(self):
        return {k: self._view(self.grad, k) for k in self.table}

    def zero_grad(self):
        self.grad.zero_()

    def _prep(self, y, y_hat):
        _lib.require_cuda(y, y_hat)
        y = y.detach().float().reshape(y.size(0), -1).contiguous()
        yh = y_hat.detach().float().reshape(y_hat.size(0), -1).contiguous()
        B, seg = y.shape
        d = _Dims(B, seg, self.dt)
        if self._ws_key != (B, seg):
            n = lib.xva_vits_disc_workspace_bytes(C.byref(d))
            if n <= 0:
                raise ValueError("VitsDiscriminator: " + lib.xva_last_error().decode())
            self._ws = torch.zeros(n, dtype=torch.uint8, device=self.device)          # zero-filled once: pad rows and the off-diagonal blocks of the dense weights
            self._ws_key = (B, seg)
        return y, yh, d

    def _forward(self, y, yh, d, mask):
        _lib.check(lib.xva_vits_disc_forward(C.byref(d), _lib.ptr(self.params), _lib.ptr(y), _lib.ptr(yh), _lib.ptr(self._ws), self._ws.numel(),
                                             _lib.ptr(self._losses), mask, _lib.stream_ptr()), "xva_vits_disc_forward")

    def d_pass(self, y, y_hat):
        y, yh, d = self._prep(y, y_hat)
        self._forward(y, yh, d, 1)
        _lib.check(lib.xva_vits_disc_backward_d(C.byref(d), _lib.ptr(self.params), _lib.ptr(self.grad), _lib.ptr(y), _lib.ptr(yh), _lib.ptr(self._ws),
                                                self._ws.numel(), _lib.stream_ptr()), "xva_vits_disc_backward_d")
        return self._losses[0].clone()

    def g_pass(self, y, y_hat, feature_grad=True):
        """feature_grad=False: d_wav = d loss_gen / d y_hat only — what the reference trainer back-propagates (its feature loss detaches the
        generated features: python/xvapitch/model.py:345-347 with losses.py:64-72); the returned loss values are the same either way."""
        y, yh, d = self._prep(y, y_hat)
        self._forward(y, yh, d, 2)
        d_wav = torch.empty_like(yh)
        _lib.check(lib.xva_vits_disc_backward_g(C.byref(d), _lib.ptr(self.params), _lib.ptr(y), _lib.ptr(yh), _lib.ptr(d_wav), int(bool(feature_grad)),
                                                _lib.ptr(self._ws), self._ws.numel(), _lib.stream_ptr()), "xva_vits_disc_backward_g")
        return self._losses[1].clone(), self._losses[2].clone(), d_wav
