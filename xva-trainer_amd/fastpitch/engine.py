"""FastPitchEngine — thin host driver of the C-ABI FastPitch engine (xva_fp_* in include/xva_hip.h).

Owns no numerics: it allocates the flat parameter / gradient buffers and the workspace as torch tensors
(device memory is the caller's, per the ABI), converts the reference's batch format to the ABI's plain int32 /
fp32 arrays, and issues forward / loss / backward as three C calls on torch's current stream.
"""
import ctypes as C

import torch

from .. import _lib

lib = _lib.lib
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class FpDims(C.Structure):
    _fields_ = [("B", i32), ("Tt", i32), ("Tm", i32), ("stage", i32), ("compute", i32), ("p_dropout", f32), ("seed", C.c_uint64)]


class FpBatch(C.Structure):
    _fields_ = [("text", vp), ("in_lens", vp), ("durs", vp), ("pitch", vp), ("energy", vp), ("pos_table", vp)]


SLOTS = ["MEL_OUT", "PITCH_PRED", "ENERGY_PRED", "LOG_DUR_PRED", "DUR_PRED", "PITCH_TGT", "ENERGY_TGT", "DEC_LENS", "LOSS_ACC",
         "LOSSES", "D_MEL", "D_PITCH", "D_ENERGY", "D_LOGDUR", "ENC_OUT", "DEC_OUT", "ENC_COND"]
SLOT = {n: i for i, n in enumerate(SLOTS)}

lib.xva_fp_param_floats.restype = i64
lib.xva_fp_num_tensors.restype = i32
lib.xva_fp_tensor_info.restype = i32
lib.xva_fp_tensor_info.argtypes = [i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32), C.POINTER(i64 * 4), C.POINTER(i32)]
lib.xva_fp_trainable_ranges.restype = i32
lib.xva_fp_trainable_ranges.argtypes = [i32, C.POINTER(i64), C.POINTER(i64), i32]
lib.xva_fp_workspace_bytes.restype = i64
lib.xva_fp_workspace_bytes.argtypes = [C.POINTER(FpDims)]
lib.xva_fp_slot_offset.restype = i32
lib.xva_fp_slot_offset.argtypes = [C.POINTER(FpDims), i32, C.POINTER(i64)]
lib.xva_fp_forward.restype = i32
lib.xva_fp_forward.argtypes = [C.POINTER(FpDims), vp, C.POINTER(FpBatch), vp, i64, vp]
lib.xva_fp_backward.restype = i32
lib.xva_fp_backward.argtypes = [C.POINTER(FpDims), vp, vp, C.POINTER(FpBatch), vp, i64, vp]
lib.xva_fp_loss_denominators.restype = i32
lib.xva_fp_loss_denominators.argtypes = [i32, vp, vp, vp, i32, i32, i32, vp]
lib.xva_fp_loss_partials.restype = i32
lib.xva_fp_loss_partials.argtypes = [i32, i32] + [vp] * 10 + [i32, i32, i32, vp]
lib.xva_fp_loss_grads.restype = i32
lib.xva_fp_loss_grads.argtypes = [i32, i32] + [vp] * 15 + [i32, i32, i32, f32, f32, f32, f32, vp]

class FpAlignBatch(C.Structure):
    _fields_ = [("text", vp), ("in_lens", vp), ("mel", vp), ("mel_lens", vp), ("attn_prior", vp)]


lib.xva_fp_align_workspace_bytes.restype = i64
lib.xva_fp_align_workspace_bytes.argtypes = [C.POINTER(FpDims)]
lib.xva_fp_align_forward.restype = i32
lib.xva_fp_align_forward.argtypes = [C.POINTER(FpDims), vp, C.POINTER(FpAlignBatch), vp, i64, vp, vp, vp, vp, vp]
lib.xva_fp_align_backward.restype = i32
lib.xva_fp_align_backward.argtypes = [C.POINTER(FpDims), vp, vp, C.POINTER(FpAlignBatch), vp, i64, f32, vp]
lib.xva_fp_infer_encode.restype = i32
lib.xva_fp_infer_encode.argtypes = [C.POINTER(FpDims), vp, C.POINTER(FpBatch), f32, f32, vp, i64, vp, vp, vp, vp, vp, vp, vp]
lib.xva_fp_infer_decode.restype = i32
lib.xva_fp_infer_decode.argtypes = [C.POINTER(FpDims), vp, vp, vp, vp, vp, i64, vp, vp]

# "f16" (round 6): IEEE-half MFMA operands (v_mfma_f32_16x16x32_f16) over an fp32 residual stream — the storage plan of "fp32" with every operand copy a
# single half tensor; the cheapest format found whose OUTPUTS stay within 1e-3 of the fp32 reference (profiles/r06_precision_probe.txt).  Its activation
# gradients live in fp16 buffers too, so the loss gradient is multiplied by `FastPitchEngine.loss_scale` (a power of two, as the reference's GradScaler does
# for its fp16 autocast path: python/fastpitch1_1/xva_train.py:350,856-859): flat_grads then hold loss_scale x gradient — pass `inv_scale = 1 / loss_scale`
# to Lamb.step (which unscales before clipping) or read `unscaled(flat_grads)`.
COMPUTE = {"fp32": 0, "bf16": 1, "f16": 2, 0: 0, 1: 1, 2: 2}
ACT_SLOTS = {"MEL_OUT", "D_MEL", "ENC_OUT", "DEC_OUT", "ENC_COND"}   # stored in the activation dtype (bf16 when compute == bf16)


def tensor_table():
    """[(name, offset, numel, reference_shape, kind)] straight from the library (single source of truth)."""
    out = []
    buf = C.create_string_buffer(128)
    for i in range(lib.xva_fp_num_tensors()):
        off, n, nd, kind = i64(), i64(), i32(), i32()
        shape = (i64 * 4)()
        _lib.check(lib.xva_fp_tensor_info(i, buf, 128, C.byref(off), C.byref(n), C.byref(nd), C.byref(shape), C.byref(kind)))
        out.append((buf.value.decode(), off.value, n.value, tuple(shape[k] for k in range(nd.value)), kind.value))
    return out


def trainable_ranges(stage):
    b, e = (i64 * 16)(), (i64 * 16)()
    n = lib.xva_fp_trainable_ranges(int(stage), b, e, 16)
    return [(b[k], e[k]) for k in range(n)]


def positional_table(T, d_model=384, device="cpu", dtype=torch.float32):
    """PositionalEmbedding (transformer.py:21-35), computed with the same torch ops as the reference module buffer."""
    inv_freq = (1 / (10000 ** (torch.arange(0.0, d_model, 2.0) / d_model))).to(device)   # module buffer: built on the CPU, moved
    pos_seq = torch.arange(T, device=device).to(dtype)
    sinusoid = torch.matmul(pos_seq.unsqueeze(-1), inv_freq.unsqueeze(0))
    return torch.cat([sinusoid.sin(), sinusoid.cos()], dim=1).contiguous()


class DeviceBatch:
    """ABI-shaped batch living on the device (contiguous int32 / fp32)."""

    def __init__(self, text, in_lens, mel_tgt=None, mel_lens=None, pitch=None, energy=None, durs=None):
        dev = text.device
        self.text = text.to(torch.int32).contiguous()
        self.in_lens = in_lens.to(device=dev, dtype=torch.int32).contiguous()
        self.B, self.Tt = self.text.shape
        self.mel_tgt = mel_tgt.float().contiguous() if mel_tgt is not None else None
        self.mel_lens = mel_lens.to(device=dev, dtype=torch.int32).contiguous() if mel_lens is not None else None
        self.Tm = int(self.mel_tgt.size(2)) if mel_tgt is not None else 1
        self.pitch = pitch.float().reshape(self.B, -1).contiguous() if pitch is not None else None
        self.energy = energy.float().contiguous() if energy is not None else None
        self.durs = durs.to(torch.int32).contiguous() if durs is not None else None
        self.num_frames = None
        self.attn_prior = None

    @staticmethod
    def from_dict(b, device):
        g = lambda k: b[k].to(device, non_blocking=True) if b.get(k) is not None else None
        db = DeviceBatch(g("text"), g("in_lens"), g("mel_tgt"), g("mel_lens"), g("pitch"), g("energy"), g("durs"))
        db.attn_prior = g("attn_prior")          # (B, Tm, Tt) beta-binomial prior: training stage 1 only
        return db


class FastPitchEngine:
    def __init__(self, device, compute="bf16", p_dropout=0.0, seed=1234):
        """p_dropout: the reference trains with 0.1 on every dropout site (model.py:248-263 defaults; nn.Module.train()); 0
        gives the deterministic network the parity tests pin.  Masks are a pure function of (seed + step, site, index)."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.XvaError("FastPitchEngine needs a GPU device; the hot path has no CPU implementation")
        self.compute = COMPUTE[compute]
        self.act_dt = 1 if self.compute == 1 else 0          # XVA_BF16 / XVA_F32: the dtype of the workspace's activation slots
        self.act_dtype = torch.bfloat16 if self.compute == 1 else torch.float32
        self.loss_scale = 1.0                                 # "f16" only: see set_loss_scale / auto_loss_scale
        self.auto_loss_scale = self.compute == 2
        self.p_dropout = float(p_dropout)
        self.seed = int(seed)
        self.step = 0
        self.table = tensor_table()
        self.total = int(lib.xva_fp_param_floats())
        self._ws = None
        self._ws_key = None
        self._dims = None
        self._pos = None
        self._slot_off = {}

    # -- loss scale of the fp16-operand mode -------------------------------------------------
    def set_loss_scale(self, scale):
        """Fix the loss scale (a power of two) and stop choosing it per batch."""
        import math
        m, _ = math.frexp(float(scale))
        if m != 0.5 or scale <= 0:
            raise ValueError("loss scale must be a positive power of two (exact in every format)")
        self.loss_scale, self.auto_loss_scale = float(scale), False

    def _choose_loss_scale(self, b):
        """The scale that brings the mel loss's seed gradient 2 (mel_out - mel_tgt) / #(mel_tgt != 0) to O(1): the power of two next to the (upper bound of
        the) denominator B * Tm * 80, from the batch geometry alone — no device read-back — and 2^-4 of it.  Measured on the oracle (profiles/r06_precision_probe.txt): at the full
        denominator the stored activation gradients peak near 2^7 with the median near 2, against fp16's 2^-14 .. 2^16 normal range; measured on the
        GPU at B = 32 x 860 (tools/f16_scale_probe.py): the gradient is the same to 1e-6 for every scale from 2^9 to 2^21 and 6e-3 off at scale 1."""
        import math
        return float(2 ** max(0, int(round(math.log2(max(1, b.B * b.Tm * 80)))) - 4))      # 2^-4: headroom for the token-level losses' larger seeds

    @property
    def grad_inv_scale(self):
        return 1.0 / self.loss_scale

    def unscaled(self, flat_grads):
        return flat_grads if self.loss_scale == 1.0 else flat_grads * (1.0 / self.loss_scale)

    # -- workspace ---------------------------------------------------------------------
    def _prepare(self, B, Tt, Tm, stage):
        key = (B, Tt, Tm, int(stage), self.compute, self.p_dropout, int(lib.xva_fp_plan_knobs()))     # (the plan also depends on two process-global switches)
        if key != self._ws_key:
            d = FpDims(B, Tt, Tm, int(stage), self.compute, self.p_dropout, self.seed)
            need = int(lib.xva_fp_workspace_bytes(C.byref(d)))
            if need < 0:
                raise _lib.XvaError("xva_fp_workspace_bytes: " + lib.xva_last_error().decode())
            if self._ws is None or self._ws.numel() < need:
                self._ws = torch.zeros(need, device=self.device, dtype=torch.uint8)  # zero ONCE: guard rows
            else:
                self._ws.zero_()   # a new geometry moves the structural-zero rows
            self._dims = d
            self._ws_key = key
            self._slot_off = {}
            for name, sid in SLOT.items():
                off = i64()
                _lib.check(lib.xva_fp_slot_offset(C.byref(d), sid, C.byref(off)), "xva_fp_slot_offset")
                self._slot_off[name] = off.value
            tmax = max(Tt, Tm)
            if self._pos is None or self._pos.size(0) < tmax:
                self._pos = positional_table(max(tmax, 1024), device=self.device)
        return self._dims

    def slot(self, name, shape, dtype=None):
        """View of a workspace slot (no copy).  Activation slots carry the engine's activation dtype."""
        if dtype is None:
            dtype = self.act_dtype if name in ACT_SLOTS else torch.float32
        off = self._slot_off[name]
        n = 1
        for s in shape:
            n *= s
        nb = n * torch.empty((), dtype=dtype).element_size()
        return self._ws[off:off + nb].view(dtype).view(*shape)

    def layer_input(self, stack, l, B, T):
        """Stored input of layer l (l = 6: the stack output) of the "encoder" / "decoder" stack after a forward: (B, T, 384) view in the
        activation dtype without the two structural pad rows (slots 100 + l / 200 + l of xva_fp_slot_offset)."""
        off = i64()
        _lib.check(lib.xva_fp_slot_offset(C.byref(self._dims), (100 if stack == "encoder" else 200) + int(l), C.byref(off)), "xva_fp_slot_offset")
        nb = B * (T + 2) * 384 * (2 if self.compute == 1 else 4)
        return self._ws[off.value:off.value + nb].view(self.act_dtype).view(B, T + 2, 384)[:, 1:T + 1]

    def _abi_batch(self, b):
        return FpBatch(_lib.ptr(b.text), _lib.ptr(b.in_lens), _lib.ptr(b.durs), _lib.ptr(b.pitch), _lib.ptr(b.energy), _lib.ptr(self._pos))

    # -- the three calls ---------------------------------------------------------------
    def forward(self, flat_params, b, stage):
        d = self._prepare(b.B, b.Tt, b.Tm, stage)
        d.seed = (self.seed + 0x9E3779B97F4A7C15 * self.step) & 0xFFFFFFFFFFFFFFFF   # backward regenerates the same masks
        self.step += 1
        self._abi = self._abi_batch(b)
        _lib.check(lib.xva_fp_forward(C.byref(d), _lib.ptr(flat_params), C.byref(self._abi), _lib.ptr(self._ws), self._ws.numel(),
                                      _lib.stream_ptr()), "xva_fp_forward")

    def _loss_args(self, b):
        sp = lambda n: C.c_void_p(self._ws.data_ptr() + self._slot_off[n])
        return [sp("MEL_OUT"), _lib.ptr(b.mel_tgt), sp("PITCH_PRED"), sp("PITCH_TGT"), sp("ENERGY_PRED"), sp("ENERGY_TGT"),
                sp("LOG_DUR_PRED"), _lib.ptr(b.durs), _lib.ptr(b.in_lens)], sp

    def loss_denominators(self, b, stage):
        """(2,) fp32: the masked means' denominators [#(mel_tgt != 0), sum(in_lens)] from the batch alone (before / beside the forward)."""
        den = torch.empty(2, device=self.device)
        _lib.check(lib.xva_fp_loss_denominators(int(stage), _lib.ptr(b.mel_tgt), _lib.ptr(b.in_lens), _lib.ptr(den), b.B, b.Tt, b.Tm, _lib.stream_ptr()),
                   "xva_fp_loss_denominators")
        return den

    def loss_partials(self, b, stage):
        args, sp = self._loss_args(b)
        _lib.check(lib.xva_fp_loss_partials(int(stage), self.act_dt, *args, sp("LOSS_ACC"), b.B, b.Tt, b.Tm, _lib.stream_ptr()), "xva_fp_loss_partials")
        return self.slot("LOSS_ACC", (8,))

    def loss_grads(self, b, stage, grad_scale=1.0, dur_w=0.1, pitch_w=0.1, energy_w=0.1):
        args, sp = self._loss_args(b)
        if self.auto_loss_scale:
            self.loss_scale = self._choose_loss_scale(b)
        _lib.check(lib.xva_fp_loss_grads(int(stage), self.act_dt, *args, sp("LOSS_ACC"), sp("LOSSES"), sp("D_MEL"), sp("D_PITCH"), sp("D_ENERGY"),
                                         sp("D_LOGDUR"), b.B, b.Tt, b.Tm, grad_scale * self.loss_scale, dur_w, pitch_w, energy_w, _lib.stream_ptr()),
                   "xva_fp_loss_grads")
        return self.slot("LOSSES", (8,))

    def backward(self, flat_params, flat_grads, b, stage):
        d = self._prepare(b.B, b.Tt, b.Tm, stage)
        _lib.check(lib.xva_fp_backward(C.byref(d), _lib.ptr(flat_params), _lib.ptr(flat_grads), C.byref(self._abi), _lib.ptr(self._ws),
                                       self._ws.numel(), _lib.stream_ptr()), "xva_fp_backward")

    def fwd_loss_bwd(self, flat_params, flat_grads, b, stage, grad_scale=1.0, reduce_acc=None):
        """One micro-batch: forward, loss (optionally all-reducing the loss numerators/denominators across DP ranks through
        `reduce_acc(acc_tensor)`), backward accumulating into flat_grads. Returns the 8-float device tensor of losses."""
        self.forward(flat_params, b, stage)
        acc = self.loss_partials(b, stage)
        if reduce_acc is not None:
            reduce_acc(acc)
        losses = self.loss_grads(b, stage, grad_scale)
        self.backward(flat_params, flat_grads, b, stage)
        return losses

    # -- training stage 1: the aligner (model.py:296-323,346-360) --------------------------
    def align_forward(self, flat_params, text, in_lens, mel_tgt, mel_lens, attn_prior, want_maps=True):
        """Returns (loss (1,), attn_hard_dur (B, Tt) int32, attn_soft, attn_logprob (B, 1, Tm, Tt) or None).  The workspace keeps what
        align_backward needs."""
        dev = self.device
        text = text.to(torch.int32).contiguous()
        B, Tt = text.shape
        mel = mel_tgt.float().contiguous()
        Tm = mel.size(2)
        self._al = dict(text=text, in_lens=in_lens.to(device=dev, dtype=torch.int32).contiguous(), mel=mel,
                        mel_lens=mel_lens.to(device=dev, dtype=torch.int32).contiguous(), prior=attn_prior.float().contiguous())
        a = self._al
        d = FpDims(B, Tt, Tm, 1, self.compute, 0.0, 0)
        need = int(lib.xva_fp_align_workspace_bytes(C.byref(d)))
        if need < 0:
            raise _lib.XvaError("xva_fp_align_workspace_bytes: " + lib.xva_last_error().decode())
        key = (B, Tt, Tm, self.compute)
        if getattr(self, "_al_ws", None) is None or self._al_ws.numel() < need:
            self._al_ws = torch.zeros(need, device=dev, dtype=torch.uint8)
        elif getattr(self, "_al_key", None) != key:
            self._al_ws.zero_()        # a new geometry moves the structural-zero (guard / padding) rows: stale rows of the previous batch are not zeros
        self._al_key = key
        self._al_dims = d
        self._al_bt = FpAlignBatch(_lib.ptr(a["text"]), _lib.ptr(a["in_lens"]), _lib.ptr(a["mel"]), _lib.ptr(a["mel_lens"]), _lib.ptr(a["prior"]))
        loss = torch.zeros(1, device=dev)
        durs = torch.zeros(B, Tt, device=dev, dtype=torch.int32)
        soft = torch.empty(B, 1, Tm, Tt, device=dev) if want_maps else None
        logp = torch.empty(B, 1, Tm, Tt, device=dev) if want_maps else None
        _lib.check(lib.xva_fp_align_forward(C.byref(d), _lib.ptr(flat_params), C.byref(self._al_bt), _lib.ptr(self._al_ws), self._al_ws.numel(),
                                            _lib.ptr(soft), _lib.ptr(logp), _lib.ptr(durs), _lib.ptr(loss), _lib.stream_ptr()), "xva_fp_align_forward")
        return loss, durs, soft, logp

    def align_backward(self, flat_params, flat_grads, grad_scale=1.0):
        _lib.check(lib.xva_fp_align_backward(C.byref(self._al_dims), _lib.ptr(flat_params), _lib.ptr(flat_grads), C.byref(self._al_bt),
                                             _lib.ptr(self._al_ws), self._al_ws.numel(), float(grad_scale), _lib.stream_ptr()), "xva_fp_align_backward")

    # -- inference (FastPitch.infer, model.py:426-481) -----------------------------------
    def infer(self, flat_params, text, in_lens, pace=1.0, max_duration=75.0):
        """text (B, Tt) int, in_lens (B).  Returns mel_out (B, 80, Tm) fp32, dec_lens (B) int64, dur_pred, pitch_pred (B, 1, Tt),
        energy_pred (B, Tt).  Two C calls: the mel length is data dependent, so dec_lens crosses to the host in between."""
        keep = self.p_dropout
        self.p_dropout = 0.0
        try:
            text = text.to(torch.int32).contiguous()
            in_lens = in_lens.to(device=self.device, dtype=torch.int32).contiguous()
            B, Tt = text.shape
            d = self._prepare(B, Tt, 1, 3)
            dev = self.device
            enc_cond = torch.empty(B, Tt + 2, 384, device=dev, dtype=self.act_dtype)
            durs = torch.empty(B, Tt, device=dev, dtype=torch.int32)
            dec_lens = torch.empty(B, device=dev, dtype=torch.int32)
            dur, pitch, energy = (torch.empty(B, Tt, device=dev) for _ in range(3))
            bt = FpBatch(_lib.ptr(text), _lib.ptr(in_lens), None, None, None, _lib.ptr(self._pos))
            _lib.check(lib.xva_fp_infer_encode(C.byref(d), _lib.ptr(flat_params), C.byref(bt), float(pace), float(max_duration), _lib.ptr(self._ws),
                                               self._ws.numel(), _lib.ptr(enc_cond), _lib.ptr(durs), _lib.ptr(dec_lens), _lib.ptr(dur), _lib.ptr(pitch),
                                               _lib.ptr(energy), _lib.stream_ptr()), "xva_fp_infer_encode")
            Tm = max(int(dec_lens.max().item()), 1)            # host sync: sizes the decoder pass
            d = self._prepare(B, Tt, Tm, 3)
            mel = torch.empty(B, 80, Tm, device=dev)
            _lib.check(lib.xva_fp_infer_decode(C.byref(d), _lib.ptr(flat_params), _lib.ptr(enc_cond), _lib.ptr(durs), _lib.ptr(self._pos),
                                               _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(mel), _lib.stream_ptr()), "xva_fp_infer_decode")
            return mel, dec_lens.long(), dur, pitch.unsqueeze(1), energy
        finally:
            self.p_dropout = keep

    # -- views of outputs in the reference's shapes --------------------------------------
    def outputs(self, b, stage):
        B, Tt, Tm = b.B, b.Tt, b.Tm
        o = {}
        if stage == 2:
            o["log_dur_pred"] = self.slot("LOG_DUR_PRED", (B, Tt + 2))[:, 1:Tt + 1]
            o["dur_pred"] = self.slot("DUR_PRED", (B, Tt + 2))[:, 1:Tt + 1]
            return o
        o["mel_out"] = self.slot("MEL_OUT", (B, Tm + 2, 80))[:, 1:Tm + 1]
        o["pitch_pred"] = self.slot("PITCH_PRED", (B, Tt + 2))[:, 1:Tt + 1].unsqueeze(1)
        o["pitch_tgt"] = self.slot("PITCH_TGT", (B, Tt + 2))[:, 1:Tt + 1].unsqueeze(1)
        o["energy_pred"] = self.slot("ENERGY_PRED", (B, Tt + 2))[:, 1:Tt + 1]
        o["energy_tgt"] = self.slot("ENERGY_TGT", (B, Tt + 2))[:, 1:Tt + 1]
        o["dec_lens"] = self.slot("DEC_LENS", (B,), torch.int32)
        return o
