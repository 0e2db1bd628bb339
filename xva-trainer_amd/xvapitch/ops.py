"""ctypes wrappers of the xVAPitch-only kernels (csrc/xvapitch_ops.hip) with the reference's function names and tensor layouts:

  maximum_path(value, mask)                       python/xvapitch/util.py:14-53      (GPU; the reference runs numpy on the CPU)
  rand_segments(x, x_lengths, segment_size)       python/xvapitch/util.py:145-165
  segment(x, segment_indices, segment_size)       python/xvapitch/util.py:166-178
  kl_loss(z_p, logs_q, m_p, logs_p, z_mask)       python/xvapitch/losses.py:87-104   (VitsGeneratorLoss.kl_loss)

segment / kl_loss are torch.autograd.Functions whose forward and backward are one C call each; nothing here has a CPU fallback."""
import ctypes as C

import torch

from .. import _lib

lib = _lib.lib
i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
lib.xva_maximum_path_workspace_bytes.restype = i64
lib.xva_maximum_path_workspace_bytes.argtypes = [i32, i32, i32]
lib.xva_maximum_path.restype = i32
lib.xva_maximum_path.argtypes = [vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
lib.xva_segment_fwd.restype = i32
lib.xva_segment_fwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
lib.xva_segment_bwd.restype = i32
lib.xva_segment_bwd.argtypes = [vp, vp, vp, i32, i32, i32, i32, vp]
lib.xva_kl_loss_fwd.restype = i32
lib.xva_kl_loss_fwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]
lib.xva_kl_loss_bwd.restype = i32
lib.xva_kl_loss_bwd.argtypes = [vp, vp, vp, vp, vp, f32, vp, vp, vp, vp, i32, i32, i32, vp]
lib.xva_bct_to_seq.restype = i32
lib.xva_bct_to_seq.argtypes = [vp, vp, i32, i32, i32, i32, i32, vp, vp]
lib.xva_seq_to_bct.restype = i32
lib.xva_seq_to_bct.argtypes = [vp, vp, i32, i32, i32, i32, i32, i32, vp]


def maximum_path(value, mask):
    """value, mask: (B, t_x, t_y); mask must be the outer product of the two length masks (as everywhere in the reference:
    `attn_mask = x_mask.unsqueeze(2) * y_mask.unsqueeze(-1)`).  Returns the 0 / 1 path (B, t_x, t_y) in value's dtype."""
    _lib.require_cuda(value, mask)
    B, t_x, t_y = value.shape
    x_lens = (mask[:, :, 0] != 0).sum(1).to(torch.int32).contiguous()
    y_lens = (mask[:, 0, :] != 0).sum(1).to(torch.int32).contiguous()
    v = value.detach().float().contiguous()
    path = torch.empty(B, t_x, t_y, device=value.device, dtype=torch.float32)
    ws = torch.empty(int(lib.xva_maximum_path_workspace_bytes(B, t_x, t_y)), device=value.device, dtype=torch.uint8)
    _lib.check(lib.xva_maximum_path(_lib.ptr(v), _lib.ptr(x_lens), _lib.ptr(y_lens), _lib.ptr(path), _lib.ptr(ws), ws.numel(), B, t_x, t_y,
                                    _lib.stream_ptr()), "xva_maximum_path")
    return path.to(value.dtype)


class _Segment(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx, S):
        _lib.require_cuda(x, idx)
        B, Cc, T = x.shape
        xc = x.float().contiguous()
        idx = idx.to(device=x.device, dtype=torch.int64).contiguous()
        out = torch.empty(B, Cc, S, device=x.device, dtype=torch.float32)
        _lib.check(lib.xva_segment_fwd(_lib.ptr(xc), _lib.ptr(idx), _lib.ptr(out), B, Cc, T, S, _lib.stream_ptr()), "xva_segment_fwd")
        ctx.save_for_backward(idx)
        ctx.dims = (B, Cc, T, S)
        return out.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        B, Cc, T, S = ctx.dims
        g = g.float().contiguous()
        dx = torch.empty(B, Cc, T, device=g.device, dtype=torch.float32)
        _lib.check(lib.xva_segment_bwd(_lib.ptr(g), _lib.ptr(idx), _lib.ptr(dx), B, Cc, T, S, _lib.stream_ptr()), "xva_segment_bwd")
        return dx, None, None


def segment(x, segment_indices, segment_size=4):
    return _Segment.apply(x, segment_indices, int(segment_size))


DEFERRED_CHECKS = []          # (device bool scalar, message): conditions of this iteration that nobody has read yet — see raise_deferred()


def raise_deferred(values=None):
    """The reference asserts `(max_idxs > 0).all()` in the middle of the forward pass (util.py:165-178), which on a device tensor drains the queue.  Here
    the condition stays on the device (the start index is clamped, so the gather stays inside the tensor either way) and whoever next moves values to the
    host anyway — the trainer's one loss transfer per iteration — reads it with them: `flags()` gives the pending conditions as a float vector to append
    to that transfer, `raise_deferred(host values)` raises the first one that is set.  Called without arguments it reads them itself (one sync)."""
    pending, DEFERRED_CHECKS[:] = list(DEFERRED_CHECKS), []
    if not pending:
        return
    if values is None:
        values = torch.stack([c.float() for c, _ in pending]).cpu()
    for v, (_, msg) in zip(values.tolist(), pending):
        if v:
            raise AssertionError(msg)


def deferred_flags(device):
    """The pending conditions as a float vector (empty when there are none); pass its host copy to raise_deferred()."""
    if not DEFERRED_CHECKS:
        return torch.zeros(0, device=device)
    return torch.stack([c.float().reshape(()) for c, _ in DEFERRED_CHECKS])


def rand_segments(x, x_lengths=None, segment_size=4):
    """Random per-item start indices (torch.rand from the CPU generator, as the reference draws them: `torch.rand([B]).type_as(x)`) + the device gather.
    No host round trip: the draw travels through pinned memory without waiting for the stream, the too-short check is deferred (raise_deferred)."""
    B, _, T = x.size()
    if T < segment_size:                    # shapes are on the host: a gather of S > T rows would leave the tensor (the per-item lengths stay deferred)
        raise AssertionError(" [!] At least one sample is shorter than the segment size.")
    if x_lengths is None:
        x_lengths = torch.full((B,), T, device=x.device)
    max_idxs = x_lengths - segment_size + 1
    if len(DEFERRED_CHECKS) > 64:           # a caller that never reads them (a forward-only script): read now rather than grow without bound
        raise_deferred()
    DEFERRED_CHECKS.append(((max_idxs <= 0).any(), " [!] At least one sample is shorter than the segment size."))
    u = torch.rand([B])
    u = (u.pin_memory() if x.is_cuda else u).to(device=x.device, dtype=x.dtype, non_blocking=True)
    segment_indices = (u * max_idxs.clamp_min(1)).long()
    return segment(x, segment_indices, segment_size), segment_indices


class _KlLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z_p, logs_q, m_p, logs_p, z_mask):
        _lib.require_cuda(z_p, logs_q, m_p, logs_p, z_mask)
        B, H, T = z_p.shape
        t = [a.float().contiguous() for a in (z_p, logs_q, m_p, logs_p)]
        mask = z_mask.float().expand(B, 1, T).contiguous()
        acc = torch.zeros(2, device=z_p.device)
        kl = torch.empty(B, H, T, device=z_p.device)
        _lib.check(lib.xva_kl_loss_fwd(_lib.ptr(t[0]), _lib.ptr(t[1]), _lib.ptr(t[2]), _lib.ptr(t[3]), _lib.ptr(mask), _lib.ptr(kl), _lib.ptr(acc),
                                       B, H, T, _lib.stream_ptr()), "xva_kl_loss_fwd")
        ctx.save_for_backward(t[0], t[2], t[3], mask, acc)
        ctx.mark_non_differentiable(kl)
        return acc[0] / acc[1], kl

    @staticmethod
    def backward(ctx, g_loss, g_kl):
        z_p, m_p, logs_p, mask, acc = ctx.saved_tensors
        B, H, T = z_p.shape
        outs = [torch.empty_like(z_p) for _ in range(4)]
        # the upstream gradient stays on the device (a .item() here drained the whole queue in the middle of the backward pass — 6 ms of a host-bound
        # iteration): the kernel scales by gscale * mask / acc[1], so the gradient is folded into a copy of the denominator
        acc = torch.stack((acc[0], acc[1] / g_loss.float().reshape(())))
        _lib.check(lib.xva_kl_loss_bwd(_lib.ptr(z_p), _lib.ptr(m_p), _lib.ptr(logs_p), _lib.ptr(mask), _lib.ptr(acc), 1.0,
                                       _lib.ptr(outs[0]), _lib.ptr(outs[1]), _lib.ptr(outs[2]), _lib.ptr(outs[3]), B, H, T, _lib.stream_ptr()),
                   "xva_kl_loss_bwd")
        return outs[0], outs[1], outs[2], outs[3], None


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    """Returns (l, kl_sample_wise) like VitsGeneratorLoss.kl_loss."""
    return _KlLoss.apply(z_p, logs_q, m_p, logs_p, z_mask)


def bct_to_seq(x, pad, dtype=torch.float32, lens=None, guard=32):
    """(B, C, T) fp32 -> time-major sequence with `guard` spare rows on both ends (the conv GEMMs read taps before row 0).
    Returns (storage, view (B, T + 2 pad, C))."""
    B, Cc, T = x.shape
    Tp = T + 2 * pad
    store = torch.zeros(2 * guard + B * Tp, Cc, device=x.device, dtype=dtype)
    view = store[guard:guard + B * Tp].view(B, Tp, Cc)
    _lib.check(lib.xva_bct_to_seq(_lib.ptr(x.float().contiguous()), _lib.ptr(view), 1 if dtype == torch.bfloat16 else 0, B, Cc, T, pad,
                                  _lib.ptr(lens), _lib.stream_ptr()), "xva_bct_to_seq")
    return store, view


def seq_to_bct(view, T, pad, into=None):
    """Time-major sequence -> (B, C, T) fp32; `into` (B, C, T) fp32 contiguous: add to it instead of allocating."""
    B, Tp, Cc = view.shape
    out = into if into is not None else torch.empty(B, Cc, T, device=view.device, dtype=torch.float32)
    _lib.check(lib.xva_seq_to_bct(_lib.ptr(view), _lib.ptr(out), 1 if view.dtype == torch.bfloat16 else 0, B, Cc, T, pad, int(into is not None),
                                  _lib.stream_ptr()), "xva_seq_to_bct")
    return out
