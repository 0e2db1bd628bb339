"""xVAPitch's generator pass WITHOUT the adversarial terms, on libxvahip: the whole of `xVAPitch.train_step` (python/xvapitch/model.py:681-870) —
acoustic path (acoustic.py), `rand_segments` of the posterior latent (:850), the waveform decoder on the slice (:852, decoder.py), the matching
`segment` of the recording (:856-860) — and the losses of VitsGeneratorLoss.forward that do not need the discriminator
(python/xvapitch/losses.py): mel L1 x 45 on TorchSTFT mels of the two segments (:187-193), KL (:213-218), duration (:220), pitch (:224-241).
The discriminator-side terms (generator / feature loss :195-196, the VitsDiscriminator pass) are not built: xVAPitch's scale discriminator
(model.py:1548-1587) has no engine variant yet.

    gp = GeneratorPass(AcousticTrainPath(...), VitsDecoder(latent, d_vector_dim), spec_segment_size=32)
    out = gp(tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=...)
    out["loss"].backward()        # gradients: gp.acoustic.grads(), gp.decoder.grads()
"""
import os

import torch
import torch.nn.functional as F

from .. import mel as xmel
from . import ops


class _MelL1(torch.autograd.Function):
    """loss = scale * l1_loss(mel_tgt, mel(y_hat)) with its gradient from the same C call (xva_mel_l1_loss_backward, M3 configuration)."""
    @staticmethod
    def forward(ctx, y_hat, mel_tgt, stft, scale):
        y = y_hat.detach().float().reshape(y_hat.size(0), -1).contiguous()
        d_wav = torch.empty_like(y)
        loss, _ = stft.l1_loss_backward(y, mel_tgt.contiguous(), d_wav, scale=scale, accumulate=False)
        ctx.save_for_backward(d_wav)
        ctx.shape = tuple(y_hat.shape)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (d_wav,) = ctx.saved_tensors
        return (d_wav * g).view(ctx.shape), None, None, None


class _JoinBranch(torch.autograd.Function):
    """loss = (the acoustic path's terms) + (the vocoder branch's terms), where the branch hangs off a DETACHED copy of z (GeneratorPass.__call__).
    backward: the branch's whole backward pass is issued first, on the branch's stream, from inside this node (a nested autograd call) — it is the
    GPU-heavy, host-cheap half (two engine calls), so the device works through it while the host issues the acoustic modules' backward on the main
    stream; d loss / d z of the branch joins the main graph in a hook on z (event wait + add) right before the posterior encoder's backward."""
    @staticmethod
    def forward(ctx, main_loss, branch_value, state):
        ctx.state = state
        if branch_value is None:                                # late_join: the VALUE is written on the branch's stream (GeneratorPass.__call__)
            return torch.empty_like(main_loss)
        return main_loss + branch_value

    @staticmethod
    def backward(ctx, g):
        st = ctx.state
        side, cur = st["stream"], torch.cuda.current_stream()
        side.wait_stream(cur)                                   # g
        g.record_stream(side)
        with torch.cuda.stream(side):                           # the caller's stream of the nested pass = the branch's: no join at its end
            torch.autograd.backward(st["loss"], g)
        st["event"] = side.record_event()
        st["loss"] = None
        if side != cur:                                         # whatever happens to z's hook, backward() returns with the branch joined
            torch.autograd.Variable._execution_engine.queue_callback(lambda: torch.cuda.current_stream().wait_event(st["event"]))
        return g, None, None


class GeneratorPass:
    def __init__(self, acoustic, decoder, spec_segment_size=32, mel_loss_alpha=45.0):
        self.acoustic, self.decoder = acoustic, decoder
        self.S, self.alpha = int(spec_segment_size), float(mel_loss_alpha)                 # model.py:77, losses.py:27
        self.stft = xmel.TorchSTFTMel(1024, 256, 1024, sample_rate=22050, mel_fmin=0.0, mel_fmax=8000.0, n_mels=80)    # losses.py:29-46
        self._side = {}

    def branch_stream(self, device):
        """The vocoder branch's stream (XVA_C5_BRANCH_STREAM=0: the current one — same code path, no concurrency)."""
        if os.environ.get("XVA_C5_BRANCH_STREAM", "1") == "0":
            return torch.cuda.current_stream(device)
        key = torch.device(device).index or 0
        if key not in self._side:
            self._side[key] = torch.cuda.Stream(device)
        return self._side[key]

    def join(self, device=None):
        """the current stream waits for the vocoder branch (values of a late_join forward pass read before / without the backward pass)"""
        cur = torch.cuda.current_stream(device)
        side = self.branch_stream(cur.device)
        if side != cur:
            cur.wait_stream(side)

    def zero_grad(self):
        self.acoustic.zero_grad()
        self.decoder.zero_grad()

    def batch_from_wav(self, wavs, wav_lengths):
        """On-device counterpart of the reference's dataset + collate for the two audio tensors of a batch (python/xvapitch/dataset.py:251,470-490):
        wavs (B, Nmax) zero-padded float clips, wav_lengths (B,) -> y (B, 513, Ty) per-clip linear spectrograms (zeros after each clip's frames),
        y_lengths (B,) = 1 + wav_lengths // 256, waveform (B, 1, Ty * 256) zero-padded to the frame grid (`max(mel_lengths) * hop_length`)."""
        y, y_lengths = self.stft.linear_ragged(wavs, wav_lengths)
        Ty = y.size(2)
        waveform = F.pad(wavs.float(), (0, Ty * 256 - wavs.size(1))).unsqueeze(1)
        return y, y_lengths, waveform

    def __call__(self, tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=None, eps=None, noise=None, slice_ids=None, tail=None,
                 late_join=False):
        """waveform (B, 1, Ty * 256).  slice_ids (B,): the segment starts (drawn like the reference's rand_segments when None).
        tail(model_outputs, waveform_seg) -> {name: loss}: further terms on the decoder's output (train_step: the adversarial ones), run inside the branch.

        Two streams.  Everything after the posterior encoder splits into two independent halves: the vocoder branch (segment of z -> waveform decoder ->
        mel / adversarial terms: a few engine calls, ~40 % of the iteration's device time, almost no host time) and the rest of the acoustic path (text
        encoder, flow, alignment, duration / pitch predictors, KL: thousands of small launches, host-bound).  On one stream the device idles through the
        second while the first waits its turn; here the branch runs on its own stream from the moment z exists, forward and backward (_JoinBranch).
        late_join: the caller's stream does NOT wait for the branch when the forward pass ends (with the discriminator pass inside it the branch is the
        longer half: the acoustic path's backward would wait 3 ms for tensors it never reads).  The branch's tensors in `out` — model_outputs, waveform_seg,
        its loss terms — and the VALUE of out["loss"] (summed on the branch's stream) are then complete only after out["loss"].backward(), which returns
        joined, or after join()."""
        dev = y.device
        main, side = torch.cuda.current_stream(dev), self.branch_stream(dev)
        st = {"stream": side}
        S = self.S

        def branch(z):
            side.wait_stream(main)
            zb = z.detach().requires_grad_(z.requires_grad)
            for t in (z, waveform, d_vectors, y_lengths) + ((slice_ids,) if slice_ids is not None else ()):
                if t.is_cuda:
                    t.record_stream(side)
            with torch.cuda.stream(side):
                g = F.normalize(d_vectors.float()).unsqueeze(-1)
                if slice_ids is None:
                    z_slice, ids = ops.rand_segments(zb, y_lengths.to(zb.device), S)                          # :850
                else:
                    z_slice, ids = ops.segment(zb, slice_ids, S), slice_ids
                o = self.decoder(z_slice, g)                                                                  # :852
                wav_seg = ops.segment(waveform.float(), ids * 256, S * 256)                                   # :856-860
                with torch.no_grad():
                    mel_tgt = self.stft(wav_seg)
                terms = {"loss_mel": _MelL1.apply(o, mel_tgt, self.stft, self.alpha)}                         # losses.py:187-193
                if tail is not None:
                    terms.update(tail(o, wav_seg))
                total = sum(terms.values())
            st.update(zb=zb, loss=total, terms=terms, o=o, wav_seg=wav_seg, ids=ids, z_slice=z_slice)

        out = self.acoustic(tokens, x_lengths, y, y_lengths, d_vectors, language_ids, eps=eps, noise=noise, pitch_padded=pitch_padded, after_posterior=branch)
        late = bool(late_join) and side != main and st["zb"].requires_grad
        if not late:
            main.wait_stream(side)                               # the caller reads the branch's tensors on its own stream
        for t in [st["o"], st["wav_seg"], st["ids"], st["z_slice"], st["loss"]] + list(st["terms"].values()):
            t.record_stream(main)
        zb = st["zb"]
        if zb.requires_grad:
            def join(gz):                                        # d loss / d z = the flow's + the branch's
                cur = torch.cuda.current_stream()
                if st.get("event") is None or zb.grad is None:
                    return gz
                cur.wait_event(st["event"])
                zb.grad.record_stream(cur)
                return gz + zb.grad
            out["z"].register_hook(join)
            if late:
                loss = _JoinBranch.apply(out["loss"], None, st)
                ev = main.record_event()                         # the acoustic path's terms are on the caller's stream
                side.wait_event(ev)
                with torch.cuda.stream(side), torch.no_grad():
                    loss.copy_(out["loss"].detach() + st["loss"].detach())
                out["loss"].record_stream(side); loss.record_stream(side)
            else:
                loss = _JoinBranch.apply(out["loss"], st["loss"].detach(), st)
        else:
            loss = out["loss"] + st["loss"]
        out.update(st["terms"])
        out.update({"model_outputs": st["o"], "waveform_seg": st["wav_seg"], "slice_ids": st["ids"], "loss": loss,
                    "z_slice": st["z_slice"]})                    # the decoder's input: its gradient marks the end of the decoder's backward (BucketedSync.attach)
        return out
