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


class GeneratorPass:
    def __init__(self, acoustic, decoder, spec_segment_size=32, mel_loss_alpha=45.0):
        self.acoustic, self.decoder = acoustic, decoder
        self.S, self.alpha = int(spec_segment_size), float(mel_loss_alpha)                 # model.py:77, losses.py:27
        self.stft = xmel.TorchSTFTMel(1024, 256, 1024, sample_rate=22050, mel_fmin=0.0, mel_fmax=8000.0, n_mels=80)    # losses.py:29-46

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

    def __call__(self, tokens, x_lengths, y, y_lengths, waveform, d_vectors, language_ids, pitch_padded=None, eps=None, noise=None, slice_ids=None):
        """waveform (B, 1, Ty * 256).  slice_ids (B,): the segment starts (drawn like the reference's rand_segments when None)."""
        out = self.acoustic(tokens, x_lengths, y, y_lengths, d_vectors, language_ids, eps=eps, noise=noise, pitch_padded=pitch_padded)
        g = F.normalize(d_vectors.float()).unsqueeze(-1)
        S = self.S
        if slice_ids is None:
            z_slice, slice_ids = ops.rand_segments(out["z"], y_lengths.to(out["z"].device), S)             # :850
        else:
            z_slice = ops.segment(out["z"], slice_ids, S)
        o = self.decoder(z_slice, g)                                                                      # :852
        wav_seg = ops.segment(waveform.float(), slice_ids * 256, S * 256)                                 # :856-860
        with torch.no_grad():
            mel_tgt = self.stft(wav_seg)
        loss_mel = _MelL1.apply(o, mel_tgt, self.stft, self.alpha)                                        # losses.py:187-193
        out.update({"model_outputs": o, "waveform_seg": wav_seg, "slice_ids": slice_ids, "loss_mel": loss_mel, "loss": out["loss"] + loss_mel,
                    "z_slice": z_slice})                          # the decoder's input: its gradient marks the end of the decoder's backward (BucketedSync.attach)
        return out
