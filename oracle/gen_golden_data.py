"""Golden vectors for the batch-preparation path and the xVAPitch mel (M3), recorded by RUNNING THE REFERENCE's own classes
(imported from /root/reference with the stubs of oracle/ref_import.py plus `unidecode` / `inflect` placeholders) on synthetic
int16 clips.  Build container only:

    python oracle/gen_golden_data.py

Writes tests/golden/data_pipeline.npz (clips + texts + what TTSCollate / batch_to_gpu / MelDataset.__getitem__ /
beta_binomial_prior_distribution / TextProcessing.encode_text produce for them) and tests/golden/mel_m3.npz (TorchSTFT mel and
linear magnitude + the gradient of an L1 mel loss through it).  Asserts the CPU oracle (oracle/data.py, oracle/mel.py) equals the
reference on the same inputs before writing.  The fixtures are data; no reference source is copied.
"""
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import data as odata  # noqa: E402
from oracle import mel as omel  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
TEXTS = ["Alpha bravo, charlie delta echo foxtrot golf hotel.", "Kilo lima!", "India juliet: it's a test / of symbols; (really)?",
         "Echo echo echo.", "Hotel golf foxtrot echo delta."]
SECONDS = [1.10, 0.30, 0.86, 0.52, 0.71]          # clip 1 is shorter than one HiFi-GAN segment (8192 samples = 0.37 s)


def import_data_modules():
    df = ref_import.import_data_function()
    from python.fastpitch1_1.common.text.text_processing import TextProcessing
    from python.fastpitch1_1.common.layers import TacotronSTFT
    from python.hifigan import meldataset as hm
    from python.xvapitch import audio as xa
    return df, TextProcessing, TacotronSTFT, hm, xa


def write_wav(path, data, sr=22050):
    import wave
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sr)
        f.writeframes(np.asarray(data, dtype="<i2").tobytes())


def main():
    df, TextProcessing, TacotronSTFT, hm, xa = import_data_modules()
    rng = np.random.RandomState(7)
    clips = [np.round(omel.synth_wave(int(s * 22050), 900 + i) * 32768.0).astype(np.int16) for i, s in enumerate(SECONDS)]
    clips[3] = (clips[3].astype(np.int32) // 3).astype(np.int16)                  # a quiet clip: peak normalisation matters
    out = {"n_clips": len(clips), "texts": np.array(TEXTS)}
    for i, c in enumerate(clips):
        out["clip%d" % i] = c

    # ---- text: TextProcessing.encode_text (english_cleaners_v2, p_arpabet 0) + get_text's space symbols ----
    tp = TextProcessing(None, "english_basic", ["english_cleaners_v2"], p_arpabet=0.0)
    space = [tp.encode_text("A A", use_arpabet=False)[1]]
    texts = [space + tp.encode_text(t, use_arpabet=False) + space for t in TEXTS]
    for i, t in enumerate(texts):
        assert odata.encode_text(TEXTS[i]) == t, (TEXTS[i], odata.encode_text(TEXTS[i]), t)
        out["text%d" % i] = np.asarray(t, dtype=np.int64)

    # ---- TTSDataset.__getitem__ (restated call sequence with the reference's own pieces) -> TTSCollate -> batch_to_gpu ----
    stft = TacotronSTFT(1024, 256, 1024, 80, 22050, 0, 8000)
    items3, items1, oitems = [], [], []
    for i, c in enumerate(clips):
        audio = torch.FloatTensor(c.astype(np.float32))                               # load_wav_to_torch (common/utils.py:42-48)
        mel = torch.squeeze(stft.mel_spectrogram((audio / 32768.0).unsqueeze(0)), 0).numpy()     # get_mel :408-416
        T = mel.shape[1]
        pitch = rng.randn(1, T).astype(np.float32)
        pitch[0, rng.rand(T) < 0.3] = 0.0
        energy = np.linalg.norm(mel, ord=2, axis=0)                                    # :327
        L = len(texts[i])
        d = np.ones(L, dtype=np.float32)
        for k in rng.randint(0, L, size=T - L):
            d[k] += 1
        d = d + rng.choice([0.0, 0.25], size=L).astype(np.float32)                    # float .npy durations: the collate truncates them
        text_t = torch.LongTensor(texts[i])
        prior = df.beta_binomial_prior_distribution(L, T)
        items3.append((text_t, mel, L, pitch, energy, None, None, d, "wavs/clip%d.wav" % i))
        items1.append((text_t, mel, L, [0], [0], None, prior, None, "wavs/clip%d.wav" % i))
        out["pitch%d" % i], out["durs%d" % i] = pitch, d
        oi = odata.item(c, texts[i], pitch, d)
        assert np.array_equal(oi["mel"], mel) and np.array_equal(oi["energy"], energy)
        oitems.append(oi)
    for stage, items in ((3, items3), (1, items1)):
        col = df.TTSCollate()
        col.training_stage = stage
        batch = col(items)
        x, y, num_frames = df.batch_to_gpu(batch, training_stage=stage)
        text_padded, input_lengths, mel_padded, output_lengths, pitch_padded, energy_padded, _, attn_prior, durs_padded = x[:9]
        pre = "s%d_" % stage
        out[pre + "text"], out[pre + "in_lens"] = text_padded.numpy(), input_lengths.numpy()
        out[pre + "mel"], out[pre + "mel_lens"] = mel_padded.numpy(), output_lengths.numpy()
        out[pre + "num_frames"] = np.int64(int(num_frames))
        oc = odata.collate(oitems, stage)
        assert np.array_equal(oc["text"], out[pre + "text"]) and np.array_equal(oc["mel"], out[pre + "mel"])
        assert np.array_equal(oc["in_lens"], out[pre + "in_lens"]) and np.array_equal(oc["mel_lens"], out[pre + "mel_lens"])
        if stage == 3:
            out[pre + "pitch"], out[pre + "energy"], out[pre + "durs"] = pitch_padded.numpy(), energy_padded.numpy(), durs_padded.numpy()
            assert np.array_equal(oc["pitch"], out[pre + "pitch"]) and np.array_equal(oc["energy"], out[pre + "energy"])
            assert np.array_equal(oc["durs"].astype(np.float32), out[pre + "durs"])
        else:
            out[pre + "attn_prior"] = attn_prior.numpy()
            assert np.allclose(oc["attn_prior"], out[pre + "attn_prior"], rtol=1e-6, atol=1e-12)

    # ---- MelDataset.__getitem__ on real files ----
    tmp = tempfile.mkdtemp()
    os.makedirs(tmp + "/wavs")
    files = []
    for i, c in enumerate(clips):
        write_wav("%s/wavs/clip%d.wav" % (tmp, i), c)
        files.append("%s/wavs/clip%d.wav" % (tmp, i))
    ds = hm.MelDataset(list(files), 8192, 1024, 80, 256, 1024, 22050, 0, 8000, shuffle=False, fmax_loss=None)
    starts = []
    real_randint = random.randint

    def spy(a, b):
        v = real_randint(a, b)
        starts.append(v)
        return v

    hm.random.randint = spy
    random.seed(99)
    segs, mels, mels_loss, st_all = [], [], [], []
    for i in range(len(files)):
        n0 = len(starts)
        mel, audio, fn, mel_loss, _ = ds[i]
        st = starts[n0] if len(starts) > n0 else 0
        st_all.append(st)
        segs.append(audio.numpy())
        mels.append(mel.numpy())
        mels_loss.append(mel_loss.numpy())
        assert np.array_equal(odata.segment(clips[i], st), audio.numpy()), i
    hm.random.randint = real_randint
    out["hg_starts"], out["hg_audio"] = np.asarray(st_all, dtype=np.int64), np.stack(segs)
    out["hg_mel"], out["hg_mel_loss"] = np.stack(mels), np.stack(mels_loss)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "data_pipeline.npz"), **out)
    print("data_pipeline.npz:", {k: getattr(v, "shape", v) for k, v in out.items() if not k.startswith(("clip", "text", "pitch", "durs"))})

    # ---- M3: xvapitch TorchSTFT (python/xvapitch/audio.py:138-181; VitsGeneratorLoss builds it with use_mel + do_amp_to_db, losses.py:29-46) ----
    import warnings
    warnings.filterwarnings("ignore")
    seg = np.stack([odata.segment(clips[i], st_all[i]) for i in (0, 2, 4)])
    y = torch.from_numpy(seg).clone().requires_grad_(True)
    real_stft = torch.stft

    def stft_compat(*a, **k):                      # torch >= 2 refuses return_complex=False on real input: same values through view_as_real
        k["return_complex"] = True
        return torch.view_as_real(real_stft(*a, **k))

    xa.torch.stft = stft_compat
    try:
        mel_fn = xa.TorchSTFT(1024, 256, 1024, sample_rate=22050, mel_fmin=0.0, mel_fmax=8000.0, n_mels=80, use_mel=True, do_amp_to_db=True)
        lin_fn = xa.TorchSTFT(1024, 256, 1024, sample_rate=22050)
        m3 = mel_fn(y)
        tgt = m3.detach() + 0.3 * torch.from_numpy(np.random.RandomState(3).randn(*m3.shape).astype(np.float32))
        loss = torch.nn.functional.l1_loss(tgt, m3) * 45.0                              # losses.py:187-193
        loss.backward()
        lin = lin_fn(y.detach())
    finally:
        xa.torch.stft = real_stft
    assert torch.equal(omel.mel_m3(y.detach()), m3.detach())
    np.savez_compressed(os.path.join(OUT, "mel_m3.npz"), wav=seg, m3=m3.detach().numpy(), linear=lin.numpy(), tgt=tgt.numpy(),
                        loss=np.float32(loss.item()), d_wav=y.grad.numpy())
    print("mel_m3.npz:", m3.shape, lin.shape, float(loss))


if __name__ == "__main__":
    main()
