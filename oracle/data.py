"""CPU oracle for batch preparation (TEST INFRASTRUCTURE ONLY — never imported by the package).

Plain numpy restatement of what the reference does on the CPU in its DataLoader workers; the HIP kernels of
csrc/data_ops.hip / xva_mel_spectrogram_ragged are checked against it and against tests/golden/data_pipeline.npz, which
oracle/gen_golden_data.py records from the reference's own classes (this oracle is asserted equal to them there).

  item():      TTSDataset.__getitem__ / get_mel / get_text      python/fastpitch1_1/fastpitch/data_function.py:297-352,385-452
  collate():   TTSCollate.__call__ + batch_to_gpu               :565-741
  betabinom(): beta_binomial_prior_distribution                 :84-94
  segment():   MelDataset.__getitem__ (audio path)              python/hifigan/meldataset.py:340-361
  encode_text(): TextProcessing.encode_text for plain text with english_cleaners_v2 reduced to its ASCII / lowercase /
               whitespace steps (common/text/text_processing.py, cleaners.py:91-102, symbols.py:15-21)
"""
import re

import numpy as np
import torch

from . import mel as omel

_ARPABET = ("AA AA0 AA1 AA2 AE AE0 AE1 AE2 AH AH0 AH1 AH2 AO AO0 AO1 AO2 AW AW0 AW1 AW2 AY AY0 AY1 AY2 B CH D DH EH EH0 EH1 EH2 ER ER0 ER1 ER2 "
            "EY EY0 EY1 EY2 F G HH IH IH0 IH1 IH2 IY IY0 IY1 IY2 JH K L M N NG OW OW0 OW1 OW2 OY OY0 OY1 OY2 P R S SH T TH UH UH0 UH1 UH2 UW UW0 "
            "UW1 UW2 V W Y Z ZH").split()
SYMBOLS = list("_-!'(),.:;? ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz") + ["@" + s for s in _ARPABET]
_ID = {s: i for i, s in enumerate(SYMBOLS)}


def encode_text(text):
    text = re.sub(r"\s+", " ", text.lower().replace("/", " "))
    ids = [_ID[c] for c in text if c in _ID and c not in "_~"]
    return [_ID[" "]] + ids + [_ID[" "]]


def betabinom(phoneme_count, mel_count):
    from scipy.stats import betabinom as bb
    x = np.arange(0, phoneme_count)
    return np.array([bb(phoneme_count, i, mel_count + 1 - i).pmf(x) for i in range(1, mel_count + 1)])


def item(wav_i16, text_ids, pitch=None, durs=None):
    """(text, mel (80, T) fp32, pitch, energy, prior source) of one clip, as TTSDataset.__getitem__ builds it."""
    audio = torch.from_numpy(wav_i16.astype(np.float32)) / 32768.0
    mel = omel.mel_m1(audio.unsqueeze(0))[0].numpy()
    energy = np.linalg.norm(mel, ord=2, axis=0)
    return {"text": np.asarray(text_ids, dtype=np.int64), "mel": mel, "pitch": pitch, "energy": energy, "durs": durs}


def collate(items, stage):
    """TTSCollate for `stage` (+ batch_to_gpu's dtype conversions).  Stable descending sort by text length."""
    lens = np.array([len(it["text"]) for it in items])
    order = np.argsort(-lens, kind="stable")
    B, Tt = len(items), int(lens.max())
    Tm = max(it["mel"].shape[1] for it in items)
    out = {"order": order, "text": np.zeros((B, Tt), np.int64), "in_lens": lens[order], "mel": np.zeros((B, 80, Tm), np.float32),
           "mel_lens": np.zeros(B, np.int64)}
    if stage in (3, 4, -1):
        out["pitch"] = np.zeros((B, 1, Tm), np.float32)
        out["energy"] = np.zeros((B, Tm), np.float32)
    if stage not in (1, -1):
        out["durs"] = np.zeros((B, Tt), np.int64)
    else:
        out["attn_prior"] = np.zeros((B, Tm, Tt), np.float32)
    for r, i in enumerate(order):
        it = items[i]
        L, T = len(it["text"]), it["mel"].shape[1]
        out["text"][r, :L] = it["text"]
        out["mel"][r, :, :T] = it["mel"]
        out["mel_lens"][r] = T
        if "pitch" in out:
            # pitch_padded / energy_padded are LongTensors in the reference (dtype=batch[0][0].dtype, :594-606): values truncate
            out["pitch"][r, :, :T] = np.trunc(it["pitch"])
            out["energy"][r, :T] = np.trunc(it["energy"])
        if "durs" in out:
            out["durs"][r, :len(it["durs"])] = torch.from_numpy(np.asarray(it["durs"], dtype=np.float32)).long().numpy()
        else:
            out["attn_prior"][r, :T, :L] = betabinom(L, T)
    out["num_frames"] = int(out["mel_lens"].sum())
    return out


def segment(wav_i16, start, seg=8192):
    """audio / 32768 (float64) -> librosa.util.normalize * 0.95 -> torch.FloatTensor -> crop / right zero pad."""
    audio = wav_i16 / 32768.0
    audio = omel.peak_normalize(audio) * 0.95
    audio = audio.astype(np.float32)
    if len(audio) >= seg:
        return audio[start:start + seg]
    return np.pad(audio, (0, seg - len(audio)))
