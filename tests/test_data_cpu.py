"""CPU: the batch-preparation oracle (oracle/data.py) and the host-side file / text helpers of the package against the vectors
recorded from the reference's own TTSCollate / batch_to_gpu / MelDataset / TextProcessing / beta-binomial prior
(tests/golden/data_pipeline.npz, written by oracle/gen_golden_data.py)."""
import os

import numpy as np
import pytest
import torch


def _load(golden_dir):
    g = np.load(os.path.join(golden_dir, "data_pipeline.npz"))
    n = int(g["n_clips"])
    return g, n


def _oracle_items(g, n):
    from oracle import data as odata
    return [odata.item(g["clip%d" % i], g["text%d" % i], g["pitch%d" % i], g["durs%d" % i]) for i in range(n)]


def test_text_encoders_match_reference_ids(golden_dir):
    """oracle.encode_text and the package's BasicTextEncoder reproduce TextProcessing.encode_text (english_cleaners_v2, text mode)
    + the prepended / appended space symbol for plain text (data_function.py:431-452)."""
    from oracle import data as odata
    from xva_trainer_amd.data import BasicTextEncoder
    g, n = _load(golden_dir)
    enc = BasicTextEncoder()
    assert enc.symbols == odata.SYMBOLS and len(enc.symbols) == 148
    for i in range(n):
        ref = g["text%d" % i].tolist()
        assert odata.encode_text(str(g["texts"][i])) == ref
        assert enc.encode(str(g["texts"][i])) == ref
    assert enc.encode("a {HH AH0} b") == [enc.space, enc.to_id["a"], enc.space, enc.to_id["@HH"], enc.to_id["@AH0"], enc.space, enc.to_id["b"], enc.space]


@pytest.mark.parametrize("stage", [3, 1])
def test_collate_oracle_matches_reference(golden_dir, stage):
    from oracle import data as odata
    g, n = _load(golden_dir)
    c = odata.collate(_oracle_items(g, n), stage)
    pre = "s%d_" % stage
    for k in ("text", "in_lens", "mel", "mel_lens"):
        assert np.array_equal(c[k], g[pre + k]), k
    assert c["num_frames"] == int(g[pre + "num_frames"])
    assert list(c["in_lens"]) == sorted(c["in_lens"], reverse=True)
    if stage == 3:
        assert np.array_equal(c["pitch"], g[pre + "pitch"]) and np.array_equal(c["energy"], g[pre + "energy"])
        assert np.array_equal(c["durs"].astype(np.float32), g[pre + "durs"])
        assert np.array_equal(c["pitch"], np.trunc(c["pitch"]))          # the reference's LongTensor truncation (data_function.py:594-606)
    else:
        assert np.allclose(c["attn_prior"], g[pre + "attn_prior"], rtol=1e-6, atol=1e-12)
        for r in range(n):                                               # each live row of the prior is a distribution over P + 1 values minus the last
            L, T = int(c["in_lens"][r]), int(c["mel_lens"][r])
            assert np.all(c["attn_prior"][r, :T, :L].sum(1) <= 1 + 1e-6) and c["attn_prior"][r, T:].sum() == 0 and c["attn_prior"][r, :, L:].sum() == 0


def test_segment_oracle_matches_reference_meldataset(golden_dir):
    from oracle import data as odata, mel as omel
    g, n = _load(golden_dir)
    for i in range(n):
        seg = odata.segment(g["clip%d" % i], int(g["hg_starts"][i]))
        assert np.array_equal(seg, g["hg_audio"][i]), i
    y = torch.from_numpy(g["hg_audio"])
    for i in range(n):                                                                  # the dataset computes its mels one item at a time
        assert torch.equal(omel.mel_m2(y[i:i + 1], fmax=8000)[0], torch.from_numpy(g["hg_mel"][i]))
        assert torch.equal(omel.mel_m2(y[i:i + 1], fmax=None)[0], torch.from_numpy(g["hg_mel_loss"][i]))
    assert len(g["clip1"]) < 8192 and np.all(g["hg_audio"][1][len(g["clip1"]):] == 0)       # the right-zero-pad case is in the fixture
    assert abs(np.abs(g["hg_audio"][3]).max() - 0.95) < 1e-3 or int(g["hg_starts"][3]) > 0     # the quiet clip is peak-normalised


def test_m3_oracle_matches_reference_golden(golden_dir):
    from oracle import mel as omel
    g = np.load(os.path.join(golden_dir, "mel_m3.npz"))
    y = torch.from_numpy(g["wav"]).clone().requires_grad_(True)
    m3 = omel.mel_m3(y)
    assert torch.equal(m3.detach(), torch.from_numpy(g["m3"]))
    loss = torch.nn.functional.l1_loss(torch.from_numpy(g["tgt"]), m3) * 45.0
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) < 1e-5
    assert torch.allclose(y.grad, torch.from_numpy(g["d_wav"]), rtol=1e-4, atol=1e-7)


def test_dataset_directory_roundtrip(tmp_path):
    """write_synthetic_dataset -> read_metadata / read_wav_int16 (the reference layout: metadata.csv, wavs/, pitch/)."""
    from xva_trainer_amd import data as D
    path = D.write_synthetic_dataset(str(tmp_path / "voice"), n_items=4, seed=3, min_s=0.3, max_s=0.6)
    items = D.read_metadata(path)
    assert len(items) == 4 and all(os.path.exists(p) for _, p, _ in items)
    wav, sr = D.read_wav_int16(items[0][1])
    assert sr == 22050 and wav.dtype == np.int16 and 0.3 * 22050 <= len(wav) <= 0.6 * 22050
    p = np.load(os.path.join(path, "pitch", items[0][0] + ".npy"))
    assert p.shape == (1, 1 + len(wav) // 256)
