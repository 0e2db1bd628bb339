"""GPU: the on-device batch preparation (csrc/data_ops.hip, xva_mel_spectrogram_ragged; host: xva-trainer_amd/data.py) against the
vectors recorded from the reference's TTSCollate / batch_to_gpu / MelDataset.__getitem__ / beta_binomial_prior_distribution
(tests/golden/data_pipeline.npz) and against the CPU oracle (oracle/data.py) on other sizes.  Index / integer work is bit-exact;
floating point at 1e-3 (log-mel: absolute, as in tests/test_mel_gpu.py)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ATOL = 1e-3


def _items(g, n):
    return [{"wav": g["clip%d" % i], "text": g["text%d" % i], "pitch": g["pitch%d" % i], "durs": g["durs%d" % i]} for i in range(n)]


def _check_energy(out, ref_trunc, ref_full):
    """truncated energies agree except where the untruncated value sits within 1e-3 of an integer (fp32 summation order)."""
    diff = (out - ref_trunc) != 0
    frac = np.abs(ref_full - np.round(ref_full))
    assert np.all(frac[diff] < 1e-3) and np.all(np.abs(out - ref_trunc)[diff] <= 1)


@pytest.mark.parametrize("stage", [3, 1, 2])
def test_device_collate_matches_reference_golden(golden_dir, stage):
    from oracle import data as odata
    from xva_trainer_amd.data import DeviceCollate
    g = np.load(os.path.join(golden_dir, "data_pipeline.npz"))
    n = int(g["n_clips"])
    b = DeviceCollate("cuda:0")(_items(g, n), stage)
    torch.cuda.synchronize()
    pre = "s3_" if stage in (2, 3) else "s1_"
    assert np.array_equal(b.text.cpu().numpy(), g[pre + "text"])                      # ids, lengths, order: bit-exact
    assert np.array_equal(b.in_lens.cpu().numpy(), g[pre + "in_lens"])
    assert np.array_equal(b.mel_lens.cpu().numpy(), g[pre + "mel_lens"])
    assert int(b.mel_lens.sum()) == int(g[pre + "num_frames"])
    mel, ref = b.mel_tgt.cpu().numpy(), g[pre + "mel"]
    assert mel.shape == ref.shape and np.abs(mel - ref).max() < ATOL
    assert np.array_equal(mel == 0, ref == 0)                                          # the zero padding is exact zeros (loss masks on mel_tgt != 0)
    if stage == 3:
        assert np.array_equal(b.pitch.cpu().numpy().reshape(ref.shape[0], 1, -1), g["s3_pitch"])
        full = odata.collate([odata.item(g["clip%d" % i], g["text%d" % i], g["pitch%d" % i], g["durs%d" % i]) for i in range(n)], 3)
        order = full["order"]
        e_full = np.zeros_like(g["s3_energy"])
        for r, i in enumerate(order):
            e = np.linalg.norm(odata.item(g["clip%d" % i], g["text%d" % i])["mel"], ord=2, axis=0)
            e_full[r, :len(e)] = e
        _check_energy(b.energy.cpu().numpy(), g["s3_energy"], e_full)
    if stage in (2, 3):
        assert np.array_equal(b.durs.cpu().numpy(), g["s3_durs"].astype(np.int32))
        assert b.attn_prior is None
    else:
        assert b.durs is None and b.pitch is None
        assert np.allclose(b.attn_prior.cpu().numpy(), g["s1_attn_prior"], rtol=1e-4, atol=1e-9)


def test_collate_without_reference_truncation_keeps_fractions(golden_dir):
    from xva_trainer_amd.data import DeviceCollate
    g = np.load(os.path.join(golden_dir, "data_pipeline.npz"))
    n = int(g["n_clips"])
    b = DeviceCollate("cuda:0", reference_int_truncation=False)(_items(g, n), 3)
    p = b.pitch.cpu().numpy()
    assert not np.array_equal(p, np.trunc(p)) and np.array_equal(np.trunc(p).reshape(n, 1, -1), g["s3_pitch"])
    e = b.energy.cpu().numpy()
    assert np.abs(np.trunc(e) - g["s3_energy"]).max() <= 1


@pytest.mark.parametrize("B,seed", [(1, 0), (7, 1), (32, 2)])
def test_device_collate_matches_oracle_ragged(B, seed):
    """Other batch sizes / lengths, ties in the text length (stable order), clips of 1..3 frames over a frame boundary."""
    from oracle import data as odata, mel as omel
    from xva_trainer_amd.data import DeviceCollate
    rng = np.random.RandomState(seed)
    items, oitems = [], []
    for i in range(B):
        n = int(rng.choice([2048, 2303, 2304, 5000, 12000, 22050]))
        wav = np.round(omel.synth_wave(n, 50 + i) * 32768.0).astype(np.int16)
        L = int(rng.randint(3, 9)) if i % 3 else 5                                     # ties
        text = rng.randint(1, 148, size=L)
        T = 1 + n // 256
        pitch = rng.randn(1, T).astype(np.float32) * 3
        durs = rng.randint(0, 4, size=L).astype(np.float32)
        items.append({"wav": wav, "text": text, "pitch": pitch, "durs": durs})
        oitems.append(odata.item(wav, text, pitch, durs))
    for stage in (3, 1):
        ref = odata.collate(oitems, stage)
        b = DeviceCollate("cuda:0")(items, stage)
        assert np.array_equal(b.order.cpu().numpy(), ref["order"])
        assert np.array_equal(b.text.cpu().numpy(), ref["text"]) and np.array_equal(b.in_lens.cpu().numpy(), ref["in_lens"])
        assert np.array_equal(b.mel_lens.cpu().numpy(), ref["mel_lens"])
        assert np.abs(b.mel_tgt.cpu().numpy() - ref["mel"]).max() < ATOL
        if stage == 3:
            assert np.array_equal(b.pitch.cpu().numpy().reshape(B, 1, -1), ref["pitch"])
            assert np.array_equal(b.durs.cpu().numpy(), ref["durs"].astype(np.int32))
        else:
            assert np.allclose(b.attn_prior.cpu().numpy(), ref["attn_prior"], rtol=1e-4, atol=1e-9)


def test_hifigan_segments_match_reference_meldataset(golden_dir):
    """int16 / 32768 -> peak normalise * 0.95 -> crop / zero pad: bit-exact (fp64 arithmetic rounded to fp32 like numpy + FloatTensor);
    the two mels the reference dataset computes per item come from the HIP mel at 1e-3."""
    from xva_trainer_amd.data import prepare_segments
    from xva_trainer_amd.mel import mel_spectrogram
    g = np.load(os.path.join(golden_dir, "data_pipeline.npz"))
    n = int(g["n_clips"])
    y = prepare_segments([g["clip%d" % i] for i in range(n)], g["hg_starts"].tolist(), 8192, "cuda:0")
    assert np.array_equal(y.cpu().numpy(), g["hg_audio"])
    assert np.abs(mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, 8000).cpu().numpy() - g["hg_mel"]).max() < ATOL
    assert np.abs(mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, None).cpu().numpy() - g["hg_mel_loss"]).max() < ATOL


def test_hifigan_segments_match_oracle_edge_cases():
    from oracle import data as odata
    from xva_trainer_amd.data import prepare_segments
    rng = np.random.RandomState(5)
    clips = [rng.randint(-32768, 32768, size=n).astype(np.int16) for n in (8192, 8193, 100, 30000)]
    clips.append(np.zeros(9000, dtype=np.int16))                                       # silence: librosa.normalize leaves it alone
    clips.append(np.full(9000, -32768, dtype=np.int16))                                # |int16 min| = 32768
    starts = [0, 1, 0, 30000 - 8192, 17, 808]
    y = prepare_segments(clips, starts, 8192, "cuda:0").cpu().numpy()
    for i, c in enumerate(clips):
        assert np.array_equal(y[i], odata.segment(c, starts[i])), i


def test_file_loaders_and_trainer_dataset(tmp_path):
    """FastPitchFileLoader / HifiFileLoader over a reference-layout directory: every batch equals the oracle's collate of the same files."""
    from oracle import data as odata
    from xva_trainer_amd import data as D
    path = D.write_synthetic_dataset(str(tmp_path / "voice"), n_items=6, seed=1, min_s=0.4, max_s=0.9)
    ld = D.FastPitchFileLoader(path, 3, 1, "cuda:0", shuffle=False)
    assert len(ld) == 2 and ld.actual_num_lines == 6
    meta = D.read_metadata(path)
    for bi, b in enumerate(ld):
        idx = range(bi * 3, bi * 3 + 3)
        oitems = [odata.item(D.read_wav_int16(meta[i][1])[0], odata.encode_text(meta[i][2])) for i in idx]
        ref = odata.collate(oitems, 1)
        assert np.array_equal(b.text.cpu().numpy(), ref["text"]) and np.array_equal(b.mel_lens.cpu().numpy(), ref["mel_lens"])
        assert np.abs(b.mel_tgt.cpu().numpy() - ref["mel"]).max() < ATOL
    with pytest.raises(FileNotFoundError):                                             # stage 3 needs the duration files stage 1 leaves behind
        next(iter(D.FastPitchFileLoader(path, 3, 3, "cuda:0")))
    hl = D.HifiFileLoader(path, 4, "cuda:0", dm=2)
    assert len(hl) == 3
    for y in hl:
        assert y.shape == (4, 8192) and y.is_cuda and float(y.abs().max()) <= 0.95 + 1e-6
    # data-parallel sharding: two ranks see disjoint halves of the same epoch order
    a = D.FastPitchFileLoader(path, 3, 1, "cuda:0", rank=0, world=2, seed=9)
    b = D.FastPitchFileLoader(path, 3, 1, "cuda:0", rank=1, world=2, seed=9)
    fa, fb = next(iter(a)), next(iter(b))
    assert len(a) == len(b) == 1 and int(fa.mel_lens.sum()) + int(fb.mel_lens.sum()) == sum(1 + len(D.read_wav_int16(m[1])[0]) // 256 for m in meta)
