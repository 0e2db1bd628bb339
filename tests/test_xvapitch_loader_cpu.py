"""CPU: the xVAPitch fine-tune / priors dataset reader (xva-trainer_amd/xvapitch/xva_train.py:XVAPitchFileLoader, priors_datasets) against what the
reference's read_datasets + TTSDataset do with the same directory (python/xvapitch/dataset.py:223-275, 292-314, 362-381, 596-690): which wav folder is
read, which lines are dropped, where the symbol ids come from — and that ids from another symbol table are an ERROR, not a silent fallback."""
import os

import numpy as np
import pytest


def _dataset(tmp_path, name="voice", **kw):
    from xva_trainer_amd.data import write_synthetic_dataset
    return write_synthetic_dataset(str(tmp_path / name), n_items=6, seed=2, min_s=0.5, max_s=0.9, with_se_embs=True, min_words=3, **kw)


def test_missing_symbol_ids_are_a_hard_error_unless_a_test_opts_in(tmp_path):
    from xva_trainer_amd.xvapitch.xva_train import XVAPitchFileLoader
    ds = _dataset(tmp_path)
    ld = XVAPitchFileLoader(ds, 2, "cpu", allow_basic_text=False)
    with pytest.raises(RuntimeError, match="no symbol ids"):
        ld.item(0)
    # a cached g2p result is used as it is
    os.makedirs(ds + "/tokens")
    np.save(ds + "/tokens/clip_0000.npy", np.array([5, 9, 300, 7]))
    assert ld.item(0)["tokens"].tolist() == [5, 9, 300, 7]
    # tests may fall back to the character table (never the pad id 0), and the fallback is counted
    ld2 = XVAPitchFileLoader(ds, 2, "cpu", allow_basic_text=True)
    tok = ld2.item(1)["tokens"]
    assert tok.min() >= 1 and ld2.missing["tokens"] == 1


def test_fine_tune_set_reads_wavs_postprocessed_and_filters_like_the_reference(tmp_path):
    import shutil
    from xva_trainer_amd.data import read_wav_int16, write_wav_int16
    from xva_trainer_amd.xvapitch.xva_train import XVAPitchFileLoader
    ds = _dataset(tmp_path)
    # the reference trains the fine-tune set on wavs_postprocessed/ (dataset.py:647): raw wavs/ may be 44.1 kHz and are NOT what the embeddings saw
    os.makedirs(ds + "/wavs_postprocessed")
    for f in os.listdir(ds + "/wavs"):
        shutil.copy(ds + "/wavs/" + f, ds + "/wavs_postprocessed/" + f)
        wav, _ = read_wav_int16(ds + "/wavs/" + f)
        write_wav_int16(ds + "/wavs/" + f, wav, sr=44100)
    # one clip shorter than a 32-frame segment, one line shorter than 15 characters
    write_wav_int16(ds + "/wavs_postprocessed/clip_0003.wav", np.zeros(256 * 20, dtype=np.int16))
    lines = open(ds + "/metadata.csv").read().split("\n")
    lines[4] = "clip_0004|too short."
    open(ds + "/metadata.csv", "w").write("\n".join(lines))
    logged = []
    ld = XVAPitchFileLoader(ds, 2, "cpu", allow_basic_text=True, log=logged.append)
    assert all("/wavs_postprocessed/" in it["wav"] for it in ld.items) and len(ld.items) == 4
    assert ld.ignored == {"short_text": 1, "short_clip": 1, "no_embedding": 0} and "Final number of dataset lines: 4" in logged[0]
    assert ld.item(0)["wav"].dtype == np.float32                                       # 22 050 Hz file: loads
    # a set without the postprocessed folder falls back to wavs/; a wrong sample rate is a RuntimeError handleTrainer reports, not a ValueError crash
    shutil.rmtree(ds + "/wavs_postprocessed")
    ld2 = XVAPitchFileLoader(ds, 2, "cpu", allow_basic_text=True)
    with pytest.raises(RuntimeError, match="22050 Hz"):
        ld2.item(0)


def test_priors_tree_is_read_like_read_datasets(tmp_path):
    from xva_trainer_amd.xvapitch.xva_train import LANG_CODES, XVAPitchFileLoader, priors_datasets
    root = tmp_path / "PRIORS"
    _dataset(root, "de_speaker1")
    _dataset(root, "en_speaker2")
    _dataset(root, "xx_unknownlang")            # not a priors language: skipped (dataset.py:616)
    os.makedirs(root / "notadataset")
    sets = priors_datasets(str(root))
    assert [os.path.basename(p) for p, _ in sets] == ["de_speaker1", "en_speaker2"] and [l for _, l in sets] == ["de", "en"]
    ld = XVAPitchFileLoader(sets, 2, "cpu", is_ft=False, allow_basic_text=True)
    assert len(ld.items) == 12 and ld.languages == {"de", "en"} and len(ld) == 6
    ids = {ld.item(i)["lang_id"] for i in range(12)}
    assert ids == {LANG_CODES.index("de"), LANG_CODES.index("en")}
    # priors items without a speaker embedding are dropped, like the reference (dataset.py:655-657)
    os.remove(str(root / "de_speaker1" / "se_embs" / "clip_0000.npy"))
    assert len(XVAPitchFileLoader(sets, 2, "cpu", is_ft=False, allow_basic_text=True).items) == 11
