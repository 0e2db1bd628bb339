"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (imported from /root/reference with the
stubs in oracle/ref_import.py) on seeded synthetic inputs.  Run in the build container only:

    python oracle/gen_golden.py [mel] [fastpitch] [hifigan]

The fixtures are data (inputs + the reference's outputs); no reference source is copied.
fp32, CPU, dropout off (model.eval()), seeds fixed.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import mel as omel  # noqa: E402
from oracle import ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen_mel(ns):
    torch.manual_seed(1234)
    wav = np.stack([omel.synth_wave(11008, 1234), omel.synth_wave(11008, 1235)])
    wav[1, 9000:] = 0.0
    y = torch.from_numpy(wav)
    m1 = ns.TacotronSTFT().mel_spectrogram(y)
    seg = np.stack([omel.synth_wave(8192, 1236 + i) for i in range(3)])
    seg = np.stack([omel.peak_normalize(s) * 0.95 for s in seg]).astype(np.float32)
    ys = torch.from_numpy(seg)
    hm = ns.hifigan_meldataset
    m2a = hm.mel_spectrogram(ys, 1024, 80, 22050, 256, 1024, 0, 8000)
    m2b = hm.mel_spectrogram(ys, 1024, 80, 22050, 256, 1024, 0, None)
    np.savez_compressed(os.path.join(OUT, "mel.npz"), wav=wav, m1=m1.numpy(), wav_seg=seg, m2_fmax8000=m2a.numpy(),
                        m2_fmaxNone=m2b.numpy(), mel_basis_8000=ns.TacotronSTFT().mel_basis.numpy())
    print("mel.npz", m1.shape, m2a.shape)


def main():
    os.makedirs(OUT, exist_ok=True)
    which = sys.argv[1:] or ["mel", "fastpitch", "hifigan"]
    ns = ref_import.import_reference()
    if "mel" in which:
        gen_mel(ns)
    if "fastpitch" in which:
        from oracle import gen_golden_fastpitch
        gen_golden_fastpitch.generate(ns, OUT)
    if "fastpitch_infer" in which or "fastpitch" in which:
        from oracle import gen_golden_fastpitch
        gen_golden_fastpitch.generate_infer(ns, OUT)
    if "fastpitch_stage1" in which or "fastpitch" in which:
        from oracle import gen_golden_fastpitch
        gen_golden_fastpitch.generate_stage1(ns, OUT)
    if "hifigan" in which:
        from oracle import gen_golden_hifigan
        gen_golden_hifigan.generate(ns, OUT)


if __name__ == "__main__":
    main()
