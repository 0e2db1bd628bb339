"""Per-shape timing of every xva_gemm launch in one HiFi-GAN D+G iteration (HIP events around each launch)."""
import sys, os, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xva_trainer_amd import _lib, synthetic
from xva_trainer_amd.hifigan.step import HifiganStep
from xva_trainer_amd.mel import mel_spectrogram
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
class A: pass
a = A(); a.compute = "bf16"; a.hg_batch = B; a.hg_steps = 1; a.steps = 1
st = HifiganStep("cuda", "bf16")
g = torch.Generator().manual_seed(0)
for which, flat in ((0, st.flat_g), (1, st.flat_d)):
    flat.copy_(torch.randn(flat.numel(), generator=g) * 0.02)
    for name, off, n, shape, kind in st.eng.table[which]:
        if name.endswith("weight_g"): flat[off:off+n] = 1.0
wav = np.stack([synthetic.synth_wave(8192, 5000 + i) for i in range(B)]); wav = wav / np.abs(wav).max(axis=1, keepdims=True) * 0.95
y = torch.from_numpy(wav.astype(np.float32)).cuda()
x = mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, 8000); ym = mel_spectrogram(y, 1024, 80, 22050, 256, 1024, 0, None)
st.train_step(x, y, ym); torch.cuda.synchronize()
_lib.lib.xva_hg_set_streams(1)     # serial: a launch's event pair times that launch alone
st.train_step(x, y, ym); torch.cuda.synchronize()
_lib.lib.xva_prof_enable(1)
st.train_step(x, y, ym); torch.cuda.synchronize()
_lib.lib.xva_prof_enable(0)
os.makedirs("gpurun_out", exist_ok=True)
_lib.lib.xva_prof_dump(b"gpurun_out/hg_gemm_launches.csv")
rows = list(csv.DictReader(open("gpurun_out/hg_gemm_launches.csv")))
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in rows:
    k = (r["variant"], r["M"], r["N"], r["K"], r["batch"], r["splitk"], r["bn"])
    agg[k][0] += 1; agg[k][1] += float(r["ms"]); agg[k][2] += float(r["gflop"])
tot = sum(v[1] for v in agg.values())
print("total GEMM ms", tot, "launches", len(rows))
names = ["NT", "NN", "TN"]
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%s m%d M=%-6s N=%-5s K=%-7s batch=%-5s sk=%-3s bn=%-3s n=%-3d ms=%8.3f  TF=%7.1f" % (names[int(k[0]) // 3], int(k[0]) % 3, k[1], k[2], k[3], k[4], k[5], k[6], v[0], v[1], v[2] / v[1] if v[1] else 0))

# by section tag (hifigan_engine.hip: phase * 1000 + network * 10; phases 1 generator fwd, 2 D fwd (both), 3 D-step bwd, 4 G-step bwd through D, 5 generator bwd)
sec = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for r in rows:
    t = int(r.get("tag", 0)); k = (t // 1000, (t % 1000) // 10)
    sec[k][0] += 1; sec[k][1] += float(r["ms"]); sec[k][2] += float(r["gflop"]); sec[k][3] += float(r["mbytes"])
print("phase net   launches   ms      TF/s   TB/s(alg)")
for k in sorted(sec):
    v = sec[k]
    print("%5d %3d   %6d %8.3f %8.1f %7.2f" % (k[0], k[1], v[0], v[1], v[2] / v[1] if v[1] else 0, v[3] / v[1] / 1e3 if v[1] else 0))

# the D-step backward (phase 3, the largest phase) taken apart launch by launch, per discriminator (MPD periods 2 / 3 / 5 / 7 / 11 =
# net 0 - 4, MSD scales = net 5 - 7; within a discriminator the launches are in issue order: conv_post's gradients come from direct kernels, then per
# layer {weight gradient TN | backward-data NN (one launch per input phase of a strided convolution)}, last the conv0 weight gradient)
if "--phase3" in sys.argv or os.environ.get("XVA_HG_PHASE3"):
    print("\nphase 3 (discs_backward_d) per launch; lanes off; ms from HIP-event pairs (incl. ~5 us of event overhead each)")
    cur = None
    for r in rows:
        t = int(r.get("tag", 0))
        if t // 1000 != 3:
            continue
        net = (t % 1000) // 10
        if net != cur:
            cur = net
            v = sec[(3, net)]
            print("-- net %d: %d launches, %.3f ms, %.0f TFLOP/s, %.2f TB/s algorithmic" % (net, v[0], v[1], v[2] / v[1], v[3] / v[1] / 1e3))
        ms = float(r["ms"])
        print("   %s M=%-7s N=%-5s K=%-7s batch=%-5s splitk=%-3s kernel=%-7s %7.3f ms %7.1f TFLOP/s %6.2f TB/s" %
              (names[int(r["variant"]) // 3], r["M"], r["N"], r["K"], r["batch"], r["splitk"], r["bn"], ms, float(r["gflop"]) / ms, float(r["mbytes"]) / ms / 1e3))
