"""One xVAPitch iteration (BASELINE config C5: generator pass fwd + bwd, discriminator pass fwd + bwd, the two AdamW updates) at the reference's model size —
python/xvapitch/model.py:55-149: latent 192, language embedding 4, speaker vector 512, text encoder 10 layers x 196 channels (ffn 768, 2 heads),
posterior encoder 16 WN layers on 513 spectrogram bins, flow 4 x 4 WN layers, pitch predictor 3 layers x 708 channels, decoder 512 -> 32 channels,
spec_segment_size 32 (8192 samples) — on a synthetic batch (random weights; sizes from argv).

    python tools/c5_step_time.py [B=16] [T_text=100] [T_spec=400] [decoder/discriminator dtype: bf16|fp32] [WaveNet stacks: fp32|bf16]
Prints ms per pass and waveform-segment samples / s; under `rocprofv3 --kernel-trace --stats` gives the per-kernel table of profiles/."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
from xva_trainer_amd.xvapitch.decoder import VitsDecoder
from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
from xva_trainer_amd.xvapitch.train_step import XVAPitchStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
Tt = int(sys.argv[2]) if len(sys.argv) > 2 else 100
Ty = int(sys.argv[3]) if len(sys.argv) > 3 else 400
dt = sys.argv[4] if len(sys.argv) > 4 else "bf16"
adt = sys.argv[5] if len(sys.argv) > 5 else "fp32"        # storage / MFMA dtype of the WaveNet stacks (posterior encoder, flow); transformer / SDP are fp32
VOCAB, LANGS, SEG = 256, 31, 32
torch.manual_seed(0)
ac = AcousticTrainPath(VOCAB, LANGS, pitch=True, compute=adt, dropout_p=0.1, sdp_dropout_p=0.5)      # the reference's defaults, train mode (model.py:55-135; dropout :88,128,166)
dec = VitsDecoder(192, 512, compute=dt)
D = VitsDiscriminator(compute=dt)
gen = torch.Generator().manual_seed(1)
for eng, std in ((dec, 0.02), (D, 0.02)):                              # random weights: weight_v ~ N, weight_g = row norms (unit-gain weight norm)
    sd = {}
    for k, (off, numel, shape) in eng.table.items():
        sd[k] = torch.randn(shape, generator=gen) * std
    for k in list(sd):
        if k.endswith("weight_g"):
            v = sd[k[:-1] + "v"]
            sd[k] = v.reshape(v.size(0), -1).norm(dim=1).reshape(sd[k].shape)
    eng.load_state_dict(sd)
step = XVAPitchStep(GeneratorPass(ac, dec, SEG), D)
dev = "cuda"
x_lens = torch.randint(Tt // 2, Tt + 1, (B,), generator=gen); x_lens[0] = Tt
y_lens = torch.randint(max(Ty // 2, SEG + 1), Ty + 1, (B,), generator=gen); y_lens[0] = Ty
tokens = (torch.randint(1, VOCAB, (B, Tt), generator=gen) * (torch.arange(Tt)[None, :] < x_lens[:, None])).to(dev)
y = (torch.rand(B, 513, Ty, generator=gen) * (torch.arange(Ty)[None, None, :] < y_lens[:, None, None])).to(dev)
wav = (torch.rand(B, 1, Ty * 256, generator=gen) * 1.6 - 0.8).to(dev)
dvec = torch.randn(B, 512, generator=gen).to(dev)
lids = torch.randint(0, LANGS, (B,), generator=gen).to(dev)
pitch = ((torch.rand(B, 1, Ty, generator=gen) * 3 - 1.2).clamp_min(0) * (torch.arange(Ty)[None, None, :] < y_lens[:, None, None])).to(dev)
x_lens, y_lens = x_lens.to(dev), y_lens.to(dev)


EAGER = os.environ.get("XVA_C5_EAGER_DISC", "1") != "0"      # the trainer's order: the discriminator pass inside the generator pass, on the vocoder branch's stream


def iteration():
    step.gen.zero_grad(); D.zero_grad()
    t0 = time.perf_counter()
    o = step.generator_pass(tokens, x_lens, y, y_lens, wav, dvec, lids, pitch_padded=pitch, eager_disc=EAGER)
    o["loss"].backward()
    t1a = time.perf_counter()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    global host_wait
    host_wait = (t1 - t1a) * 1e3                                  # how long the host waits for the GPU after issuing the pass: ~0 = issue-bound
    ld = step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    step.optimizer_step(lr=1e-6, lr_disc=1e-6)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, float(o["loss"]), float(ld), (t3 - t2) * 1e3


for _ in range(2):
    iteration()
N = 5
acc = [0.0, 0.0, 0.0]
for _ in range(N):
    a, b, lg, ld, c_ = iteration()
    acc[0] += a / N; acc[1] += b / N; acc[2] += c_ / N
tot = acc[0] + acc[1] + acc[2]
print("  (generator pass: the host waited %.1f ms for the GPU after issuing the last launch)" % host_wait)
print("xVAPitch C5 iteration, B=%d x (%d symbols, %d spectrogram frames), segment %d samples, decoder / discriminator %s, WaveNet stacks %s, transformer / SDP fp32:" % (B, Tt, Ty, SEG * 256, dt, adt))
print("  generator pass fwd + bwd %.1f ms | discriminator pass fwd + bwd %.1f ms | 2 x AdamW %.1f ms | iteration %.1f ms = %.0f k segment-samples / s, %.0f spectrogram frames / s (losses %.3f / %.3f)"
      % (acc[0], acc[1], acc[2], tot, B * SEG * 256 / tot, float(y_lens.sum()) / tot * 1e3, lg, ld))

if os.environ.get("XVA_C5_GLUE_SITES", "0") != "0":
    # which Python lines of the package issue torch-native device work (memcpys, fills, elementwise glue): one iteration under torch.profiler with Python
    # stacks, exported as a chrome trace; tools/c5_glue_sites.py attributes every runtime launch (hipMemcpy*, hipLaunchKernel of an at::native kernel,
    # hipMemset*) to the innermost package frame that encloses it on its thread (the autograd thread's frames included)
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        iteration(); torch.cuda.synchronize()
    os.makedirs("gpurun_out", exist_ok=True)
    prof.export_chrome_trace("gpurun_out/c5_trace.json.gz")
    import subprocess
    subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "c5_glue_sites.py"), "gpurun_out/c5_trace.json.gz"])
    sys.exit(0)

if os.environ.get("XVA_C5_GEMM_PROFILE", "0") != "0":
    # every xva_gemm launch of one iteration with its own HIP event pair (csrc/core.hip xva_prof_*), by shape
    import collections, csv
    from xva_trainer_amd import _lib
    _lib.lib.xva_prof_enable(1)
    iteration(); torch.cuda.synchronize()
    _lib.lib.xva_prof_enable(0)
    os.makedirs("gpurun_out", exist_ok=True)
    _lib.lib.xva_prof_dump(b"gpurun_out/c5_gemm_launches.csv")
    rows = list(csv.DictReader(open("gpurun_out/c5_gemm_launches.csv")))
    agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for r in rows:
        k = (r["variant"], r["M"], r["N"], r["K"], r["batch"], r["splitk"], r["bn"])
        agg[k][0] += 1; agg[k][1] += float(r["ms"]); agg[k][2] += float(r["gflop"])
    print("  xva_gemm: %d launches, %.2f ms" % (len(rows), sum(v[1] for v in agg.values())))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print("  %s m%d M=%-6s N=%-5s K=%-6s batch=%-4s sk=%-3s bn=%-6s n=%-4d ms=%7.3f us/launch=%6.1f TF=%6.1f"
              % (["NT", "NN", "TN"][int(k[0]) // 3], int(k[0]) % 3, k[1], k[2], k[3], k[4], k[5], k[6], v[0], v[1], v[1] / v[0] * 1e3, v[2] / v[1] if v[1] else 0))

if os.environ.get("XVA_C5_CPU_BASELINE", "0") != "0":
    # The CPU restatement (oracle/: the checker of tests/, timed here as the reference-algorithm baseline on this host's cores) on the first
    # Bc items of the same batch, same weights: generator pass + discriminator pass, forward + backward, fp32 torch-CPU.
    import torch.nn.functional as F
    from oracle import hifigan as ohg, mel as omel, xvapitch as oxv
    Bc, threads = min(B, 4), 8
    torch.set_num_threads(threads)
    cfg = {"latent": 192, "lang_dim": 4, "dvec": 512, "heads": 2, "te_layers": 10, "pe_layers": 16, "flow_layers": 4, "num_flows": 4}
    cpu = lambda t: t[:Bc].detach().cpu()
    leaves = {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point()) for k, v in ac.state_dict().items()}
    dl = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in dec.state_dict().items()}
    ddl = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in D.state_dict().items()}
    tk, xl, yy, yl, ww, dv, li, pp = (cpu(t) for t in (tokens, x_lens, y, y_lens, wav, dvec, lids, pitch))
    Tyc, Ttc = int(yl.max()), int(xl.max())
    tk, yy, ww, pp = tk[:, :Ttc], yy[:, :, :Tyc], ww[:, :, :Tyc * 256], pp[:, :, :Tyc]
    eps, noise = torch.randn(Bc, 192, Tyc), torch.randn(Bc, 2, Ttc)
    ids = (torch.rand(Bc) * (yl - SEG + 1)).long()

    def cpu_iteration():
        t0 = time.perf_counter()
        o = oxv.acoustic_losses(leaves, tk, xl, yy, yl, dv, li, eps, noise, cfg, pitch_padded=pp)
        g = F.normalize(dv).unsqueeze(-1)
        wav_hat = ohg.vits_decoder(dl, oxv.segment(o["z"], ids, SEG), g)
        seg = oxv.segment(ww, ids * 256, SEG * 256)
        loss_mel = F.l1_loss(omel.mel_m3(seg.squeeze(1)), omel.mel_m3(wav_hat.squeeze(1)), reduction="none").mean() * 45
        rs, fr, gs, fg = ohg.vits_disc({k: v.detach() for k, v in ddl.items()}, seg, wav_hat)
        loss = o["loss"] + loss_mel + ohg.generator_loss(gs) + ohg.feature_loss(fr, [[t.detach() for t in f] for f in fg])
        loss.backward()
        rs, fr, gs, fg = ohg.vits_disc(ddl, seg, wav_hat.detach())
        ohg.discriminator_loss(rs, gs).backward()
        return time.perf_counter() - t0
    cpu_iteration()
    s = min(cpu_iteration() for _ in range(2))
    print("  CPU baseline (oracle port, %d threads of %d host CPUs, first %d items, 1 warm-up + best of 2): %.2f s / iteration = %.1f k segment-samples / s"
          % (threads, os.cpu_count(), Bc, s, Bc * SEG * 256 / s / 1e3))
