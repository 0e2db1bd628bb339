"""GPU: the data-parallel step (fastpitch/dp.py: per-bucket HIP events, side-stream RCCL all-reduce) on a 1-rank RCCL group
reproduces the plain step (up to the summation order of fp32 atomics); world-size-2 semantics are covered on CPU/gloo in tests/test_dp_cpu.py."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("stage", [3, 2])
def test_gradsync_single_rank_rccl_matches_plain_step(stage):
    import torch.distributed as dist
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import engine as E, params as P
    from xva_trainer_amd.fastpitch.dp import GradSync
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sd = ofp.init_state_dict(21)
        batch = E.DeviceBatch.from_dict(ofp.synth_batch(3, 17, 70, 22), "cuda")
        eng = E.FastPitchEngine("cuda", "bf16", p_dropout=0.1, seed=5)
        flat = torch.zeros(eng.total, device="cuda")
        P.to_flat(sd, eng.table, flat)
        g_plain = torch.zeros_like(flat)
        l_plain = eng.fwd_loss_bwd(flat, g_plain, batch, stage).clone()
        eng.step = 0                                                  # same dropout masks for the second pass
        g_dp = torch.zeros_like(flat)
        sync = GradSync(eng, flat, g_dp, 1)
        l_dp = sync.fwd_loss_bwd(batch, stage)
        torch.cuda.synchronize()
        assert torch.allclose(l_dp, l_plain, rtol=1e-5, atol=1e-7)
        assert ((g_dp - g_plain).norm() / g_plain.norm()).item() < 1e-5          # fp32 atomics (bias / LayerNorm gradients) reorder between runs
        # gradient accumulation: only the last micro-batch synchronises
        eng.step = 0
        g2 = torch.zeros_like(flat)
        sync2 = GradSync(eng, flat, g2, 1)
        sync2.fwd_loss_bwd(batch, stage, grad_scale=0.5, sync=False)
        eng.step = 0
        sync2.fwd_loss_bwd(batch, stage, grad_scale=0.5, sync=True)
        torch.cuda.synchronize()
        rel = ((g2 - g_plain).norm() / g_plain.norm()).item()
        assert rel < 2e-2, rel                                        # two half-scaled bf16 passes vs one full pass
    finally:
        dist.destroy_process_group()


def test_xvapitch_c5_sync_gradients_single_rank_rccl(golden_dir):
    """xvapitch/train_step.py:XVAPitchStep.sync_gradients on a 1-rank RCCL group: the flat gather / all-reduce / scatter over the acoustic modules'
    own gradient tensors (incl. the non-contiguous view of the padded posterior-encoder weight), the decoder's and the discriminator's flat
    buffers returns every gradient unchanged (mean over one rank); world-size-2 arithmetic is covered on gloo in tests/test_dp_cpu.py."""
    import numpy as np
    import torch.distributed as dist
    from oracle import hifigan as ohg
    from xva_trainer_amd.xvapitch.acoustic import AcousticTrainPath
    from xva_trainer_amd.xvapitch.decoder import VitsDecoder
    from xva_trainer_amd.xvapitch.discriminator import VitsDiscriminator
    from xva_trainer_amd.xvapitch.generator_pass import GeneratorPass
    from xva_trainer_amd.xvapitch.train_step import XVAPitchStep
    g = np.load(os.path.join(golden_dir, "xvapitch_genpass.npz"))
    c = {str(k): int(v) for k, v in zip(g["cfg_keys"], g["cfg_vals"])}
    ac = AcousticTrainPath(c["vocab"], c["langs"], latent_size=c["latent"], embedded_language_dim=c["lang_dim"], d_vector_dim=c["dvec"],
                           hidden_channels_ffn=c["ffn"], num_heads=c["heads"], text_layers=c["te_layers"], posterior_layers=c["pe_layers"],
                           flow_layers=c["flow_layers"], num_flows=c["num_flows"], spec_bins=c["spec_bins"], pitch=True)
    ac.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd/")})
    dec = VitsDecoder(c["latent"], c["dvec"]); dec.load_state_dict(ohg.init_vits_decoder_sd(int(g["dec_seed"]), c["latent"], c["dvec"]))
    D = VitsDiscriminator(); D.load_state_dict(ohg.init_vits_disc_sd(3))
    step = XVAPitchStep(GeneratorPass(ac, dec, spec_segment_size=int(g["seg"])), D)
    t = lambda k: torch.from_numpy(g[k]).cuda()
    o = step.generator_pass(t("tokens"), t("x_lens"), t("y"), t("y_lens"), t("wav"), t("dvec"), t("lids"), pitch_padded=t("pitch"), eps=t("eps"),
                            noise=t("noise"), slice_ids=t("slice_ids"))
    o["loss"].backward()
    step.discriminator_pass(o["model_outputs"].detach(), o["waveform_seg"])
    before = {n + k: v.detach().clone() for n, m in (("ac/", ac), ("dec/", dec), ("D/", D)) for k, v in m.grads().items()}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        step.sync_gradients()
        torch.cuda.synchronize()
        after = {n + k: v.clone() for n, m in (("ac/", ac), ("dec/", dec), ("D/", D)) for k, v in m.grads().items()}
        # the trainer's form: buckets on a side stream (decoder, then the acoustic modules group by group; the discriminator after its pass),
        # the compute stream waiting for a group's buckets right before that group's optimiser step
        from xva_trainer_amd.xvapitch.train_step import BucketedSync
        bs = BucketedSync(step)
        bs.start_generator(); bs.start_discriminator()
        assert len(bs.pending["gen"]) >= 5 and len(bs.pending["disc"]) == 1
        bs.finish("gen"); bs.finish("disc")
        torch.cuda.synchronize()
        after2 = {n + k: v for n, m in (("ac/", ac), ("dec/", dec), ("D/", D)) for k, v in m.grads().items()}
    finally:
        dist.destroy_process_group()
    assert len(before) > 800 and all(torch.equal(before[k], after[k]) for k in before)
    assert all(torch.equal(before[k], after2[k]) for k in before)
