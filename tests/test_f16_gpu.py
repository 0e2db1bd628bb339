"""The fp16-operand mode (round 6): IEEE-half MFMA operands (v_mfma_f32_16x16x32_f16) over an fp32 residual stream.

(a) xva_gemm with XVA_F16 operands on every direct-to-LDS tile against fp64 references built from the same fp16-rounded operands: the only difference
    with an fp32 C is the accumulation order; fp16 C / residual / gate tensors add one rounding;
(b) the FastPitch engine with compute = "f16" against the golden vectors recorded from the REFERENCE (tests/golden/fp_*.npz) and against the CPU
    oracle on larger ragged batches — OUTPUTS and LOSSES at north_star's 1e-3 (the bar the bf16 mode misses by 10x); gradients at the bound the
    format gives (profiles/r06_precision_probe.txt: 1e-3 .. 2.5e-2 rel-L2 depending on the tensor);
(c) the same against the oracle's own restatement of the mode's rounding points (oracle/fastpitch.py storage "f16_r32"), tightly."""
import numpy as np
import pytest
import torch

from fp_util import build_engine, grad_report, load_case, rel

pytestmark = pytest.mark.gpu
RTOL = 1e-3


def _lib():
    from xva_trainer_amd import _lib
    _lib.lib.xva_gemm_set_mainloop.restype = int
    return _lib


@pytest.fixture(params=[(1, 1, 1), (2, 1, 1), (2, 0, 1), (3, 1, 1), (4, 1, 1), (5, 1, 1), (-1, 1, 1), (7, 1, 1), (7, 1, 0), (8, 1, 1)],
                ids=["tile128", "tile256", "tile256_lockstep", "tile128x64", "tile64", "tile128x32", "auto", "tile384x128", "tile384x128_lockstep", "tile256x128k32"])
def mainloop(request):
    L = _lib()
    old = L.lib.xva_gemm_set_mainloop(request.param[0])
    oldk = L.lib.xva_gemm_set_kloop(request.param[1])
    oldk3 = L.lib.xva_gemm_set_kloop384(request.param[2])
    yield request.param[0]
    L.lib.xva_gemm_set_mainloop(old)
    L.lib.xva_gemm_set_kloop(oldk)
    L.lib.xva_gemm_set_kloop384(oldk3)


def _h(rows, cols, ld=None, scale=1.0):
    ld = ld or (cols + 7) // 8 * 8
    return (torch.randn(rows, ld, device="cuda") * scale).half(), ld


def _rel(out, ref):
    return ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(300, 200, 256), (257, 136, 1000), (1000, 520, 328), (64, 72, 192), (129, 1000, 4608)])
def test_gemm_f16_nt(mainloop, M, N, K):
    L = _lib()
    torch.manual_seed(M + N + K)
    A, lda = _h(M, K, K + 8)
    B, ldb = _h(N, K)
    C32 = torch.full((M, N), 3.0, device="cuda")
    L.gemm(A, B, C32, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1)
    ref = A[:, :K].double() @ B[:, :K].double().t()
    assert _rel(C32, ref) < 2e-6
    C16 = torch.zeros(M, N + 8, device="cuda", dtype=torch.float16)
    L.gemm(A, B, C16, M, N, K, lda, ldb, N + 8, layout=L.GEMM_NT, compute=1)
    assert _rel(C16[:, :N], ref) < 6e-4                      # one fp16 rounding (2^-11)
    assert C16[:, N:].abs().max().item() == 0.0


@pytest.mark.parametrize("M,N,K", [(300, 200, 256), (130, 136, 864), (1000, 520, 328), (257, 72, 1000), (96, 40, 640), (1000, 32, 448)])
def test_gemm_f16_nn_tn(mainloop, M, N, K):
    L = _lib()
    torch.manual_seed(M * 3 + N + K)
    A, lda = _h(M, K)
    B, ldb = _h(K, N, N + 16)
    C32 = torch.zeros(M, N, device="cuda")
    L.gemm(A, B, C32, M, N, K, lda, ldb, N, layout=L.GEMM_NN, compute=1)
    assert _rel(C32, A[:, :K].double() @ B[:, :N].double()) < 2e-6
    Mt = (M + 7) // 8 * 8
    At, ldat = _h(K, Mt)
    C0 = torch.randn(Mt, N, device="cuda")
    Ct = C0.clone()
    L.gemm(At, B, Ct, Mt, N, K, ldat, ldb, N, layout=L.GEMM_TN, compute=1, accumulate=True)
    assert _rel(Ct, C0.double() + At[:, :Mt].double().t() @ B[:, :N].double()) < 2e-6


def test_gemm_f16_splitk_slabs_and_reduce_epilogue():
    """weight-gradient form through slabs (deterministic), and a split product whose reduce pass stores fp16 with a residual"""
    L = _lib()
    torch.manual_seed(5)
    M, N, K = 384, 1152, 6001        # K not a multiple of 8: rows past K come from the zero page (TN)
    A, lda = _h(K, M, scale=0.3)
    B, ldb = _h(K, N, scale=0.3)
    C0 = torch.randn(M, N, device="cuda")
    Cm = C0.clone()
    ws = torch.empty(8 * M * N, device="cuda")
    L.gemm(A, B, Cm, M, N, K, lda, ldb, N, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws)
    assert _rel(Cm, C0.double() + A[:, :M].double().t() @ B[:, :N].double()) < 3e-6
    C2 = C0.clone()
    L.gemm(A, B, C2, M, N, K, lda, ldb, N, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=0, sk_ws=ws)
    assert torch.equal(C2, Cm)
    # forward form, long reduction into few tiles: split with the epilogue in the reduce kernel (bias, fp32 residual, ReLU, fp16 store)
    M, N, K = 600, 384, 4608
    X, ldx = _h(M, K, scale=0.2)
    W, ldw = _h(N, K, scale=0.2)
    bias = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda")
    Y = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    L.gemm(X, W, Y, M, N, K, ldx, ldw, N, layout=L.GEMM_NT, compute=1, bias=bias, R=R, ldr=N, relu=True, splitk=0, sk_ws=ws)
    ref = torch.relu(X.double() @ W.double().t() + bias.double() + R.double())
    assert _rel(Y, ref) < 6e-4


def test_gemm_f16_epilogue_fp32_residual_gate_and_mixed_outputs(mainloop):
    """the products of the fp16-operand FastPitch schedule: fp16 operands, fp32 residual -> fp32 C ; fp16 gate (hi of the stored activation) -> fp16 C"""
    L = _lib()
    torch.manual_seed(11)
    M, N, K = 700, 384, 1152
    A, lda = _h(M, K, scale=0.3)
    B, ldb = _h(N, K, scale=0.3)
    R = torch.randn(M, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    C32 = torch.zeros(M, N, device="cuda")
    L.gemm(A, B, C32, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, bias=bias, R=R, ldr=N)
    assert _rel(C32, A.double() @ B.double().t() + bias.double() + R.double()) < 3e-6
    G = torch.randn(M, N, device="cuda").half()
    C16 = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    L.gemm(A, B, C16, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, G=G, ldg=N, gate_slope=0.0)
    ref = (A.double() @ B.double().t()) * (G.double() > 0)
    assert _rel(C16, ref) < 6e-4
    # dropout + fp16 gate + fp32 residual -> fp32 C (the row-contiguous epilogue's fp32-residual variants on the 256-wide tiles): the mask is the
    # one the residual-free launch draws, dropped elements carry the residual alone
    Cn = torch.zeros(M, N, device="cuda")
    L.gemm(A, B, Cn, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, bias=bias, G=G, ldg=N, gate_slope=0.0, drop_p=0.25, drop_seed=5, drop_stream=2)
    Cr = torch.zeros(M, N, device="cuda")
    L.gemm(A, B, Cr, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, bias=bias, G=G, ldg=N, gate_slope=0.0, drop_p=0.25, drop_seed=5, drop_stream=2,
           R=R, ldr=N)
    assert _rel(Cr, Cn.double() + R.double()) < 1e-6
    kept = (Cn != 0).float().mean().item()
    assert 0.3 < kept < 0.45                                 # 0.75 kept x about half through the gate
    C16r = torch.zeros(M, N, device="cuda", dtype=torch.float16)
    L.gemm(A, B, C16r, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, bias=bias, R=R, ldr=N)
    assert _rel(C16r, A.double() @ B.double().t() + bias.double() + R.double()) < 6e-4


def test_gemm_rejects_mixed_16bit_formats_and_offpath_f16():
    L = _lib()
    A = torch.randn(64, 64, device="cuda").half()
    B = torch.randn(64, 64, device="cuda").bfloat16()
    Cm = torch.zeros(64, 64, device="cuda")
    with pytest.raises(L.XvaError):
        L.gemm(A, B, Cm, 64, 64, 64, 64, 64, 64, layout=L.GEMM_NT, compute=1)
    Bh = B.half()
    with pytest.raises(L.XvaError):                         # bf16 C with fp16 operands
        L.gemm(A, Bh, torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16), 64, 64, 64, 64, 64, 64, layout=L.GEMM_NT, compute=1)
    with pytest.raises(L.XvaError):                         # K = 8 < 16: no direct-to-LDS tile, and fp16 has no other kernel
        L.gemm(A, Bh, Cm, 64, 64, 8, 64, 64, 64, layout=L.GEMM_NT, compute=1)


# ----------------------------------------------------------------------------------------------------------- the FastPitch engine ----
def _run(eng, flat, grads, batch, stage):
    from xva_trainer_amd.fastpitch.engine import DeviceBatch
    b = DeviceBatch.from_dict(batch, "cuda")
    grads.zero_()
    losses = eng.fwd_loss_bwd(flat, grads, b, stage)
    torch.cuda.synchronize()
    return b, losses.cpu()


@pytest.mark.parametrize("case", ["fp_stage3_small", "fp_stage4_small", "fp_stage2_small"])
def test_f16_against_reference_golden(golden_dir, case):
    """outputs and losses of the reference's own run at 1e-3 (north_star), in the mode the bench's `value_at_tolerance` is timed in"""
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import params as P
    from xva_trainer_amd.fastpitch.lamb import Lamb
    g, batch = load_case(golden_dir, case)
    stage, seed = int(g["stage"]), int(g["seed"])
    sd = ofp.init_state_dict(seed)
    eng, flat, grads = build_engine(sd, "f16")
    b, losses = _run(eng, flat, grads, batch, stage)
    out = eng.outputs(b, stage)
    if stage == 2:
        assert rel(out["log_dur_pred"], torch.from_numpy(g["log_dur_pred"])) < RTOL
        assert rel(out["dur_pred"], torch.from_numpy(g["dur_pred"])) < RTOL
    else:
        assert rel(out["mel_out"], torch.from_numpy(g["mel_out"])) < RTOL
        assert rel(out["pitch_pred"], torch.from_numpy(g["pitch_pred"])) < RTOL
        assert rel(out["pitch_tgt"], torch.from_numpy(g["pitch_tgt"])) < RTOL
        assert rel(out["energy_pred"], torch.from_numpy(g["energy_pred"])) < RTOL
        assert rel(out["energy_tgt"], torch.from_numpy(g["energy_tgt"])) < RTOL
    assert abs(losses[0].item() - float(g["loss"])) < RTOL * abs(float(g["loss"]))
    for mine, ref in zip([losses[1], losses[2], losses[3], losses[4]], g["comps"]):
        assert abs(mine.item() - ref) <= RTOL * max(abs(ref), 1e-6)
    # gradients (loss-scaled in the buffer): per-tensor norms at the format's bound, and the step LAMB takes from them
    assert eng.loss_scale > 1.0
    mine = P.from_flat(eng.unscaled(grads), eng.table)
    keys = [str(k) for k in g["grad_keys"]]
    gtol = 3e-2
    for k, l2 in zip(keys, g["grad_l2"]):
        m = mine[k].double().cpu()
        assert abs(m.norm().item() - l2) <= gtol * max(l2, 1e-12), (k, m.norm().item(), l2)
    have = set(keys)
    for name in mine:
        if name not in have:
            assert mine[name].abs().max().item() == 0.0, name
    opt = Lamb(flat, eng.table, lr=ofp.adjust_learning_rate(int(g["total_iter"])), betas=(0.9, 0.98), eps=1e-9, weight_decay=1e-6)
    opt.step(grads, have, max_grad_norm=1000.0, inv_scale=eng.grad_inv_scale)
    torch.cuda.synchronize()
    assert abs(opt.grad_norm.item() - float(g["grad_norm"])) < gtol * float(g["grad_norm"])
    after = P.from_flat(flat, eng.table)
    for k, l2a in zip(keys, g["param_l2_after"]):
        assert abs(after[k].double().norm().item() - l2a) <= 1e-4 * max(l2a, 1e-12)


@pytest.mark.parametrize("stage", [3, 4, 2])
def test_f16_against_oracle_ragged(stage):
    """Larger ragged batch: both stacks on the fp16 tiles.  fp32 oracle: outputs / loss 1e-3.  The oracle's restatement of the mode ("f16_r32": the engine's
    own rounding points) is matched several times tighter — what is left is accumulation order."""
    from oracle import fastpitch as ofp
    sd = ofp.init_state_dict(77)
    batch = ofp.synth_batch(4, 37, 210, 78)
    names = ofp.trainable_names(sd.keys(), stage)

    def oracle(storage):
        leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
        work = dict(sd); work.update(leaves)
        out = ofp.forward(work, batch, stage, storage=storage)
        l, _ = ofp.loss(out, batch, stage)
        l.backward()
        return out, l, {k: v.grad for k, v in leaves.items() if v.grad is not None}

    out_ref, loss_ref, ref_grads = oracle(None)
    out_q, loss_q, q_grads = oracle("f16_r32")
    eng, flat, grads = build_engine(sd, "f16")
    b, losses = _run(eng, flat, grads, batch, stage)
    out = eng.outputs(b, stage)
    pairs = [("log_dur_pred", 3)] if stage == 2 else [("mel_out", 0), ("pitch_pred", 4), ("energy_pred", 6)]
    for name, idx in pairs:
        e32, eq, oq = rel(out[name], out_ref[idx]), rel(out[name], out_q[idx]), rel(out_q[idx], out_ref[idx])
        print("%s: engine vs fp32 oracle %.2e, engine vs f16_r32 oracle %.2e (oracle f16_r32 vs fp32 %.2e)" % (name, e32, eq, oq))
        assert e32 < RTOL, (name, e32)
    if stage != 2:
        assert torch.equal(out["dec_lens"].cpu().long(), batch["mel_lens"])
    assert abs(losses[0].item() - loss_ref.item()) < RTOL * abs(loss_ref.item())
    g = torch.zeros_like(grads); g.copy_(eng.unscaled(grads))
    bad, worst = grad_report(eng, g, ref_grads, 4e-2)
    print("worst grad tensor vs fp32 oracle:", worst)
    assert not bad, bad[:10]
    from xva_trainer_amd.fastpitch import params as P
    mine = P.from_flat(g, eng.table)
    a = torch.cat([mine[k].double().cpu().flatten() for k in ref_grads])
    r = torch.cat([ref_grads[k].double().flatten() for k in ref_grads])
    cos = (a @ r / (a.norm() * r.norm())).item()
    assert cos > 0.9999, cos


def test_f16_dropout_and_loss_scale_invariance():
    """training mode (p = 0.1) against the oracle with the same hash masks; and the gradient does not depend on the (power-of-two) loss scale"""
    from oracle import fastpitch as ofp
    from xva_trainer_amd.fastpitch import engine as E, params as P
    sd = ofp.init_state_dict(31)
    batch = ofp.synth_batch(3, 29, 150, 32)
    seed = 987654321
    out_ref = ofp.forward(sd, batch, 3, drop=ofp.HashDropout(0.1, seed))
    loss_ref, _ = ofp.loss(out_ref, batch, 3)
    eng = E.FastPitchEngine("cuda", "f16", p_dropout=0.1, seed=seed)
    flat = torch.zeros(eng.total, device="cuda")
    P.to_flat(sd, eng.table, flat)
    grads = torch.zeros_like(flat)
    b, losses = _run(eng, flat, grads, batch, 3)
    out = eng.outputs(b, 3)
    assert rel(out["mel_out"], out_ref[0]) < RTOL
    assert rel(out["pitch_pred"], out_ref[4]) < RTOL
    assert abs(losses[0].item() - loss_ref.item()) < RTOL * abs(loss_ref.item())
    auto = eng.loss_scale
    g_auto = eng.unscaled(grads).clone()
    eng.step -= 1                                   # the same masks again
    eng.set_loss_scale(auto / 16)
    _run(eng, flat, grads, batch, 3)
    g_16 = eng.unscaled(grads)
    assert ((g_16 - g_auto).norm() / g_auto.norm()).item() < 2e-3      # only the fp16 gradient buffers' subnormal tail moves
