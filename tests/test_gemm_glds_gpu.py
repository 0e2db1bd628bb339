"""GPU parity of the direct-to-LDS GEMM main loop (csrc/gemm_glds.h: global_load_lds staging, LDS transpose reads, 64-deep
K tiles, 128x128 and 256x256 tiles) against fp64 references built from the same bf16-rounded operands.  With an fp32 C the
only difference to the reference is fp32 accumulation order, so the bound is tight; bf16 C adds one rounding."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from xva_trainer_amd import _lib
    _lib.lib.xva_gemm_set_mainloop.restype = int
    return _lib


# (mainloop mode, K loop of the 256x256 tile, K loop of the 384x128 tile): 2/0 = two barriers per K tile, 2/1 = staggered wave groups; 8 = 256x128 tile with a
# 32-deep K tile; 7 = the 384x128 tile with the staggered (1) or the lock-step (0) loop
@pytest.fixture(params=[(1, 1, 1), (2, 1, 1), (2, 0, 1), (3, 1, 1), (4, 1, 1), (5, 1, 1), (-1, 1, 1), (6, 1, 1), (7, 1, 1), (7, 1, 0), (8, 1, 1)],
                ids=["tile128", "tile256", "tile256_lockstep", "tile128x64", "tile64", "tile128x32", "auto", "auto_no_resident_conv", "tile384x128",
                     "tile384x128_lockstep", "tile256x128k32"])
def mainloop(request):
    L = _lib()
    old = L.lib.xva_gemm_set_mainloop(request.param[0])
    oldk = L.lib.xva_gemm_set_kloop(request.param[1])
    oldk3 = L.lib.xva_gemm_set_kloop384(request.param[2])
    yield request.param[0]
    L.lib.xva_gemm_set_mainloop(old)
    L.lib.xva_gemm_set_kloop(oldk)
    L.lib.xva_gemm_set_kloop384(oldk3)


def _bf(rows, cols, ld=None, scale=1.0):
    ld = ld or (cols + 7) // 8 * 8
    return (torch.randn(rows, ld, device="cuda") * scale).bfloat16(), ld


def _rel(out, ref):
    return ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(300, 200, 256), (257, 136, 1000), (1000, 520, 328), (64, 72, 192), (129, 1000, 4608)])
def test_nt_ragged(mainloop, M, N, K):
    L = _lib()
    torch.manual_seed(M + N + K)
    A, lda = _bf(M, K, K + 8)
    B, ldb = _bf(N, K)
    C32 = torch.full((M, N), 3.0, device="cuda")
    L.gemm(A, B, C32, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1)
    ref = A[:, :K].double() @ B[:, :K].double().t()
    assert _rel(C32, ref) < 2e-6
    C16 = torch.zeros(M, N + 8, device="cuda", dtype=torch.bfloat16)
    L.gemm(A, B, C16, M, N, K, lda, ldb, N + 8, layout=L.GEMM_NT, compute=1)
    assert _rel(C16[:, :N], ref) < 5e-3
    assert C16[:, N:].abs().max().item() == 0.0


@pytest.mark.parametrize("M,N,K", [(300, 200, 256), (130, 136, 864), (1000, 520, 328), (257, 72, 1000), (700, 64, 384), (96, 40, 640), (1000, 32, 448), (520, 16, 192)])
def test_nn_and_tn_ragged(mainloop, M, N, K):
    L = _lib()
    torch.manual_seed(M * 3 + N + K)
    A, lda = _bf(M, K)
    B, ldb = _bf(K, N, N + 16)
    C32 = torch.zeros(M, N, device="cuda")
    L.gemm(A, B, C32, M, N, K, lda, ldb, N, layout=L.GEMM_NN, compute=1)
    assert _rel(C32, A[:, :K].double() @ B[:, :N].double()) < 2e-6
    # TN: both operands k-major
    Mt = (M + 7) // 8 * 8
    At, ldat = _bf(K, Mt)
    C0 = torch.randn(Mt, N, device="cuda")
    Ct = C0.clone()
    L.gemm(At, B, Ct, Mt, N, K, ldat, ldb, N, layout=L.GEMM_TN, compute=1, accumulate=True)
    ref = C0.double() + At[:, :Mt].double().t() @ B[:, :N].double()
    assert _rel(Ct, ref) < 2e-6


@pytest.mark.parametrize("splitk,slabs", [(3, False), (5, True), (16, True)])
def test_tn_splitk_atomics_and_slabs(mainloop, splitk, slabs):
    """Weight-gradient form: huge K, fp32 accumulate into C; split-K through fp32 atomics or through scratch slabs + reduce."""
    L = _lib()
    torch.manual_seed(5)
    M, N, K = 384, 1152, 6000
    A, lda = _bf(K, M, scale=0.3)
    B, ldb = _bf(K, N, scale=0.3)
    C0 = torch.randn(M, N, device="cuda")
    Cm = C0.clone()
    ws = torch.empty(splitk * M * N, device="cuda") if slabs else None
    L.gemm(A, B, Cm, M, N, K, lda, ldb, N, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=splitk, sk_ws=ws)
    ref = C0.double() + A[:, :M].double().t() @ B[:, :N].double()
    assert _rel(Cm, ref) < 3e-6
    if slabs:   # the slab path is deterministic
        C2 = C0.clone()
        L.gemm(A, B, C2, M, N, K, lda, ldb, N, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=splitk, sk_ws=ws)
        assert torch.equal(C2, Cm)


@pytest.mark.parametrize("T,splitk", [(256, 1), (200, 1), (98, 3), (130, 0)])
def test_tn_kblocks_and_column_segments(mainloop, T, splitk):
    """K-block addressing (per-item weight gradients merged into one launch; block lengths need not be multiples of the 64-row K
    tile: the tail tile of each block is zero-filled) + TN column segments (dilated taps)."""
    L = _lib()
    torch.manual_seed(6)
    items, Cout, Cin, k, d = 3, 136, 64, 3, 2
    PAD = 4
    x = torch.zeros(items, T + 2 * PAD, Cin, device="cuda", dtype=torch.bfloat16)
    x[:, PAD:PAD + T] = torch.randn(items, T, Cin, device="cuda").bfloat16()
    dy = torch.randn(items, T, Cout, device="cuda").bfloat16()
    dW = torch.zeros(Cout, k * Cin, device="cuda")
    # dW[co][j*Cin + ci] = sum_{b,t} dy[b,t,co] * x[b, t + (j-1)*d, ci]
    ws = torch.zeros(8 * Cout * k * Cin, device="cuda")
    L.gemm(dy, x, dW, Cout, k * Cin, items * T, Cout, Cin, k * Cin, layout=L.GEMM_TN, compute=1, accumulate=True, splitk=splitk, sk_ws=ws,
           b_offset=(PAD - d) * Cin, seglen=Cin, seg0=0, segstride=d * Cin - Cin,
           kb_len=T, kb_sA=T * Cout, kb_sB=(T + 2 * PAD) * Cin)
    xr = x[:, PAD:PAD + T].double().transpose(1, 2)
    w = torch.zeros(Cout, Cin, k, dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv1d(xr, w, padding=d, dilation=d)
    (y * dy.double().transpose(1, 2)).sum().backward()
    ref = w.grad.permute(0, 2, 1).reshape(Cout, k * Cin)
    assert _rel(dW, ref) < 3e-6


@pytest.mark.parametrize("Cin,Cout,d", [(64, 136, 1), (128, 72, 3), (32, 32, 5), (16, 64, 2), (64, 64, 5), (128, 128, 1), (32, 64, 3)])
def test_conv_tap_segments_fwd_bwd_data(mainloop, Cin, Cout, d):
    """Dilated k=3 conv as implicit GEMM: A tap segments (NT forward) and B row segments (NN backward-data)."""
    L = _lib()
    torch.manual_seed(7)
    T, k, PAD = 700, 3, 8
    xs = torch.zeros(T + 2 * PAD, Cin, device="cuda", dtype=torch.bfloat16)
    xs[PAD:PAD + T] = torch.randn(T, Cin, device="cuda").bfloat16()
    W = (torch.randn(Cout, Cin, k, device="cuda") * 0.1).bfloat16()
    Wt = W.permute(0, 2, 1).contiguous().view(Cout, k * Cin)            # tap-major
    bias = torch.randn(Cout, device="cuda")
    y = torch.zeros(T, Cout, device="cuda")
    L.gemm(xs, Wt, y, T, Cout, k * Cin, Cin, k * Cin, Cout, layout=L.GEMM_NT, compute=1, bias=bias,
           a_offset=(PAD - d) * Cin, a_seglen=Cin, a_segadj=d * Cin - Cin)
    ref = F.conv1d(xs[PAD:PAD + T].double().t()[None], W.double(), bias.double(), padding=d, dilation=d)[0].t()
    assert _rel(y, ref) < 3e-6
    dys = torch.zeros(T + 2 * PAD, Cout, device="cuda", dtype=torch.bfloat16)
    dys[PAD:PAD + T] = torch.randn(T, Cout, device="cuda").bfloat16()
    dx = torch.zeros(T, Cin, device="cuda")
    L.gemm(dys, Wt, dx, T, Cin, k * Cout, Cout, k * Cin, Cin, layout=L.GEMM_NN, compute=1, a_offset=(PAD - d) * Cout,
           a_seglen=Cout, a_segadj=d * Cout - Cout, seglen=Cout, seg0=(k - 1) * Cin, segstride=-Cin)
    xr = xs[PAD:PAD + T].double().t()[None].requires_grad_(True)
    F.conv1d(xr, W.double(), padding=d, dilation=d).backward(dys[PAD:PAD + T].double().t()[None])
    assert _rel(dx, xr.grad[0].t()) < 3e-6


@pytest.mark.parametrize("C,k,d", [(32, 11, 5), (64, 7, 3), (128, 3, 1)])
def test_resblock_conv_resident_input(mainloop, C, k, d):
    """HiFi-GAN ResBlock1 convolution (models.py:62-66): y = conv_k,d(lrelu(x)) + bias + residual on 32/64/128 channels — the shape the
    resident-input kernel serves (input rows loaded once, taps read shifted LDS rows), and its backward-data with the lrelu gate."""
    L = _lib()
    torch.manual_seed(C + k)
    nseq, T, PAD = 3, 500, 32
    Hp = T + 2 * PAD
    P_ = d * (k - 1) // 2
    xs = torch.zeros(nseq * Hp + 2 * PAD, C, device="cuda", dtype=torch.bfloat16)      # guard rows before / after
    xv = xs[PAD:PAD + nseq * Hp].view(nseq, Hp, C)
    xv[:, PAD:PAD + T] = torch.randn(nseq, T, C, device="cuda").bfloat16()
    W = (torch.randn(C, C, k, device="cuda") * 0.05).bfloat16()
    Wt = W.permute(0, 2, 1).contiguous().view(C, k * C)
    bias = torch.randn(C, device="cuda")
    R = torch.randn(nseq * Hp, C, device="cuda").bfloat16()
    y = torch.zeros(nseq * Hp, C, device="cuda", dtype=torch.bfloat16)
    rows = nseq * Hp
    L.gemm(xs, Wt, y, rows, C, k * C, C, k * C, C, layout=L.GEMM_NT, compute=1, bias=bias, R=R, ldr=C, a_lrelu=0.1,
           a_offset=(PAD - P_) * C, a_seglen=C, a_segadj=d * C - C, mask_mode=L.MASK_PAD, Tp=Hp, mask_pad=PAD, mask_len=T)
    lr = lambda t: torch.where(t > 0, t, (t.float() * 0.1).bfloat16())
    xin = lr(xv[:, PAD:PAD + T]).double().transpose(1, 2)
    ref = torch.nn.functional.conv1d(xin, W.double(), bias.double(), padding=P_, dilation=d).transpose(1, 2) + R.view(nseq, Hp, C)[:, PAD:PAD + T].double()
    out = y.view(nseq, Hp, C)
    assert _rel(out[:, PAD:PAD + T], ref) < 6e-3
    assert out[:, :PAD].abs().max().item() == 0.0 and out[:, PAD + T:].abs().max().item() == 0.0
    # backward-data: dx = lrelu'(x) * conv^T(dy)
    dys = torch.zeros_like(xs)
    dys[PAD:PAD + rows].view(nseq, Hp, C)[:, PAD:PAD + T] = torch.randn(nseq, T, C, device="cuda").bfloat16()
    dx = torch.zeros(rows, C, device="cuda")
    L.gemm(dys, Wt, dx, rows, C, k * C, C, k * C, C, layout=L.GEMM_NN, compute=1, a_offset=(PAD - P_) * C, a_seglen=C, a_segadj=d * C - C,
           seglen=C, seg0=(k - 1) * C, segstride=-C, G=xs[PAD:], ldg=C, gate_slope=0.1, mask_mode=L.MASK_PAD, Tp=Hp, mask_pad=PAD, mask_len=T)
    xr = xv[:, PAD:PAD + T].double().transpose(1, 2).requires_grad_(True)
    xa = torch.where(xr > 0, xr, xr * 0.1)
    torch.nn.functional.conv1d(xa, W.double(), padding=P_, dilation=d).backward(dys[PAD:PAD + rows].view(nseq, Hp, C)[:, PAD:PAD + T].double().transpose(1, 2))
    assert _rel(dx.view(nseq, Hp, C)[:, PAD:PAD + T], xr.grad.transpose(1, 2)) < 3e-6
    # the engine's form of the same product: taps walk BACKWARDS over dY (a_segadj < -C), weights in natural tap order
    dx2 = torch.zeros(rows, C, device="cuda")
    L.gemm(dys, Wt, dx2, rows, C, k * C, C, k * C, C, layout=L.GEMM_NN, compute=1, a_offset=(PAD + P_) * C, a_seglen=C, a_segadj=-d * C - C,
           seglen=C, seg0=0, segstride=C, G=xs[PAD:], ldg=C, gate_slope=0.1, mask_mode=L.MASK_PAD, Tp=Hp, mask_pad=PAD, mask_len=T)
    assert _rel(dx2.view(nseq, Hp, C)[:, PAD:PAD + T], xr.grad.transpose(1, 2)) < 3e-6


def test_epilogue_options(mainloop):
    """bias, alpha, dropout, gate, residual, ReLU, row mask, accumulate, bf16 / fp32 / transposed C — the documented order."""
    L = _lib()
    torch.manual_seed(8)
    Bn, T, N, K = 3, 100, 264, 320
    Tp = T + 2
    M = Bn * Tp
    A, lda = _bf(M, K)
    Bw, ldb = _bf(N, K)
    bias = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda").bfloat16()
    G = torch.randn(M, N, device="cuda").bfloat16()
    lens = torch.tensor([100, 37, 64], device="cuda", dtype=torch.int32)
    acc = A.double() @ Bw.double().t()
    v = 0.5 * (acc + bias.double())
    v = torch.where(G.double() > 0, v, v * 0.1)
    v = v + 2.0 * R.double()
    v = torch.relu(v)
    t = torch.arange(M, device="cuda") % Tp
    b = torch.arange(M, device="cuda") // Tp
    live = (t >= 1) & (t < Tp - 1) & ((t - 1) < lens[b])
    ref = v * live[:, None]
    for cdt in (torch.float32, torch.bfloat16):
        Cm = torch.full((M, N), 5.0, device="cuda", dtype=cdt)
        L.gemm(A, Bw, Cm, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, bias=bias, alpha=0.5, beta=2.0, R=R, ldr=N, G=G, ldg=N,
               gate_slope=0.1, relu=True, mask_mode=L.MASK_LEN, lens=lens, Tp=Tp)
        assert _rel(Cm, ref) < (3e-6 if cdt == torch.float32 else 5e-3)
    # accumulate + transposed store
    Ct = torch.ones(N, M, device="cuda")
    L.gemm(A, Bw, Ct, M, N, K, lda, ldb, M, layout=L.GEMM_NT, compute=1, accumulate=True, c_trans=1)
    assert _rel(Ct, 1.0 + acc.t()) < 3e-6
    # dropout in the epilogue: kept elements scaled by 1/(1-p), dropped exactly 0, before the residual
    Cd = torch.zeros(M, N, device="cuda")
    L.gemm(A, Bw, Cd, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, drop_p=0.25, drop_seed=99, drop_stream=3)
    Cg = torch.zeros(M, N, device="cuda")
    old = L.lib.xva_gemm_set_mainloop(0)
    L.gemm(A, Bw, Cg, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, drop_p=0.25, drop_seed=99, drop_stream=3)
    L.lib.xva_gemm_set_mainloop(old)
    kept = Cd != 0
    assert abs(kept.float().mean().item() - 0.75) < 0.01
    assert torch.equal(kept, Cg != 0)                                  # same mask as the general kernel
    assert _rel(Cd[kept], (acc / 0.75)[kept]) < 3e-6


def test_leaky_relu_fused_into_operands(mainloop):
    """HiFi-GAN's generator feeds lrelu(x) to its convolutions and their weight gradients: the activation is applied to the MFMA
    fragments (models.py:62-66).  x * slope is re-rounded to bf16 exactly like the reference rounds lrelu(x) stored in bf16."""
    L = _lib()
    torch.manual_seed(10)
    M, N, K = 520, 136, 448
    A, lda = _bf(M, K)
    Bw, ldb = _bf(N, K)
    lr = lambda t, s: torch.where(t > 0, t, (t.float() * s).bfloat16())
    Cm = torch.zeros(M, N, device="cuda")
    L.gemm(A, Bw, Cm, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, a_lrelu=0.1)
    assert _rel(Cm, lr(A, 0.1).double() @ Bw.double().t()) < 3e-6
    Bn, ldbn = _bf(K, N)
    L.gemm(A, Bn, Cm, M, N, K, lda, ldbn, N, layout=L.GEMM_NN, compute=1, a_lrelu=0.2, b_lrelu=0.01)
    assert _rel(Cm, lr(A, 0.2).double() @ lr(Bn, 0.01)[:, :N].double()) < 3e-6
    At, ldat = _bf(K, M)
    Ct = torch.zeros(M, N, device="cuda")
    L.gemm(At, Bn, Ct, M, N, K, ldat, ldbn, N, layout=L.GEMM_TN, compute=1, accumulate=True, b_lrelu=0.1)
    assert _rel(Ct, At[:, :M].double().t() @ lr(Bn, 0.1)[:, :N].double()) < 3e-6


def test_batched_two_level(mainloop):
    L = _lib()
    torch.manual_seed(9)
    b1, b2, M, N, K = 2, 3, 140, 96, 256
    A = torch.randn(b1, b2, M, K, device="cuda").bfloat16()
    Bw = torch.randn(b1, b2, N, K, device="cuda").bfloat16()
    Cm = torch.zeros(b1, b2, M, N, device="cuda")
    L.gemm(A, Bw, Cm, M, N, K, K, K, N, layout=L.GEMM_NT, compute=1, batch=b1, sA=b2 * M * K, sB=b2 * N * K, sC=b2 * M * N,
           batch2=b2, sA2=M * K, sB2=N * K, sC2=M * N)
    assert _rel(Cm, A.double() @ Bw.double().transpose(-1, -2)) < 2e-6


@pytest.mark.parametrize("M,N,K,cdt", [(300, 200, 256, torch.bfloat16), (257, 136, 320, torch.bfloat16), (1000, 64, 384, torch.float32), (130, 92, 128, torch.bfloat16)])
def test_second_output_activated_copy(mainloop, M, N, K, cdt):
    """C2 = lrelu(C): the activated copy a producer stores next to its raw output (HiFi-GAN residual stream, models.py:41-48) —
    through the row-contiguous epilogue (N % 8 == 0) and the 4-column one (N % 4 == 0); the general kernel refuses it loudly."""
    L = _lib()
    torch.manual_seed(M + N)
    A, lda = _bf(M, K)
    B, ldb = _bf(N, K)
    bias = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda").to(cdt)
    Cm = torch.zeros(M, N, device="cuda", dtype=cdt)
    C2 = torch.full((M, N), 7.0, device="cuda", dtype=cdt)
    L.gemm(A, B, Cm, M, N, K, lda, ldb, N, compute=1, bias=bias, R=R, ldr=N, C2=C2.data_ptr(), c2_slope=0.1)
    ref = A[:, :K].double() @ B[:, :K].double().t() + bias.double() + R.double()
    assert _rel(Cm, ref) < (6e-3 if cdt == torch.bfloat16 else 1e-5)
    act = torch.where(Cm.double() > 0, Cm.double(), Cm.double() * 0.1)      # C2 is the LeakyReLU of the value C holds (before C's rounding)
    assert _rel(C2, act) < (8e-3 if cdt == torch.bfloat16 else 1e-6)
    assert bool(((C2.double() > 0) == (Cm.double() > 0)).all())
    L.lib.xva_gemm_set_mainloop(0)
    with pytest.raises(L.XvaError):
        L.gemm(A, B, Cm, M, N, K, lda, ldb, N, compute=1, C2=C2.data_ptr(), c2_slope=0.1)


@pytest.mark.parametrize("C,Cout,groups,k,s", [(128, 128, 4, 41, 2), (128, 256, 16, 41, 2), (256, 512, 16, 41, 4), (512, 1024, 16, 41, 4), (1024, 1024, 16, 41, 1),
                                               (64, 64, 1, 5, 2), (256, 128, 2, 7, 4), (32, 48, 2, 3, 1)])
def test_grouped_strided_conv_resident_input(mainloop, C, Cout, groups, k, s):
    """HiFi-GAN MultiScaleDiscriminator convolutions (models.py:203-216: k = 41, stride 2 / 4, groups 4 / 16) in the form hg_conv.h issues
    them — per-item batches, the group index as the second batch level, lda = stride * C (a_rowpitch = C) — forward and the polyphase
    backward-data (one stride-1 problem per input phase), against fp64 F.conv1d.  With 32 / 64 channels per group these take the
    resident-input kernel (input rows of a group loaded once, strided MFMA fragments)."""
    L = _lib()
    torch.manual_seed(C + k + s)
    nseq, T, PAD = 3, 1000, 24
    P_ = (k - 1) // 2
    Cig, Cog = C // groups, Cout // groups
    To = (T + 2 * P_ - (k - 1) - 1) // s + 1
    Hp, Hpo = T + 2 * PAD, To + 2 * PAD
    xs = torch.zeros(nseq * Hp + 2 * PAD, C, device="cuda", dtype=torch.bfloat16)
    xv = xs[PAD:PAD + nseq * Hp].view(nseq, Hp, C)
    xv[:, PAD:PAD + T] = torch.randn(nseq, T, C, device="cuda").bfloat16()
    W = (torch.randn(Cout, Cig, k, device="cuda") * 0.05).bfloat16()
    Wt = W.permute(0, 2, 1).contiguous().view(Cout, k * Cig)                       # [G][Cog][k * Cig] tap-major
    bias = torch.randn(Cout, device="cuda")
    y = torch.zeros(nseq * Hpo, Cout, device="cuda", dtype=torch.bfloat16)
    L.gemm(xs, Wt, y, To, Cog, k * Cig, s * C, k * Cig, Cout, layout=L.GEMM_NT, compute=1, bias=bias, act=L.ACT_LRELU, act_slope=0.1,
           a_offset=(PAD + PAD - P_) * C, a_seglen=Cig, a_segadj=C - Cig, a_rowpitch=C, batch=nseq, sA=Hp * C, sC=Hpo * Cout,
           batch2=groups, sA2=Cig, sB2=Cog * k * Cig, sC2=Cog, sbias2=Cog, b_offset=0)
    ref = F.leaky_relu(F.conv1d(xv[:, PAD:PAD + T].double().transpose(1, 2), W.double(), bias.double(), stride=s, padding=P_, groups=groups), 0.1)
    out = y.view(nseq, Hpo, Cout)[:, :To]                                          # the per-item GEMM writes rows [0, To) of each item's slab
    assert _rel(out, ref.transpose(1, 2)) < 6e-3
    # backward-data, phase by phase (hg_conv_bwd_data): dX[s q + psi] = sum_m dY[q + c0 - m] W[:, :, j0 + m s]
    dys = torch.zeros(nseq * Hpo + 2 * PAD, Cout, device="cuda", dtype=torch.bfloat16)
    dyv = dys[PAD:PAD + nseq * Hpo].view(nseq, Hpo, Cout)
    dyv[:, PAD:PAD + To] = torch.randn(nseq, To, Cout, device="cuda").bfloat16()
    dx = torch.zeros(nseq * Hp, C, device="cuda")
    for psi in range(s):
        j0, c0 = (psi + P_) % s, (psi + P_) // s
        ntap = (k - j0 + s - 1) // s
        Q = (T - psi + s - 1) // s
        L.gemm(dys, Wt, dx, Q, Cig, ntap * Cog, Cout, k * Cig, s * C, layout=L.GEMM_NN, compute=1,
               a_offset=(PAD + PAD + c0) * Cout, a_seglen=Cog, a_segadj=-Cout - Cog, seglen=Cog, seg0=j0 * Cig, segstride=s * Cig,
               batch=nseq, sA=Hpo * Cout, sC=Hp * C, batch2=groups, sA2=Cog, sB2=Cog * k * Cig, sC2=Cig, c_offset=(PAD + psi) * C)
    xr = xv[:, PAD:PAD + T].double().transpose(1, 2).requires_grad_(True)
    F.conv1d(xr, W.double(), stride=s, padding=P_, groups=groups).backward(dyv[:, PAD:PAD + To].double().transpose(1, 2))
    assert _rel(dx.view(nseq, Hp, C)[:, PAD:PAD + T], xr.grad.transpose(1, 2)) < 3e-6


@pytest.mark.parametrize("layout", ["nt", "nn"])
def test_splitk_with_the_full_epilogue_in_the_reduce_pass(layout):
    """Forward / backward-data products with a long reduction and a small tile grid (FastPitch's encoder feed-forward: 4 864 rows x 384 columns
    over K = 4 608) may be split along K when the caller passes splitk = 0 and slab scratch: the reduce pass applies the WHOLE epilogue (bias,
    dropout, gate, residual, ReLU, row mask, bf16 store).  Against the unsplit product: the same values up to fp32 summation order (one bf16 ulp
    on a bf16 C), identical dropout / mask pattern."""
    L = _lib()
    torch.manual_seed(5)
    M, N, K, Tp = 4864, 384, 4608, 152
    A, lda = _bf(M, K, scale=0.05)
    if layout == "nt":
        B, ldb = _bf(N, K)
        kw = dict(layout=L.GEMM_NT)
    else:
        B, ldb = _bf(K, N)
        kw = dict(layout=L.GEMM_NN)
    bias = torch.randn(N, device="cuda")
    R = torch.randn(M, N, device="cuda").bfloat16()
    G = torch.randn(M, N, device="cuda").bfloat16()
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    common = dict(compute=1, bias=bias, relu=True, R=R, ldr=N, G=G, ldg=N, gate_slope=0.0, mask_mode=L.MASK_PAD, Tp=Tp, mask_pad=1, mask_len=Tp - 2,
                  drop_p=0.1, drop_seed=77, drop_stream=3, **kw)
    out = {}
    for tag, extra in (("one", dict(splitk=1)), ("split", dict(splitk=0, sk_ws=ws))):
        for cdt in (torch.bfloat16, torch.float32):
            Cm = torch.full((M, N), 7.0, device="cuda", dtype=cdt)
            L.gemm(A, B, Cm, M, N, K, lda, ldb, N, **common, **extra)
            out[(tag, cdt)] = Cm.float()
    for cdt, tol in ((torch.float32, 2e-5), (torch.bfloat16, 1.2e-2)):
        a, b = out[("one", cdt)], out[("split", cdt)]
        assert torch.equal(a == 0, b == 0) or ((a == 0) != (b == 0)).float().mean().item() < 1e-4      # same ReLU / dropout / mask zeros (up to ReLU gates at rounding)
        assert ((a - b).abs().max() / a.abs().max()).item() < tol
    pad_rows = torch.arange(M, device="cuda") % Tp
    assert out[("split", torch.float32)][(pad_rows == 0) | (pad_rows == Tp - 1)].abs().max().item() == 0.0


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
def test_nt_wholeline_pieces_are_bit_identical_to_the_32_deep_tiles(dt):
    """round 6: NT products on the 256 x 256 tile fetch their operands as 8-row x 128-byte DMA pieces into a ring of five operand units
    (xva_gemm_glds8w_kernel) when K % 64 == 0 and the tap segments are multiples of 64.  The same 32-deep phases in the same order: every result is
    bit-identical to the 16-row x 64-byte form (xva_gemm_set_wholeline(0)) — plain products, ragged edges, short and long K, the conv tap-segment form,
    split-K slabs, two batch levels, the epilogue with gate + dropout + residual — and right against fp64."""
    L = _lib()
    L.lib.xva_gemm_set_wholeline.restype = int
    old = L.lib.xva_gemm_set_mainloop(2)
    oldk = L.lib.xva_gemm_set_kloop(1)
    torch.manual_seed(3)
    mk = lambda r, c, s=1.0: (torch.randn(r, c, device="cuda") * s).to(dt)
    cdt = dt
    cases = []
    for M, N, K in [(700, 520, 64), (257, 264, 128), (1000, 1536, 1152), (300, 256, 4608), (64, 72, 192)]:
        A, B = mk(M, K, 0.5), mk(N, K, 0.5)
        cases.append(("plain %dx%dx%d" % (M, N, K), dict(args=(A, B, M, N, K, K, K), kw={}, ref=A.double() @ B.double().t())))
    # conv k = 3 over 384 channels (FastPitch conv1): three tap segments of 384 = 6 x 64
    T, Cin, Cout, PAD = 900, 384, 520, 8
    xs = torch.zeros(T + 2 * PAD, Cin, device="cuda", dtype=dt); xs[PAD:PAD + T] = mk(T, Cin)
    W = mk(Cout, Cin * 3, 0.05)
    refc = sum(xs[PAD - 1 + j:PAD - 1 + j + T].double() @ W[:, j * Cin:(j + 1) * Cin].double().t() for j in range(3))
    cases.append(("conv taps", dict(args=(xs, W, T, Cout, 3 * Cin, Cin, 3 * Cin), kw=dict(a_offset=(PAD - 1) * Cin, a_seglen=Cin, a_segadj=0), ref=refc)))
    try:
        for name, c in cases:
            A, B, M, N, K, lda, ldb = c["args"]
            outs = {}
            G = (torch.randn(M, N, device="cuda") > 0).to(cdt); R = torch.randn(M, N, device="cuda").to(cdt); bias = torch.randn(N, device="cuda")
            for mode in (1, 0):
                L.lib.xva_gemm_set_wholeline(mode)
                C32 = torch.full((M, N), 7.0, device="cuda")
                L.gemm(A, B, C32, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, **c["kw"])
                C16 = torch.full((M, N), 7.0, device="cuda", dtype=cdt)
                L.gemm(A, B, C16, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=1, bias=bias, G=G, ldg=N, gate_slope=0.0, R=R, ldr=N,
                       drop_p=0.1, drop_seed=5, drop_stream=1, **c["kw"])
                outs[mode] = (C32, C16)
            assert torch.equal(outs[1][0], outs[0][0]) and torch.equal(outs[1][1], outs[0][1]), name
            assert _rel(outs[1][0], c["ref"]) < 3e-6, name
            kept = (outs[1][1].float() != R.float()).float().mean().item()       # a dropped or gated-off product leaves the residual alone
            assert 0.35 < kept < 0.55, (name, kept)
        # split-K through slabs (forward form with the epilogue in the reduce pass) and two batch levels
        M, N, K = 520, 264, 4608
        A, B = mk(M, K, 0.3), mk(N, K, 0.3)
        ws = torch.empty(16 * M * N, device="cuda")
        Ab, Bb = mk(6 * 300, 256, 0.5), mk(6 * 260, 256, 0.5)
        res = {}
        for mode in (1, 0):
            L.lib.xva_gemm_set_wholeline(mode)
            Cs = torch.zeros(M, N, device="cuda")
            L.gemm(A, B, Cs, M, N, K, K, K, N, layout=L.GEMM_NT, compute=1, splitk=0, sk_ws=ws)
            Cb = torch.zeros(6, 300, 260, device="cuda")
            L.gemm(Ab, Bb, Cb, 300, 260, 256, 256, 256, 260, layout=L.GEMM_NT, compute=1, batch=3, batch2=2, sA=2 * 300 * 256, sA2=300 * 256, sB=2 * 260 * 256,
                   sB2=260 * 256, sC=2 * 300 * 260, sC2=300 * 260)
            res[mode] = (Cs, Cb)
        assert torch.equal(res[1][0], res[0][0]) and torch.equal(res[1][1], res[0][1])
        assert _rel(res[1][0], A.double() @ B.double().t()) < 3e-6
        refb = torch.bmm(Ab.view(6, 300, 256).double(), Bb.view(6, 260, 256).double().transpose(1, 2))
        assert _rel(res[1][1], refb) < 3e-6
    finally:
        L.lib.xva_gemm_set_wholeline(1)
        L.lib.xva_gemm_set_mainloop(old)
        L.lib.xva_gemm_set_kloop(oldk)
