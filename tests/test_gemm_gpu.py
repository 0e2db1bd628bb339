"""GPU parity of the MFMA GEMM (C ABI xva_gemm) against fp64 torch matmul / conv1d.

Covers all three layouts, both compute modes, ragged sizes, batching, the overlapping-row
k=3 convolution form (forward, backward-data with tap segments, backward-weight with
split-K) and every epilogue option.  Inputs are asymmetric random (transpose-detecting).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

class _Tol(dict):
    """relative to max |ref|: fp32-exact MFMA (or, under the fp32_products fixture, the bf16 hi + lo split: 16 mantissa bits per operand) vs bf16-input MFMA"""
    split = False

    def __getitem__(self, compute):
        return {0: 6e-5 if self.split else 2e-5, 1: 2e-2}[compute]


TOL = _Tol()


@pytest.fixture(autouse=True, params=[0, 1], ids=["fp32_exact", "fp32_split3"])
def fp32_products(request):
    """every case twice: compute == 0 products by the exact fp32 MFMA and by three bf16 MFMAs on split operands (xva_gemm_set_fp32_products)"""
    from xva_trainer_amd import _lib
    old = _lib.lib.xva_gemm_set_fp32_products(request.param)
    TOL.split = bool(request.param)
    yield request.param
    _lib.lib.xva_gemm_set_fp32_products(old)
    TOL.split = False


def _lib():
    from xva_trainer_amd import _lib
    return _lib


def _alloc(rows, cols, ld=None, dev="cuda"):
    ld = ld or (cols + 3) // 4 * 4
    buf = torch.randn(rows, ld, device=dev, dtype=torch.float32)
    return buf, ld


def _relerr(out, ref):
    return ((out.double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)).item()


@pytest.mark.parametrize("compute", [0, 1])
@pytest.mark.parametrize("M,N,K", [(200, 150, 70), (128, 128, 32), (1, 1, 256), (300, 80, 513), (257, 192, 384)])
def test_nt(compute, M, N, K):
    L = _lib()
    torch.manual_seed(M * 7 + N)
    A, lda = _alloc(M, K)
    B, ldb = _alloc(N, K)
    Cm = torch.full((M, N), 7.0, device="cuda")
    L.gemm(A, B, Cm, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=compute)
    ref = A[:, :K].double() @ B[:, :K].double().t()
    assert _relerr(Cm, ref) < TOL[compute]


@pytest.mark.parametrize("compute", [0, 1])
@pytest.mark.parametrize("M,N,K", [(200, 152, 70), (130, 64, 862), (64, 384, 80)])
def test_nn(compute, M, N, K):
    L = _lib()
    torch.manual_seed(1)
    A, lda = _alloc(M, K)
    B, ldb = _alloc(K, N)
    Cm = torch.zeros(M, N, device="cuda")
    L.gemm(A, B, Cm, M, N, K, lda, ldb, N, layout=L.GEMM_NN, compute=compute)
    ref = A[:, :K].double() @ B[:, :N].double()
    assert _relerr(Cm, ref) < TOL[compute]


@pytest.mark.parametrize("compute", [0, 1])
@pytest.mark.parametrize("splitk", [1, 4])
@pytest.mark.parametrize("M,N,K", [(192, 384, 1000), (1, 256, 300), (80, 384, 517)])
def test_tn_splitk(compute, splitk, M, N, K):
    L = _lib()
    torch.manual_seed(2)
    A, lda = _alloc(K, M)
    B, ldb = _alloc(K, N)
    Cm = torch.randn(M, N, device="cuda")
    C0 = Cm.clone()
    L.gemm(A, B, Cm, M, N, K, lda, ldb, N, layout=L.GEMM_TN, compute=compute, accumulate=True, splitk=splitk)
    ref = C0.double() + A[:, :M].double().t() @ B[:, :N].double()
    assert _relerr(Cm, ref) < TOL[compute]


@pytest.mark.parametrize("compute", [0, 1])
def test_batched_attention_shapes(compute):
    """S = scale * Q K^T, O = P V, dV = P^T dO with the (B, T, 192) qkv packing (ld = 192)."""
    L = _lib()
    torch.manual_seed(3)
    Bn, T, D = 3, 150, 64
    qkv = torch.randn(Bn, T, 192, device="cuda")
    Ts = (T + 3) // 4 * 4
    S = torch.zeros(Bn, T, Ts, device="cuda")
    q, k, v = qkv[..., :64], qkv[..., 64:128], qkv[..., 128:]
    L.gemm(q, k, S, T, T, D, 192, 192, Ts, layout=L.GEMM_NT, compute=compute, batch=Bn, sA=T * 192, sB=T * 192,
           sC=T * Ts, alpha=0.125)
    ref = 0.125 * q.double() @ k.double().transpose(1, 2)
    assert _relerr(S[..., :T], ref) < TOL[compute]
    P = torch.softmax(S[..., :T], -1)
    Pb = torch.zeros(Bn, T, Ts, device="cuda"); Pb[..., :T] = P
    O = torch.zeros(Bn, T, 64, device="cuda")
    L.gemm(Pb, v, O, T, D, T, Ts, 192, 64, layout=L.GEMM_NN, compute=compute, batch=Bn, sA=T * Ts, sB=T * 192, sC=T * 64)
    assert _relerr(O, P.double() @ v.double()) < TOL[compute]
    dO = torch.randn(Bn, T, 64, device="cuda")
    dqkv = torch.zeros(Bn, T, 192, device="cuda")
    dv = dqkv[..., 128:]
    L.gemm(Pb, dO, dv, T, D, T, Ts, 64, 192, layout=L.GEMM_TN, compute=compute, batch=Bn, sA=T * Ts, sB=T * 64, sC=T * 192)
    assert _relerr(dv, P.double().transpose(1, 2) @ dO.double()) < TOL[compute]
    assert dqkv[..., :128].abs().max().item() == 0.0


def _padded(Bn, T, Cc, lens=None, guard=True):
    """(B, T+2, C) padded token-major tensor inside a flat buffer with one guard row each side."""
    Tp = T + 2
    flat = torch.zeros((Bn * Tp + 2) * Cc, device="cuda")
    x = flat[Cc:Cc + Bn * Tp * Cc].view(Bn, Tp, Cc)
    x[:, 1:T + 1] = torch.randn(Bn, T, Cc, device="cuda")
    if lens is not None:
        for b, l in enumerate(lens):
            x[b, 1 + l:] = 0
    return flat, x


@pytest.mark.parametrize("compute", [0, 1])
def test_conv_k3_forward_backward(compute):
    """Conv1d(k=3, pad=1) over padded token-major activations == overlapping-row GEMMs; checked against
    F.conv1d forward, grad-input and grad-weight (reference op: transformer.py:67-71 CoreNet convs)."""
    L = _lib()
    torch.manual_seed(4)
    Bn, T, Cin, Cout = 3, 37, 64, 96
    Tp = T + 2
    lens = torch.tensor([37, 20, 5], dtype=torch.int32, device="cuda")
    xflat, x = _padded(Bn, T, Cin, lens.tolist())
    W = torch.randn(Cout, Cin, 3, device="cuda") * 0.2      # reference layout
    bias = torch.randn(Cout, device="cuda")
    Wt = W.permute(0, 2, 1).contiguous()                    # tap-major [Cout][3][Cin] (internal layout)
    rows = Bn * Tp
    # forward, PAD mask, relu
    hflat = torch.zeros((rows + 2) * Cout, device="cuda")
    h = hflat[Cout:Cout + rows * Cout].view(Bn, Tp, Cout)
    L.gemm(x, Wt, h, rows, Cout, 3 * Cin, Cin, 3 * Cin, Cout, layout=L.GEMM_NT, compute=compute, bias=bias, relu=True,
           mask_mode=L.MASK_PAD, lens=lens, Tp=Tp, a_offset=-Cin)
    xr = x[:, 1:T + 1].transpose(1, 2).double().requires_grad_(True)
    Wd = W.double().requires_grad_(True)
    hpre = F.conv1d(xr, Wd, bias.double(), padding=1)
    href = F.relu(hpre)
    assert _relerr(h[:, 1:T + 1], href.transpose(1, 2).detach()) < TOL[compute]
    assert h[:, 0].abs().max().item() == 0.0 and h[:, Tp - 1].abs().max().item() == 0.0
    # backward: dH given (zero on structural rows), relu gate by h
    dyflat, dy = _padded(Bn, T, Cout)
    gh = dy * (h > 0).float()
    # backward through the SAME relu mask as the kernel output (sign flips of borderline pre-activations between the bf16
    # and fp64 forward would otherwise dominate the comparison)
    hpre.backward((dy[:, 1:T + 1] * (h[:, 1:T + 1] > 0)).transpose(1, 2).double())
    # grad-input: NN with 3 tap segments, LEN mask
    ghflat = torch.zeros((rows + 2) * Cout, device="cuda")
    ghv = ghflat[Cout:Cout + rows * Cout].view(Bn, Tp, Cout); ghv.copy_(gh)
    dx = torch.zeros(Bn, Tp, Cin, device="cuda")
    L.gemm(ghv, Wt, dx, rows, Cin, 3 * Cout, Cout, 3 * Cin, Cin, layout=L.GEMM_NN, compute=compute, seglen=Cout,
           seg0=2 * Cin, segstride=-Cin, mask_mode=L.MASK_LEN, lens=lens, Tp=Tp, a_offset=-Cout)
    dxref = xr.grad.transpose(1, 2)
    for b, l in enumerate(lens.tolist()):
        assert _relerr(dx[b, 1:1 + l], dxref[b, :l]) < TOL[compute]
        assert dx[b, 1 + l:].abs().max().item() == 0.0
    # grad-weight: TN, split-K, accumulate into zeroed tap-major grad
    dWt = torch.zeros(Cout, 3 * Cin, device="cuda")
    L.gemm(ghv, x, dWt, Cout, 3 * Cin, rows, Cout, Cin, 3 * Cin, layout=L.GEMM_TN, compute=compute, accumulate=True, splitk=3,
           b_offset=-Cin)
    dWref = Wd.grad.permute(0, 2, 1).reshape(Cout, 3 * Cin)
    assert _relerr(dWt, dWref) < TOL[compute]


def test_epilogue_residual_gate_accumulate():
    L = _lib()
    torch.manual_seed(5)
    M, N, K = 140, 96, 64
    A, lda = _alloc(M, K)
    B, ldb = _alloc(N, K)
    R = torch.randn(M, N, device="cuda")
    G = torch.randn(M, N, device="cuda")
    Cm = torch.randn(M, N, device="cuda")
    C0 = Cm.clone()
    L.gemm(A, B, Cm, M, N, K, lda, ldb, N, layout=L.GEMM_NT, compute=0, R=R, ldr=N, G=G, ldg=N, accumulate=True, alpha=0.5)
    # epilogue order: alpha * acc -> gate -> + beta * R  (the residual path is not gated)
    ref = C0.double() + torch.where(G > 0, 0.5 * (A[:, :K].double() @ B[:, :K].double().t()), torch.zeros((), device="cuda", dtype=torch.float64)) + R.double()
    assert _relerr(Cm, ref) < TOL[0]


def test_bad_arguments_fail_loudly():
    L = _lib()
    A = torch.zeros(8, 6, device="cuda")
    with pytest.raises(L.XvaError):
        L.gemm(A, A, A, 8, 8, 6, 6, 6, 8)  # lda not a multiple of 4


def _conv1d_tm(x, W, bias, dilation, stride=1):
    """Reference conv over a time-major (B, T, C) tensor with 'same' zero padding (odd k)."""
    k = W.shape[2]
    pad = dilation * (k - 1) // 2
    y = F.conv1d(x.transpose(1, 2), W, bias, stride=stride, padding=pad, dilation=dilation)
    return y.transpose(1, 2)


@pytest.mark.parametrize("dtype,compute", [(torch.float32, 0), (torch.float32, 1), (torch.bfloat16, 1)])
@pytest.mark.parametrize("Cin,Cout,k,d", [(32, 32, 11, 5), (64, 64, 7, 3), (128, 256, 3, 1), (8, 16, 41, 1)])
def test_dilated_conv_segments_fwd_bwd(dtype, compute, Cin, Cout, k, d):
    """HiFi-GAN style Conv1d(k, dilation d) on time-major activations with PAD structural rows per item, fused
    LeakyReLU on the input (models.py:42-47), forward + backward-data + backward-weight, fp32 and bf16 storage."""
    L = _lib()
    torch.manual_seed(9)
    Bn, T, PAD = 2, 150, 32
    Tp = T + 2 * PAD
    rows = Bn * Tp
    slope = 0.1
    tol = TOL[compute] if dtype == torch.float32 else 3e-2
    x = torch.zeros(rows + 2 * PAD, Cin, device="cuda")            # guard rows front/back
    xv = x[PAD:PAD + rows].view(Bn, Tp, Cin)
    xv[:, PAD:PAD + T] = torch.randn(Bn, T, Cin, device="cuda")
    W = torch.randn(Cout, Cin, k, device="cuda") * (1.0 / (Cin * k) ** 0.5)
    bias = torch.randn(Cout, device="cuda")
    Wt = W.permute(0, 2, 1).contiguous()                           # tap-major [Cout][k][Cin]
    xs, Wts = x.to(dtype), Wt.to(dtype)
    y = torch.zeros(rows, Cout, device="cuda", dtype=dtype)
    half = (k - 1) // 2
    # forward: A(r, kk=(j,c)) = x[r + (j - half) * d][c]
    L.gemm(xs[PAD:], Wts, y, rows, Cout, k * Cin, Cin, k * Cin, Cout, layout=L.GEMM_NT, compute=compute, bias=bias,
           a_seglen=Cin, a_segadj=d * Cin - Cin, a_offset=-half * d * Cin, a_lrelu=slope, mask_mode=L.MASK_PAD, Tp=Tp, mask_pad=PAD)
    xr = xs[PAD:PAD + rows].view(Bn, Tp, Cin)[:, PAD:PAD + T].double().requires_grad_(True)
    Wd = Wts.view(Cout, k, Cin).permute(0, 2, 1).double().requires_grad_(True)
    yref = _conv1d_tm(F.leaky_relu(xr, slope), Wd, bias.double(), d)
    yv = y.view(Bn, Tp, Cout)
    assert _relerr(yv[:, PAD:PAD + T], yref.detach()) < tol
    assert yv[:, :PAD].abs().max().item() == 0 and yv[:, PAD + T:].abs().max().item() == 0
    # backward
    dy = torch.zeros(rows + 2 * PAD, Cout, device="cuda")
    dyv = dy[PAD:PAD + rows].view(Bn, Tp, Cout)
    dyv[:, PAD:PAD + T] = torch.randn(Bn, T, Cout, device="cuda")
    dys = dy.to(dtype)
    yref.backward(dys[PAD:PAD + rows].view(Bn, Tp, Cout)[:, PAD:PAD + T].double())
    # grad-input: dX[r] = sum_j dY[r - (j - half) d] W[:, j, :] -> A taps reversed, gate = lrelu'(x)
    if Cout % 32 == 0:
        dx = torch.zeros(rows, Cin, device="cuda", dtype=dtype)
        L.gemm(dys[PAD:], Wts, dx, rows, Cin, k * Cout, Cout, k * Cin, Cin, layout=L.GEMM_NN, compute=compute,
               a_seglen=Cout, a_segadj=d * Cout - Cout, a_offset=-half * d * Cout, seglen=Cout, seg0=(k - 1) * Cin, segstride=-Cin,
               G=xs[PAD:], ldg=Cin, gate_slope=slope, mask_mode=L.MASK_PAD, Tp=Tp, mask_pad=PAD)
        assert _relerr(dx.view(Bn, Tp, Cin)[:, PAD:PAD + T], xr.grad) < tol
    # grad-weight: dWt[co][j*Cin + c] = sum_r dY[r][co] lrelu(x)[r + (j - half) d][c]
    dWt = torch.zeros(Cout, k * Cin, device="cuda")
    L.gemm(dys[PAD:], xs[PAD:], dWt, Cout, k * Cin, rows, Cout, Cin, k * Cin, layout=L.GEMM_TN, compute=compute, accumulate=True,
           splitk=2, seglen=Cin, seg0=-half * d * Cin, segstride=d * Cin - Cin, b_lrelu=slope)
    dWref = Wd.grad.permute(0, 2, 1).reshape(Cout, k * Cin)
    assert _relerr(dWt, dWref) < tol


@pytest.mark.parametrize("dtype,compute", [(torch.float32, 0), (torch.bfloat16, 1)])
def test_strided_conv_and_two_level_batch(dtype, compute):
    """Stride-3 (k,1) Conv2d of the period discriminators (models.py:140-152) on a time-major wave folded to (T/p, p):
    width w is a second batch level, stride is lda = 3 * p * C."""
    L = _lib()
    torch.manual_seed(10)
    Bn, p_, Hin, Cin, Cout, k, s = 2, 3, 40, 32, 64, 5, 3
    PADH = 2
    Hp = Hin + 2 * PADH                                  # padded height per item, zero rows top/bottom
    Hout = (Hin + 2 * PADH - k) // s + 1
    x = torch.zeros(Bn, Hp, p_, Cin, device="cuda")
    x[:, PADH:PADH + Hin] = torch.randn(Bn, Hin, p_, Cin, device="cuda")
    W = torch.randn(Cout, Cin, k, device="cuda") * 0.1
    Wt = W.permute(0, 2, 1).contiguous().to(dtype)
    xs = x.to(dtype)
    y = torch.zeros(Bn, Hout, p_, Cout, device="cuda", dtype=dtype)
    # output row h' of item b, width w: A row = x[b, s*h' + j, w, :] ; batch = b (stride Hp*p*C), batch2 = w (stride C)
    L.gemm(xs, Wt, y, Hout, Cout, k * Cin, s * p_ * Cin, k * Cin, p_ * Cout, layout=L.GEMM_NT, compute=compute,
           a_seglen=Cin, a_segadj=p_ * Cin - Cin, batch=Bn, sA=Hp * p_ * Cin, sC=Hout * p_ * Cout, batch2=p_, sA2=Cin, sC2=Cout,
           act=L.ACT_LRELU, act_slope=0.1)
    xr = xs[:, PADH:PADH + Hin].double().permute(0, 3, 1, 2)      # (B, C, H, W)
    Wd = Wt.view(Cout, k, Cin).permute(0, 2, 1).double().unsqueeze(-1)
    yref = F.leaky_relu(F.conv2d(xr, Wd, None, stride=(s, 1), padding=(PADH, 0)), 0.1).permute(0, 2, 3, 1)
    assert yref.shape == y.shape
    assert _relerr(y, yref) < (TOL[compute] if dtype == torch.float32 else 3e-2)


@pytest.mark.parametrize("N", [1, 16, 32, 48, 64, 100])
@pytest.mark.parametrize("dtype,compute", [(torch.float32, 0), (torch.bfloat16, 1)])
def test_narrow_n_tiles(N, dtype, compute):
    L = _lib()
    torch.manual_seed(11)
    M, K = 300, 96
    A = torch.randn(M, K, device="cuda").to(dtype)
    ldb = (N + 7) // 8 * 8
    Bk = torch.randn(K, ldb, device="cuda").to(dtype)
    Bn_ = torch.randn(N, K, device="cuda").to(dtype)
    Cm = torch.zeros(M, N, device="cuda", dtype=dtype)
    tol = TOL[compute] if dtype == torch.float32 else 3e-2
    L.gemm(A, Bn_, Cm, M, N, K, K, K, N, layout=L.GEMM_NT, compute=compute)
    assert _relerr(Cm, A.double() @ Bn_.double().t()) < tol
    L.gemm(A, Bk, Cm, M, N, K, K, ldb, N, layout=L.GEMM_NN, compute=compute)
    assert _relerr(Cm, A.double() @ Bk[:, :N].double()) < tol
    C32 = torch.zeros(K, N, device="cuda")
    At = torch.randn(M, K, device="cuda").to(dtype)
    Bt = torch.randn(M, ldb, device="cuda").to(dtype)
    L.gemm(At, Bt, C32, K, N, M, K, ldb, N, layout=L.GEMM_TN, compute=compute, accumulate=True)
    assert _relerr(C32, At.double().t() @ Bt[:, :N].double()) < tol


def test_bf16_attention_shapes_and_tanh():
    L = _lib()
    torch.manual_seed(12)
    Bn, T, D = 2, 150, 64
    qkv = torch.randn(Bn, T, 192, device="cuda").bfloat16()
    Ts = 152
    S = torch.zeros(Bn, T, Ts, device="cuda", dtype=torch.bfloat16)
    q, k = qkv[..., :64], qkv[..., 64:128]
    L.gemm(q, k, S, T, T, D, 192, 192, Ts, layout=L.GEMM_NT, compute=1, batch=Bn, sA=T * 192, sB=T * 192, sC=T * Ts, alpha=0.125,
           act=L.ACT_TANH)
    ref = torch.tanh(0.125 * q.double() @ k.double().transpose(1, 2))
    assert _relerr(S[..., :T], ref) < 3e-2
