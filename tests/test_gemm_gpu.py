"""GPU parity of the MFMA GEMM (C ABI xva_gemm) against fp64 torch matmul / conv1d.

Covers all three layouts, both compute modes, ragged sizes, batching, the overlapping-row
k=3 convolution form (forward, backward-data with tap segments, backward-weight with
split-K) and every epilogue option.  Inputs are asymmetric random (transpose-detecting).
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {0: 2e-5, 1: 2e-2}  # relative to max |ref|: fp32-exact MFMA vs bf16-input MFMA


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
    Bn, T, Cin, Cout = 3, 37, 48, 72
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
    href = F.relu(F.conv1d(xr, Wd, bias.double(), padding=1))
    assert _relerr(h[:, 1:T + 1], href.transpose(1, 2).detach()) < TOL[compute]
    assert h[:, 0].abs().max().item() == 0.0 and h[:, Tp - 1].abs().max().item() == 0.0
    # backward: dH given (zero on structural rows), relu gate by h
    dyflat, dy = _padded(Bn, T, Cout)
    gh = dy * (h > 0).float()
    href.backward((dy[:, 1:T + 1] * (h[:, 1:T + 1] > 0)).transpose(1, 2).double())
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
    import ctypes as C
    p = L.GemmParams()
    p.A = ghv.data_ptr(); p.B = x.data_ptr() - 4 * Cin; p.C = dWt.data_ptr()
    p.M, p.N, p.K = Cout, 3 * Cin, rows
    p.lda, p.ldb, p.ldc = Cout, Cin, 3 * Cin
    p.batch = 1; p.alpha = 1.0; p.accumulate = 1; p.splitk = 3; p.compute = compute; p.layout = L.GEMM_TN
    L.check(L.lib.xva_gemm(C.byref(p), L.stream_ptr()), "xva_gemm dW")
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
    ref = C0.double() + torch.where(G > 0, 0.5 * (A[:, :K].double() @ B[:, :K].double().t()) + R.double(), torch.zeros((), device="cuda", dtype=torch.float64))
    assert _relerr(Cm, ref) < 2e-5


def test_bad_arguments_fail_loudly():
    L = _lib()
    A = torch.zeros(8, 6, device="cuda")
    with pytest.raises(L.XvaError):
        L.gemm(A, A, A, 8, 8, 6, 6, 6, 8)  # lda not a multiple of 4
