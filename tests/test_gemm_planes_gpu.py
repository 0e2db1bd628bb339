"""Round 5: split-bf16 PLANES through the direct-to-LDS GEMM kernels (include/xva_gemm.h `planes`, `c_plane`): operands stored as hi = bf16(x) and
lo = bf16(x - hi), the product hi hi + hi lo + lo hi in ONE launch whose K loop runs three passes — the arithmetic of the fp32 mode's split products
(compute 2) with the split made once by the producer.  Checked against fp64 on the same planes for every layout, the staggered 256 x 256 / 384 x 128
tiles and the lock-step ones, split-K through slabs, the overlapping-row k = 3 convolution form FastPitch's feed-forward uses, and the plane OUTPUT."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _planes(x):
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    return torch.stack([hi, lo]).contiguous()            # [2, ...]: lo plane x.numel() elements after hi


def _ref(Ap, Bp, f):
    ah, al, bh, bl = Ap[0].double(), Ap[1].double(), Bp[0].double(), Bp[1].double()
    return f(ah, bh) + f(ah, bl) + f(al, bh)


@pytest.mark.parametrize("layout,M,N,K", [("NT", 1024, 1536, 1152), ("NT", 777, 384, 4608), ("NT", 300, 192, 384), ("NN", 1024, 1536, 1152), ("NN", 515, 384, 4608),
                                          ("TN", 1536, 1152, 4096), ("TN", 384, 1152, 5120), ("TN", 192, 384, 2048), ("TN", 384, 1152, 1726), ("TN", 1536, 1152, 35)])
def test_planes_product_equals_the_three_term_sum(layout, M, N, K):
    from xva_trainer_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    if layout == "NT":
        A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
        f = lambda a, b: a @ b.t()
        lda, ldb = K, K
    elif layout == "NN":
        A, B = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
        f = lambda a, b: a @ b
        lda, ldb = K, N
    else:
        A, B = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
        f = lambda a, b: a.t() @ b
        lda, ldb = M, N
    Ap, Bp = _planes(A).cuda(), _planes(B).cuda()
    ref = _ref(Ap.cpu(), Bp.cpu(), f)
    lay = {"NT": _lib.GEMM_NT, "NN": _lib.GEMM_NN, "TN": _lib.GEMM_TN}[layout]
    C = torch.zeros(M, N, device="cuda")
    kw = dict(layout=lay, compute=1, planes=1, a_plane=A.numel(), b_plane=B.numel())
    if layout == "TN":
        ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
        _lib.gemm(Ap, Bp, C, M, N, K, lda, ldb, N, accumulate=True, splitk=0, sk_ws=ws, **kw)        # the weight-gradient form: slabs + reduce
    else:
        _lib.gemm(Ap, Bp, C, M, N, K, lda, ldb, N, **kw)
    torch.cuda.synchronize()
    err = ((C.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 3e-6, err
    full = f(A.double(), B.double())                                             # against the unsplit operands: the split keeps ~16 mantissa bits
    assert ((C.double().cpu() - full).abs().max() / full.abs().max()).item() < 5e-5


def test_planes_conv3_form_with_relu_mask_and_plane_output():
    """conv1 of the feed-forward block (transformer.py:59-77) as FastPitch issues it: overlapping rows (K = 3 C_in from row r - 1), bias, ReLU, pad-row mask,
    the output written as a split-bf16 pair; then that pair as the A operand of the next product with an fp32 result + fp32 residual."""
    from xva_trainer_amd import _lib
    g = torch.Generator().manual_seed(7)
    Bn, T, Cin, Cmid = 3, 126, 384, 1536
    Tp = T + 2
    rows = Bn * Tp
    x = torch.zeros(rows + 2, Cin)
    x[1:rows + 1].view(Bn, Tp, Cin)[:, 1:T + 1] = torch.randn(Bn, T, Cin, generator=g)
    W1, b1 = torch.randn(Cmid, 3 * Cin, generator=g) * 0.05, torch.randn(Cmid, generator=g)
    W2, b2 = torch.randn(Cin, 3 * Cmid, generator=g) * 0.03, torch.randn(Cin, generator=g)
    xp, w1p, w2p = _planes(x).cuda(), _planes(W1).cuda(), _planes(W2).cuda()
    h = torch.zeros(2, rows + 2, Cmid, device="cuda", dtype=torch.bfloat16)
    hplane = (rows + 2) * Cmid
    _lib.gemm(xp, w1p, h, rows, Cmid, 3 * Cin, Cin, 3 * Cin, Cmid, layout=_lib.GEMM_NT, compute=1, planes=1, a_plane=x.numel(), b_plane=W1.numel(), c_plane=hplane,
              bias=b1.cuda(), relu=True, mask_mode=_lib.MASK_PAD, Tp=Tp, a_offset=0, c_offset=Cmid)
    torch.cuda.synchronize()
    xs = (xp[0].double() + xp[1].double()).cpu()
    xcat = torch.cat([xs[0:rows], xs[1:rows + 1], xs[2:rows + 2]], dim=1)
    w1 = (w1p[0].double() + w1p[1].double()).cpu()
    t = torch.arange(rows) % Tp
    live = ((t > 0) & (t < Tp - 1)).double()[:, None]
    href = torch.relu(xcat @ w1.t() + b1.double()) * live
    hgot = (h[0].double() + h[1].double()).cpu()[1:rows + 1]
    assert ((hgot - href).abs().max() / href.abs().max()).item() < 5e-5
    assert float(h[0][0].abs().max()) == 0 and float(h[:, rows + 1].abs().max()) == 0              # guard rows untouched
    # second product: planes in, fp32 out + fp32 residual
    R = torch.randn(rows, Cin, generator=g).cuda()
    y = torch.zeros(rows, Cin, device="cuda")
    _lib.gemm(h, w2p, y, rows, Cin, 3 * Cmid, Cmid, 3 * Cmid, Cin, layout=_lib.GEMM_NT, compute=1, planes=1, a_plane=hplane, b_plane=W2.numel(), bias=b2.cuda(), R=R, ldr=Cin)
    torch.cuda.synchronize()
    hs = (h[0].double() + h[1].double()).cpu()
    hcat = torch.cat([hs[0:rows], hs[1:rows + 1], hs[2:rows + 2]], dim=1)
    w2 = (w2p[0].double() + w2p[1].double()).cpu()
    yref = hcat @ w2.t() + b2.double() + R.double().cpu()
    assert ((y.double().cpu() - yref).abs().max() / yref.abs().max()).item() < 5e-5


def test_split_bf16_kernel_and_argument_checks():
    from xva_trainer_amd import _lib
    import ctypes as C
    x = torch.randn(4096 + 8, generator=torch.Generator().manual_seed(1)).cuda() * 100
    out = torch.zeros(2, x.numel(), device="cuda", dtype=torch.bfloat16)
    assert _lib.lib.xva_split_bf16(_lib.ptr(x), _lib.ptr(out), C.c_int64(x.numel()), C.c_int64(x.numel()), _lib.stream_ptr()) == 0
    torch.cuda.synchronize()
    want = _planes(x.cpu())
    assert torch.equal(out.cpu(), want)
    assert ((out[0].double() + out[1].double() - x.double()).abs().max() / x.abs().max()).item() < 2 ** -16
    A = torch.zeros(2, 64, 64, device="cuda", dtype=torch.bfloat16)
    Cm = torch.zeros(64, 64, device="cuda")
    with pytest.raises(Exception):                      # tap segments and planes do not combine
        _lib.gemm(A, A, Cm, 64, 64, 64, 64, 64, 64, compute=1, planes=1, a_plane=4096, b_plane=4096, a_seglen=32, a_segadj=64)
