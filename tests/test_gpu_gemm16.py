"""gemm16 (forward GEMM / implicit-GEMM convolution on pre-split "s16" operands, LDS-DMA ring + ping-pong wave groups)
against torch fp64 references, through the C ABI.  Same bounds as the f16x3 form of actmi_op_gemm (test_gpu_kernels.py): the
per-product error of the split is 2^-22, the rest is fp32 accumulation order."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from actmi import ops  # noqa: E402

D = "cuda:0"


def rel_err(got, exp):
    got, exp = got.detach().cpu().double(), exp.detach().cpu().double()
    return float((got - exp).abs().max() / (exp.abs().max() + 1e-30))


def test_split16v2_round_trip_and_layout():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(512, 64, generator=g) * torch.logspace(-3, 3, 512).unsqueeze(1)
    s = ops.split16v2(x.to(D), 4.0)
    back = ops.unsplit16v2(s, 4.0).cpu()
    # 2^-22 relative (two 11-bit pieces); values whose lo piece is an fp16 subnormal keep an absolute floor of 2^-25 / scale
    assert bool(((back - x).abs() <= x.abs() * 2.0 ** -21 + 2.0 ** -25 / 4.0).all())
    halves = s.cpu().view(torch.float16).view(-1, 16).double()                  # per group of 8: hi0..7, lo0..7
    xs = (x.double() * 4.0).view(-1, 8)
    hi = xs.float().half()                                                       # rn16
    assert torch.equal(halves[:, :8], hi.double())
    assert torch.equal(halves[:, 8:], (xs - hi.double()).float().half().double())


@pytest.mark.parametrize("bm", [128, 256, 512])
@pytest.mark.parametrize("M,N,K", [(128, 128, 32), (1000, 136, 64), (1202, 1536, 512), (2404, 512, 3200), (300, 512, 4608),
                                   (77, 8, 96), (9616, 128, 128)])
def test_gemm16_shapes(M, N, K, bm):
    g = torch.Generator().manual_seed(M * 7 + N + K)
    A, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    exp = F.linear(A.double(), W.double(), b.double())
    sa, sw = 16.0, 64.0
    got = ops.gemm16(ops.split16v2(A.to(D), sa), ops.split16v2(W.to(D), sw), alpha=1.0 / (sa * sw), bias=b.to(D), out_fmt="f32",
                     bm=bm)
    assert rel_err(got, exp) < 1.5e-6 * max(1.0, (K / 512) ** 0.5)
    again = ops.gemm16(ops.split16v2(A.to(D), sa), ops.split16v2(W.to(D), sw), alpha=1.0 / (sa * sw), bias=b.to(D), out_fmt="f32",
                       bm=bm)
    assert torch.equal(got, again)


@pytest.mark.parametrize("bm", [128, 256, 512])
def test_gemm16_epilogue_forms(bm):
    """FrozenBN scale, bias, residual in both forms (s16 tensor; f32 table shared by the batch through row % res_mod), ReLU,
    s16 output with its own scale, row scatter, range flag."""
    M, N, K, R = 700, 256, 320, 50
    g = torch.Generator().manual_seed(3)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    b, sc = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5
    res, tab = torch.randn(M, N, generator=g), torch.randn(R, N, generator=g)
    A16, W16 = ops.split16v2(A.to(D), 16.0), ops.split16v2(W.to(D), 256.0)
    al = 1.0 / (16.0 * 256.0)
    base = F.linear(A.double(), W.double()) * sc.double() + b.double()
    flag = torch.zeros(1, dtype=torch.int32, device=D)
    # s16 residual + ReLU -> s16 output (scale 8)
    o = ops.gemm16(A16, W16, alpha=al, bias=b.to(D), scale=sc.to(D), res=ops.split16v2(res.to(D), 4.0), res_scale=0.25,
                   relu=True, out_scale=8.0, bm=bm, flag=flag)
    assert rel_err(ops.unsplit16v2(o, 8.0), F.relu(base + res.double())) < 2e-6
    assert int(flag.item()) == 0
    # f32 table residual by row modulo -> f32 output
    o = ops.gemm16(A16, W16, alpha=al, bias=b.to(D), scale=sc.to(D), res=tab.to(D), res_fmt="f32", res_mod=R, out_fmt="f32", bm=bm)
    assert rel_err(o, base + tab.double()[torch.arange(M) % R]) < 2e-6
    # row scatter into a larger output
    perm = torch.randperm(M + 40, generator=g)[:M].to(torch.int32)
    o = ops.gemm16(A16, W16, alpha=al, bias=b.to(D), out_fmt="f32", rowmap=perm.to(D), out_rows=M + 40, bm=bm)
    exp = torch.zeros(M + 40, N, dtype=torch.float64)
    exp[perm.long()] = F.linear(A.double(), W.double()) + b.double()
    assert rel_err(o, exp) < 2e-6
    # a value beyond the fp16 range of the split output raises the flag
    ops.gemm16(A16, W16, alpha=al, out_scale=2.0 ** 20, bm=bm, flag=flag)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("bm", [128, 256, 512])
@pytest.mark.parametrize("Cin,Cout,H,W,k,stride,pad", [(64, 128, 30, 40, 3, 2, 1), (128, 128, 15, 20, 3, 1, 1),
                                                       (64, 128, 30, 40, 1, 2, 0), (32, 64, 9, 7, 3, 1, 1)])
def test_gemm16_convolution(Cin, Cout, H, W, k, stride, pad, bm):
    """NHWC implicit im2col, per-camera weights (groups), FrozenBN affine + residual + ReLU; vs torch conv2d in fp64."""
    G, B = 2, 3
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(G, B, H, W, Cin, generator=g)
    w = torch.randn(G, Cout, k, k, Cin, generator=g) / (k * k * Cin) ** 0.5
    sc, bi = torch.rand(G, Cout, generator=g) + 0.5, torch.randn(G, Cout, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(G, B, Ho, Wo, Cout, generator=g)
    o = ops.gemm16(ops.split16v2(x.to(D), 16.0), ops.split16v2(w.to(D), 256.0), alpha=1.0 / 4096.0, scale=sc.to(D), bias=bi.to(D),
                   res=ops.split16v2(res.to(D), 16.0), res_scale=1.0 / 16.0, relu=True, out_scale=16.0, bm=bm,
                   conv=dict(stride=stride, pad=pad))
    got = ops.unsplit16v2(o, 16.0).cpu()
    for c in range(G):
        y = F.conv2d(x[c].permute(0, 3, 1, 2).double(), w[c].permute(0, 3, 1, 2).double(), stride=stride, padding=pad)
        y = y * sc[c].double().view(1, -1, 1, 1) + bi[c].double().view(1, -1, 1, 1)
        exp = F.relu(y.permute(0, 2, 3, 1) + res[c].double())
        assert rel_err(got[c], exp) < 2e-6


def test_gemm16_split_contraction_slices():
    M, N, K, S = 300, 512, 4608, 4
    g = torch.Generator().manual_seed(9)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    A16, W16 = ops.split16v2(A.to(D), 16.0), ops.split16v2(W.to(D), 256.0)
    parts = ops.gemm16(A16, W16, alpha=1.0 / 4096.0, splitk=S)
    assert tuple(parts.shape) == (S, M, N)
    assert rel_err(parts.sum(0), F.linear(A.double(), W.double())) < 3e-6
    with pytest.raises(RuntimeError, match="K tile"):
        ops.gemm16(ops.split16v2(torch.randn(64, 64).to(D)), ops.split16v2(torch.randn(64, 64).to(D)), splitk=4)


def test_gemm16_rejects_unsupported_shapes():
    a, w = ops.split16v2(torch.randn(64, 48).to(D)), ops.split16v2(torch.randn(64, 48).to(D))
    with pytest.raises(RuntimeError, match="multiple of 32"):
        ops.gemm16(a, w)
