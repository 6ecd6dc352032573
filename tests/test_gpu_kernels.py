"""Kernel-level parity of the HIP path (through the C ABI) against plain torch fp32 CPU references.
Tolerances: fp32 MFMA is an exact fp32 fma chain, so differences are summation-order only; bounds below are
relative to the result scale and an order of magnitude above what fp32 reassociation produces at these K.
The GEMM / implicit-conv tests run in both precisions: "f32" (native fp32 MFMA) and "f16x3" (operands split exactly
into two fp16 pieces, three fp16 MFMA products, fp32 accumulation: per-product error 2^-22, same bounds)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from actmi import ops  # noqa: E402


def dev():
    return torch.device("cuda:0")


def rel_err(got, exp):
    got, exp = got.detach().cpu().double(), exp.detach().cpu().double()
    return float((got - exp).abs().max() / (exp.abs().max() + 1e-30))


PRECS = ["f32", "f16x3"]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (37, 16, 64), (300, 512, 4608), (1202, 1536, 512), (129, 65, 36),
                                   (800, 3200, 512), (2404, 512, 3200), (64, 64, 4)])
def test_gemm_shapes(M, N, K, prec):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g)
    b = torch.randn(N, generator=g)
    exp = F.linear(A.double(), W.double(), b.double())
    got = ops.gemm(A.to(dev()), W.to(dev()), bias=b.to(dev()), prec=prec)
    # a K-long sequential fp32 fma chain: error grows ~sqrt(K) ulp of the running sum
    assert rel_err(got, exp) < 1.5e-6 * max(1.0, (K / 512) ** 0.5)
    if prec == "f16x3" and K % 4 == 0:
        # weights split ahead of time (what the engine does at finalize) give the same bits as splitting in the kernel
        got2 = ops.gemm(A.to(dev()), ops.split16(W.to(dev())), bias=b.to(dev()), prec=prec, w_split=True)
        assert torch.equal(got, got2)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("M,N,K,S,relu", [(100, 512, 3200, 8, False), (300, 192, 1024, 4, True), (77, 64, 2048, 3, "gelu")])
def test_gemm_sliced_splitk(M, N, K, S, relu, prec):
    """small-grid form used by the engine at B = 1-2: the contraction split into plain slices + a fixed-order combine
    pass that carries the epilogue (scale, bias, residual, activation); agrees with the one-pass product and with fp64"""
    g = torch.Generator().manual_seed(M + K)
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5
    b, sc, res = torch.randn(N, generator=g), torch.rand(N, generator=g) + 0.5, torch.randn(M, N, generator=g)
    d = dev()
    kw = dict(bias=b.to(d), scale=sc.to(d), res=res.to(d), relu=relu, prec=prec)
    one = ops.gemm(A.to(d), W.to(d), **kw)
    two = ops.gemm(A.to(d), W.to(d), splitk=S, **kw)
    again = ops.gemm(A.to(d), W.to(d), splitk=S, **kw)
    assert torch.equal(two, again)                                   # no atomics: bitwise repeatable
    exp = F.linear(A.double(), W.double()) * sc.double() + b.double() + res.double()
    exp = F.relu(exp) if relu is True else (F.gelu(exp) if relu == "gelu" else exp)
    assert rel_err(two, exp) < 2e-6 and rel_err(one, exp) < 2e-6
    assert (two.cpu() - one.cpu()).abs().max() < 1e-5


def test_gemm_sliced_splitk_rejects_empty_splits():
    A, W = torch.randn(64, 64).to(dev()), torch.randn(64, 64).to(dev())
    with pytest.raises(RuntimeError, match="K tile"):
        ops.gemm(A, W, splitk=4)                                     # 2 K tiles of 32 cannot feed 4 splits


def test_split16_is_an_exact_two_piece_split():
    """hi = rn16(x), lo = rn16(x - hi): hi + lo reproduces x to 2^-22 relative; tiny values survive as fp16
    subnormals (absolute error floor 2^-25); the split image keeps the fp32 matrix's addressing (4 hi halfs + 4 lo halfs
    in the 16 bytes of every aligned group of 4 floats)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1024, 64, generator=g) * torch.logspace(-6, 4, 1024).unsqueeze(1)
    x[0, :4] = torch.tensor([0.0, -0.0, 65503.0, -1.0])
    s = ops.split16(x.to(dev())).cpu()
    halves = s.view(torch.float16).view(-1, 8).float().double()          # per group: hi0..3, lo0..3
    hi, lo = halves[:, :4].reshape(x.shape), halves[:, 4:].reshape(x.shape)
    xd = x.double()
    assert torch.equal(hi, x.half().double())                            # round-to-nearest-even fp16
    assert torch.equal(lo, (xd - hi).float().half().double())
    err = (hi + lo - xd).abs()
    # relative 2^-22, with the absolute floor of the fp16 subnormal grid (2^-24 spacing -> 2^-25 rounding error)
    assert bool((err <= torch.maximum(xd.abs() * 2.0 ** -22, torch.tensor(2.0 ** -25, dtype=torch.float64))).all())


@pytest.mark.parametrize("a_scale", [1e-2, 1.0, 300.0])
def test_gemm_f16x3_magnitudes(a_scale):
    """the engine's operating range: activations from 1e-2 to a few hundred against weights around 2e-2 whose split
    image carries the 2^8 scale (lo pieces stay normal fp16 numbers) -> fp32-grade accuracy"""
    g = torch.Generator().manual_seed(11)
    M, N, K = 256, 192, 512
    A, W = torch.randn(M, K, generator=g) * a_scale, torch.randn(N, K, generator=g) * 0.02
    exp = F.linear(A.double(), W.double())
    got = ops.gemm(A.to(dev()), ops.split16(W.to(dev()), 256.0), prec="f16x3", w_split=256.0)
    assert rel_err(got, exp) < 2e-6


def test_gemm_f16x3_tiny_operands_degrade_to_an_absolute_floor():
    """both operands around 1e-4 (below the fp16 normal range, unscaled): each piece sits on the 2^-24 subnormal grid,
    so the element error is absolute (2^-25), i.e. ~2^-12 relative at this magnitude -- documented limit of the mode;
    callers with such operands pre-scale by a power of two (as the engine does for weights) or use prec="f32"."""
    g = torch.Generator().manual_seed(12)
    M, N, K = 128, 128, 512
    A, W = torch.randn(M, K, generator=g) * 1e-4, torch.randn(N, K, generator=g) * 1e-4
    exp = F.linear(A.double(), W.double())
    got = ops.gemm(A.to(dev()), W.to(dev()), prec="f16x3")
    assert rel_err(got, exp) < 2e-3
    got32 = ops.gemm(A.to(dev()), W.to(dev()), prec="f32")
    assert rel_err(got32, exp) < 2e-6


@pytest.mark.parametrize("mag", [1e-7, 1e-3, 1.0, 3e4])
def test_pow2_scale_and_scaled_operands(mag):
    """operands far outside the fp16 range (gradients in the backward pass) keep fp32-grade accuracy when the kernel is
    handed the device-computed power-of-two scale; the scale brings max|x| into [2^13, 2^14)"""
    g = torch.Generator().manual_seed(21)
    M, N, K = 200, 160, 256
    A, W = torch.randn(M, K, generator=g) * mag, torch.randn(N, K, generator=g) * 0.02
    d = dev()
    Ad, Wd = A.to(d), W.to(d)
    sc = ops.pow2_scale(Ad)
    s = float(sc[0])
    assert s == 2.0 ** round(np.log2(s)) and 2.0 ** 13 <= float(A.abs().max()) * s < 2.0 ** 14 and float(sc[1]) == 0.0
    exp = F.linear(A.double(), W.double())
    got = ops.gemm(Ad, Wd, prec="f16x3", a_scale_dev=sc, b_scale=256.0)
    assert rel_err(got, exp) < 2e-6
    # column-sliced view (row stride > width), all-zero operand
    assert float(ops.pow2_scale(Ad[:, :64])[0]) >= s
    assert float(ops.pow2_scale(torch.zeros(4, 8, device=d))[0]) == 1.0


@pytest.mark.parametrize("prec", PRECS)
def test_gemm_epilogue_features(prec):
    g = torch.Generator().manual_seed(3)
    M, N, K, mod = 250, 200, 128, 50
    A, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    add = torch.randn(mod, K, generator=g)
    scale, bias = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    res = torch.randn(7, N, generator=g)
    rowmap = torch.randperm(M + 11, generator=g)[:M].to(torch.int32)
    ncols = 128       # addend applies to column blocks starting below 128 (block tiles are 64 or 128 wide)
    rows = torch.arange(M)
    Aadd = A + add[rows % mod]
    full = torch.empty(M, N, dtype=torch.float64)
    # the kernel applies the addend per column BLOCK: emulate with the widest tile that divides ncols
    full[:, :ncols] = F.linear(Aadd.double(), W[:ncols].double())
    full[:, ncols:] = F.linear(A.double(), W[ncols:].double())
    full = torch.relu(full * scale.double() + bias.double() + res[rows % 7].double())
    exp = torch.zeros(M + 11, N, dtype=torch.float64)
    exp[rowmap.long()] = full
    d = dev()
    got = ops.gemm(A.to(d), W.to(d), bias=bias.to(d), scale=scale.to(d), res=res.to(d), res_mod=7, relu=True,
                   a_add=add.to(d), add_mod=mod, add_ncols=ncols, rowmap=rowmap.to(d), out_rows=M + 11, prec=prec)
    assert rel_err(got, exp) < 2e-6


@pytest.mark.parametrize("G,B,H,W,Cin,Cout,k,stride,pad", [
    (2, 2, 12, 16, 8, 8, 3, 1, 1), (3, 1, 15, 20, 64, 128, 3, 2, 1), (2, 2, 16, 24, 16, 32, 1, 2, 0),
    (1, 2, 30, 40, 256, 256, 3, 1, 1), (4, 1, 15, 20, 512, 512, 3, 1, 1)])
@pytest.mark.parametrize("prec", PRECS)
def test_conv_implicit_gemm(G, B, H, W, Cin, Cout, k, stride, pad, prec):
    g = torch.Generator().manual_seed(G * 100 + Cin)
    x = torch.randn(G, B, Cin, H, W, generator=g)
    w = torch.randn(G, Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5
    scale, bias = torch.rand(G, Cout, generator=g) + 0.5, torch.randn(G, Cout, generator=g)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    res = torch.randn(G, B, Cout, Ho, Wo, generator=g)
    exp = torch.stack([torch.relu(F.conv2d(x[i].double(), w[i].double(), None, stride, pad) *
                                  scale[i].double().view(1, -1, 1, 1) + bias[i].double().view(1, -1, 1, 1) + res[i].double())
                       for i in range(G)])
    d = dev()
    got = ops.conv2d_nhwc(x.permute(0, 1, 3, 4, 2).contiguous().to(d), w.permute(0, 1, 3, 4, 2).contiguous().to(d),
                          scale.to(d), bias.to(d), res.permute(0, 1, 3, 4, 2).contiguous().to(d), relu=True,
                          stride=stride, pad=pad, prec=prec)
    assert rel_err(got.permute(0, 1, 4, 2, 3), exp) < 2e-6 * max(1.0, (Cin * k * k / 512) ** 0.5)


@pytest.mark.parametrize("G,B,H,W,with_res", [(2, 2, 16, 32, True), (1, 3, 15, 20, False), (4, 1, 120, 160, True), (1, 1, 9, 70, True)])
def test_conv3x3_c64_direct(G, B, H, W, with_res):
    """the LDS-patch 3x3 convolution of layer1 (f16x3) against torch fp64, incl. ragged tiles (H % 8, W % 32 != 0)"""
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(G, B, 64, H, W, generator=g)
    w = torch.randn(G, 64, 64, 3, 3, generator=g) / 24.0
    scale, bias = torch.rand(G, 64, generator=g) + 0.5, torch.randn(G, 64, generator=g)
    res = torch.randn(G, B, 64, H, W, generator=g) if with_res else None
    exp = torch.stack([torch.relu(F.conv2d(x[i].double(), w[i].double(), None, 1, 1) * scale[i].double().view(1, -1, 1, 1)
                                  + bias[i].double().view(1, -1, 1, 1) + (res[i].double() if with_res else 0.0))
                       for i in range(G)])
    d = dev()
    got = ops.conv3x3_c64(x.permute(0, 1, 3, 4, 2).contiguous().to(d), w.permute(0, 1, 3, 4, 2).contiguous().to(d),
                          scale.to(d), bias.to(d), res.permute(0, 1, 3, 4, 2).contiguous().to(d) if with_res else None, relu=True)
    assert rel_err(got.permute(0, 1, 4, 2, 3), exp) < 2e-6
    # same bits as the implicit-GEMM path up to summation order: compare loosely against it too
    ref = ops.conv2d_nhwc(x.permute(0, 1, 3, 4, 2).contiguous().to(d), w.permute(0, 1, 3, 4, 2).contiguous().to(d),
                          scale.to(d), bias.to(d), res.permute(0, 1, 3, 4, 2).contiguous().to(d) if with_res else None,
                          relu=True, stride=1, pad=1, prec="f16x3")
    assert rel_err(got, ref) < 2e-6


@pytest.mark.parametrize("B,H,Nq,Nk,hd,masked,shared", [(2, 8, 1202, 1202, 64, False, False), (3, 8, 100, 1202, 64, False, True),
                                                       (4, 8, 102, 102, 64, True, False), (2, 4, 14, 14, 16, False, False),
                                                       (2, 4, 10, 12, 16, True, False), (1, 2, 33, 65, 32, True, False)])
@pytest.mark.parametrize("prec", PRECS)
def test_attention(B, H, Nq, Nk, hd, masked, shared, prec):
    g = torch.Generator().manual_seed(Nq + Nk)
    D = H * hd
    q = torch.randn((Nq, D) if shared else (B, Nq, D), generator=g)
    kv = torch.randn(B, Nk, 2 * D, generator=g)      # interleaved K|V rows exercise the row strides
    k, v = kv[..., :D], kv[..., D:]
    kpm = None
    if masked:
        kpm = torch.zeros(B, Nk, dtype=torch.bool)
        for b in range(B):
            kpm[b, Nk - 1 - 3 * b:] = True
    qq = (q.unsqueeze(0).expand(B, -1, -1) if shared else q).double().view(B, Nq, H, hd).transpose(1, 2)
    kk = k.double().reshape(B, Nk, H, hd).transpose(1, 2)
    vv = v.double().reshape(B, Nk, H, hd).transpose(1, 2)
    s = qq @ kk.transpose(-1, -2) / hd ** 0.5
    if masked:
        s = s.masked_fill(kpm.view(B, 1, 1, Nk), float("-inf"))
    exp = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(B, Nq, D)
    d = dev()
    kvd = kv.to(d)
    got, lse = ops.attention(q.to(d), kvd[..., :D], kvd[..., D:], H, kpm=kpm.to(torch.uint8).to(d) if masked else None,
                             q_shared=shared, want_lse=True, prec=prec)
    assert rel_err(got, exp) < 3e-6
    assert rel_err(lse, torch.logsumexp(s, -1)) < 3e-6


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,H,N,hd", [(3, 8, 33, 32), (2, 4, 4, 16), (2, 8, 130, 64)])
def test_attention_causal(B, H, N, hd, prec):
    """causal mask (attn_mask = triu(ones, 1) of latent_model.py:26): key j visible to query i only for j <= i"""
    g = torch.Generator().manual_seed(N)
    D = H * hd
    q, k, v = (torch.randn(B, N, D, generator=g) for _ in range(3))
    qq, kk, vv = (x.double().view(B, N, H, hd).transpose(1, 2) for x in (q, k, v))
    s = qq @ kk.transpose(-1, -2) / hd ** 0.5
    s = s.masked_fill(torch.triu(torch.ones(N, N, dtype=torch.bool), diagonal=1), float("-inf"))
    exp = (torch.softmax(s, -1) @ vv).transpose(1, 2).reshape(B, N, D)
    d = dev()
    got = ops.attention(q.to(d), k.to(d), v.to(d), H, causal=True, prec=prec)
    assert rel_err(got, exp) < 3e-6


@pytest.mark.parametrize("prec", PRECS)
def test_gemm_gelu_epilogue(prec):
    g = torch.Generator().manual_seed(8)
    A, W, b = torch.randn(100, 256, generator=g), torch.randn(1024, 256, generator=g) / 16, torch.randn(1024, generator=g)
    exp = F.gelu(F.linear(A.double(), W.double(), b.double()))
    got = ops.gemm(A.to(dev()), W.to(dev()), bias=b.to(dev()), relu="gelu", prec=prec)
    assert rel_err(got, exp) < 2e-6


@pytest.mark.parametrize("prec", PRECS)
def test_attention_online_softmax_rescale_branch(prec):
    """Force the running max to jump late: one key dominates in the LAST tile (rule: a rare branch needs its own test)."""
    B, H, Nq, Nk, hd = 1, 1, 40, 200, 64
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B, Nq, hd, generator=g)
    k = torch.randn(B, Nk, hd, generator=g) * 0.1
    v = torch.randn(B, Nk, hd, generator=g)
    k[0, 197] = q[0, 5] * 4.0          # spikes the score of query 5 (and some others) in the final tile
    s = (q.double() @ k.double().transpose(-1, -2)) / 8.0
    exp = torch.softmax(s, -1) @ v.double()
    d = dev()
    got = ops.attention(q.to(d), k.to(d), v.to(d), 1, prec=prec)
    assert rel_err(got, exp) < 3e-6


@pytest.mark.parametrize("M,D", [(5, 512), (1202 * 2, 512), (33, 64), (7, 2048)])
def test_layernorm(M, D):
    g = torch.Generator().manual_seed(M)
    x, r = torch.randn(M, D, generator=g) * 3, torch.randn(3, D, generator=g)
    w, b = torch.randn(D, generator=g), torch.randn(D, generator=g)
    w2, b2 = torch.randn(D, generator=g), torch.randn(D, generator=g)
    rows = torch.arange(M) % 3
    e1 = F.layer_norm((x + r[rows]).double(), (D,), w.double(), b.double(), 1e-5)
    e2 = F.layer_norm(e1, (D,), w2.double(), b2.double(), 1e-5)
    d = dev()
    g1 = ops.layernorm(x.to(d), w.to(d), b.to(d), res=r.to(d), res_mod=3)
    g2 = ops.layernorm(x.to(d), w.to(d), b.to(d), res=r.to(d), res_mod=3, w2=w2.to(d), b2=b2.to(d))
    assert rel_err(g1, e1) < 2e-6 and rel_err(g2, e2) < 4e-6


@pytest.mark.parametrize("H,W", [(23, 31), (24, 32), (22, 29), (240, 320)])
def test_maxpool_bit_exact(H, W):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(3, 64, H, W, generator=g)
    exp = F.max_pool2d(x, 3, 2, 1)
    got = ops.maxpool3x3s2(x.permute(0, 2, 3, 1).contiguous().to(dev()))
    assert torch.equal(got.permute(0, 3, 1, 2).cpu(), exp)      # max is order independent: bit exact


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("B,C,H,W,Cout", [(2, 2, 64, 96, 8), (1, 3, 96, 64, 8), (1, 1, 480, 640, 64), (2, 2, 70, 150, 64)])
def test_conv1_u8_and_f32(B, C, H, W, Cout, prec):
    g = torch.Generator().manual_seed(H)
    img = torch.randint(0, 256, (B, C, H, W, 3), dtype=torch.uint8, generator=g)
    w = torch.randn(C, Cout, 3, 7, 7, generator=g) / 147 ** 0.5
    scale, bias = torch.rand(C, Cout, generator=g) + 0.5, torch.randn(C, Cout, generator=g) * 0.1
    x = torch.from_numpy(np.moveaxis(img.numpy(), -1, -3) / 255.0).float()        # get_image contract
    mean = torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1)
    std = torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1)
    xn = ((x - mean) / std).double()
    exp = torch.stack([torch.relu(F.conv2d(xn[:, c], w[c].double(), None, 2, 3) * scale[c].double().view(1, -1, 1, 1)
                                  + bias[c].double().view(1, -1, 1, 1)) for c in range(C)])      # [C,B,Cout,Ho,Wo]
    d = dev()
    got_u8 = ops.conv1(img.to(d), w.to(d), scale.to(d), bias.to(d), prec=prec).permute(0, 1, 4, 2, 3)
    got_f32 = ops.conv1(x.to(d), w.to(d), scale.to(d), bias.to(d), prec=prec).permute(0, 1, 4, 2, 3)
    assert rel_err(got_u8, exp) < 2e-6
    assert rel_err(got_f32, exp) < 2e-6
    assert torch.equal(got_u8, got_f32)          # LUT path == arithmetic path, bit for bit


def test_ensemble_matches_reference_transcription():
    from oracle.act_ref import TemporalEnsembleRef
    E, T, Q, A = 5, 37, 10, 16
    rng = np.random.default_rng(0)
    chunks = rng.standard_normal((T, E, Q, A)).astype(np.float32)
    chunks[3, 1, 2, 7] = 0.0        # episode 1: row written at t=3 for step 5 is not "populated"
    chunks[20, 4, 0, :] = 0.0       # episode 4: the newest row itself is all zero at t=20
    refs = [TemporalEnsembleRef(T, Q, A) for _ in range(E)]
    ens = ops.TemporalEnsemble(E, Q, A, 0.01, dev())
    for t in range(T):
        out = ens.step(torch.from_numpy(chunks[t]).to(dev())).cpu()
        pop = ens.populated.cpu().numpy()
        for e in range(E):
            raw, popref = refs[e].step(t, torch.from_numpy(chunks[t, e:e + 1]))
            assert out.dtype == torch.float64
            assert np.allclose(out[e].numpy(), raw.numpy()[0], rtol=0, atol=1e-13)
            ref_rows = popref.numpy()[max(0, t - Q + 1):t + 1]
            mine = pop[e][Q - len(ref_rows):]
            assert np.array_equal(mine.astype(bool), ref_rows)       # populated mask: bit exact


def test_ensemble_with_no_populated_row_is_the_reference_empty_sum():
    """Edge case of imitate_episodes.py:405-410: when NO row for the current step passes `all(actions != 0)` the reference
    indexes an empty set, its weights are an empty array (0/0 never evaluated) and the sum over zero rows is zeros[1, A].
    The kernel (misc.hip: w = 0 for every row) must give exactly that, with an all-false populated mask, and carry on
    correctly on the following steps."""
    from oracle.act_ref import TemporalEnsembleRef
    E, T, Q, A = 3, 6, 4, 16
    rng = np.random.default_rng(5)
    chunks = rng.standard_normal((T, E, Q, A)).astype(np.float32)
    chunks[0, 0, 0, 3] = 0.0             # episode 0, t = 0: the only row for step 0 has a zero -> nothing populated
    chunks[0, 1, :, :] = 0.0             # episode 1: an all-zero first chunk -> nothing populated at t = 0
    chunks[1, 1, 0, 0] = 0.0             # ... and at t = 1 both candidate rows fail (old row all zero, new row has a zero)
    chunks[2, 2, 0, 5] = 0.0             # episode 2, t = 2: the newest row fails, the two older ones count
    refs = [TemporalEnsembleRef(T, Q, A) for _ in range(E)]
    ens = ops.TemporalEnsemble(E, Q, A, 0.01, dev())
    n_empty = 0
    for t in range(T):
        out = ens.step(torch.from_numpy(chunks[t]).to(dev())).cpu()
        pop = ens.populated.cpu().numpy()
        for e in range(E):
            raw, popref = refs[e].step(t, torch.from_numpy(chunks[t, e:e + 1]))
            ref_rows = popref.numpy()[max(0, t - Q + 1):t + 1]
            assert np.array_equal(pop[e][Q - len(ref_rows):].astype(bool), ref_rows)
            assert not pop[e][:Q - len(ref_rows)].any()                  # rows before the episode began never count
            if int(popref.sum()) == 0:
                n_empty += 1
                assert raw.shape == (1, A) and float(raw.abs().max()) == 0.0          # the reference's empty sum
                assert torch.equal(out[e], torch.zeros(A, dtype=torch.float64))      # bit-exact zeros, no NaN
            else:
                assert np.allclose(out[e].numpy(), raw.numpy()[0], rtol=0, atol=1e-13)
    assert n_empty == 3


@pytest.mark.parametrize("G,B,H,W", [(1, 1, 6, 32), (2, 3, 9, 37), (4, 2, 30, 40), (1, 2, 5, 70), (3, 1, 1, 1)])
def test_wgrad3x3_c64_matches_torch(G, B, H, W):
    """weight gradient of the layer1 convolutions on the direct kernel (strip walk, transposed staging, nine taps per staged
    row) vs torch.nn.grad.conv2d_weight in fp64: map widths below / across / above the 32-pixel strip, single-row and
    single-pixel maps, several images per group; with a device-side operand scale for tiny gradients; bitwise repeatable."""
    g = torch.Generator().manual_seed(G * 100 + H + W)
    x = torch.randn(G, B, H, W, 64, generator=g)
    dy = torch.randn(G, B, H, W, 64, generator=g)
    got = ops.wgrad3x3_c64(dy.to(dev()), x.to(dev())).cpu()
    sc = torch.tensor([2.0 ** 20], device=dev())              # gradients of 1e-6: split after the device-side scale
    got_s = ops.wgrad3x3_c64(dy.to(dev()) * 2.0 ** -20, x.to(dev()), dy_scale=sc).cpu()
    for c in range(G):
        exp = torch.nn.grad.conv2d_weight(x[c].permute(0, 3, 1, 2).double(), (64, 64, 3, 3), dy[c].permute(0, 3, 1, 2).double(),
                                          padding=1).permute(0, 2, 3, 1)          # [O][kh][kw][I]
        assert rel_err(got[c], exp) < 2e-6
    assert torch.equal(got_s, got * 2.0 ** -20)
    assert torch.equal(got, ops.wgrad3x3_c64(dy.to(dev()), x.to(dev())).cpu())


@pytest.mark.parametrize("G,B,H,W", [(1, 1, 12, 64), (2, 2, 30, 46), (4, 1, 64, 96), (1, 3, 9, 130), (2, 1, 2, 2)])
def test_wgrad7x7s2_matches_torch(G, B, H, W):
    """weight gradient of the stem (7x7 / s2 / p3 on the 4-channel-padded image) on the direct kernel (strip walk, image rows
    staged as transposed im2col slices, nine-slot ring) vs torch.nn.grad.conv2d_weight in fp64: widths below / across the
    32-pixel strip, odd sizes, a 1x1 output map; device-side operand scale; bitwise repeatable."""
    g = torch.Generator().manual_seed(G * 10 + H + W)
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    x = torch.randn(G, B, H, W, 4, generator=g)
    dy = torch.randn(G, B, Ho, Wo, 64, generator=g)
    got = ops.wgrad7x7s2(dy.to(dev()), x.to(dev())).cpu()
    sc = torch.tensor([2.0 ** 20], device=dev())
    got_s = ops.wgrad7x7s2(dy.to(dev()) * 2.0 ** -20, x.to(dev()), dy_scale=sc).cpu()
    for c in range(G):
        exp = torch.nn.grad.conv2d_weight(x[c].permute(0, 3, 1, 2).double(), (64, 4, 7, 7), dy[c].permute(0, 3, 1, 2).double(),
                                          stride=2, padding=3).permute(0, 2, 3, 1)          # [O][kh][kw][I]
        assert rel_err(got[c], exp) < 2e-6
    assert torch.equal(got_s, got * 2.0 ** -20)
    assert torch.equal(got, ops.wgrad7x7s2(dy.to(dev()), x.to(dev())).cpu())


@pytest.mark.parametrize("G,B,H,W,Cc,Cx,Cout,splitk", [(2, 2, 15, 20, 64, 32, 64, 0), (1, 3, 30, 40, 128, 64, 128, 0),
                                                       (2, 1, 8, 12, 256, 128, 256, 0), (1, 1, 15, 20, 512, 256, 512, 4),
                                                       (1, 2, 7, 9, 64, 32, 96, 2)])
def test_conv2_with_downsample_as_second_source(G, B, H, W, Cc, Cx, Cout, splitk):
    """relu(bn2(conv3x3(y1)) + bn_ds(conv1x1_s2(x))) of a torchvision BasicBlock (reference backbone.py:66-71) as ONE implicit
    GEMM whose contraction continues from the nine taps of y1 into the strided pixels of x, FrozenBN scales folded into the
    weights; against torch's conv2d in float64.  Odd map sizes (x is (2H-1) x (2W-1) or 2H x 2W), several tile shapes, and the
    sliced split-K form the engine uses at small batch."""
    g = torch.Generator().manual_seed(G * 100 + H)
    d = dev()
    Hx, Wx = 2 * H - (H % 2), 2 * W - (W % 2)                # both parities of the stride-2 input size
    y1 = torch.randn(G, B, Cc, H, W, generator=g)
    x = torch.randn(G, B, Cx, Hx, Wx, generator=g)
    w2 = torch.randn(G, Cout, Cc, 3, 3, generator=g) / (9 * Cc) ** 0.5
    wd = torch.randn(G, Cout, Cx, 1, 1, generator=g) / Cx ** 0.5
    s2, sd = torch.rand(G, Cout, generator=g) + 0.5, torch.rand(G, Cout, generator=g) * 2 + 0.1
    b2, bd = torch.randn(G, Cout, generator=g) * 0.1, torch.randn(G, Cout, generator=g) * 0.1
    exp = torch.stack([torch.relu(F.conv2d(y1[i].double(), w2[i].double(), None, 1, 1) * s2[i].double().view(1, -1, 1, 1)
                                  + b2[i].double().view(1, -1, 1, 1)
                                  + F.conv2d(x[i].double(), wd[i].double(), None, 2, 0) * sd[i].double().view(1, -1, 1, 1)
                                  + bd[i].double().view(1, -1, 1, 1)) for i in range(G)])             # [G,B,Cout,H,W]
    assert exp.shape[-2:] == (H, W)
    wf = torch.cat([(w2 * s2.view(G, Cout, 1, 1, 1)).permute(0, 1, 3, 4, 2).reshape(G, Cout, 9 * Cc),
                    (wd * sd.view(G, Cout, 1, 1, 1)).reshape(G, Cout, Cx)], dim=2).contiguous()
    scale = 256.0
    wf16 = ops.split16(wf.to(d), scale)
    y1n, xn = y1.permute(0, 1, 3, 4, 2).contiguous().to(d), x.permute(0, 1, 3, 4, 2).contiguous().to(d)
    got = ops.conv2d_with_second_source(y1n, xn, wf16, scale, (b2 + bd).to(d), splitk=splitk)
    assert rel_err(got.permute(0, 1, 4, 2, 3), exp) < 3e-6
    # the engine's K order (channel blocks outer, taps inner; the second source's columns stay behind them): same result
    wfp16 = ops.split16(ops.permute_conv_k(wf.to(d), 9, Cc), scale)
    got2 = ops.conv2d_with_second_source(y1n, xn, wfp16, scale, (b2 + bd).to(d), splitk=splitk, k_tap_inner=True)
    assert rel_err(got2.permute(0, 1, 4, 2, 3), exp) < 3e-6


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("G,B,H,W,Cin,Cout,stride", [(2, 2, 15, 20, 64, 64, 1), (1, 2, 30, 40, 128, 96, 2), (1, 1, 9, 11, 32, 256, 1)])
def test_conv_k_order_taps_inner(G, B, H, W, Cin, Cout, stride, prec):
    """actmi_gemm_desc.k_tap_inner: weight rows in (channel block, tap, channel) order give the convolution of the (tap, channel)
    rows (only the summation order inside the fp32 accumulation differs); the permutation itself is checked element by element."""
    g = torch.Generator().manual_seed(Cin + H)
    d = dev()
    x = torch.randn(G, B, H, W, Cin, generator=g).to(d)
    w = (torch.randn(G, Cout, 3, 3, Cin, generator=g) / (9 * Cin) ** 0.5).to(d)
    bias = torch.randn(G, Cout, generator=g).to(d)
    ref = ops.conv2d_nhwc(x, w, bias=bias, relu=True, stride=stride, pad=1, prec=prec, b_scale=256.0 if prec == "f16x3" else 0.0)
    wp = ops.permute_conv_k(w.reshape(G * Cout, 9 * Cin), 9, Cin)
    exp_perm = w.reshape(G * Cout, 9, Cin // 32, 32).permute(0, 2, 1, 3).reshape(G * Cout, 9 * Cin)
    assert torch.equal(wp, exp_perm)
    got = ops.conv2d_nhwc(x, wp.reshape(G, Cout, 3, 3, Cin), bias=bias, relu=True, stride=stride, pad=1, prec=prec,
                          b_scale=256.0 if prec == "f16x3" else 0.0, k_tap_inner=True)
    assert rel_err(got, ref) < 2e-6


@pytest.mark.parametrize("B,H,HD,Nq,Nk,pad,dscale", [(2, 8, 64, 300, 300, False, 1.0), (1, 4, 64, 130, 257, True, 1.0),
                                                      (2, 2, 32, 129, 64, False, 1.0), (1, 4, 16, 70, 200, True, 1.0),
                                                      (1, 8, 64, 1202, 1202, False, 3e-7)])
def test_attention_backward_without_materialised_scores(B, H, HD, Nq, Nk, pad, dscale):
    """csrc/attn_bwd.hip against autograd through softmax(q k^T / sqrt(hd) + mask) v in float64: dq, dk, dv within 2e-5 of each
    tensor's largest element (ragged tiles, key padding, all head widths, gradients at 3e-7 of unit scale with the device-side
    operand scale), two runs bit-identical, the amax word equal to the bits of the largest gradient written."""
    g = torch.Generator().manual_seed(Nq + Nk)
    D = H * HD
    qkv = [torch.randn(B, n, D, generator=g, dtype=torch.float64).requires_grad_(True) for n in (Nq, Nk, Nk)]
    dout = torch.randn(B, Nq, D, generator=g, dtype=torch.float64) * dscale
    kpm = None
    if pad:
        kpm = torch.zeros(B, Nk, dtype=torch.bool)
        kpm[:, Nk - 37:] = True
        kpm[0, 5] = True
    q, k, v = (t.reshape(B, -1, H, HD).transpose(1, 2) for t in qkv)
    sc = (q @ k.transpose(-1, -2)) / HD ** 0.5
    if kpm is not None:
        sc = sc.masked_fill(kpm[:, None, None, :], float("-inf"))
    out = (sc.softmax(-1) @ v).transpose(1, 2).reshape(B, Nq, D)
    out.backward(dout)
    dvc = [t.detach().float().to(dev()) for t in qkv]
    kd = kpm.to(torch.uint8).to(dev()) if kpm is not None else None
    o, lse = ops.attention(dvc[0], dvc[1], dvc[2], H, kpm=kd, want_lse=True, prec="f16x3")
    do = dout.float().to(dev())
    dsc = ops.pow2_scale(do.view(B * Nq, D)) if dscale != 1.0 else None
    dq, dk, dv, amax = ops.attention_bwd(dvc[0], dvc[1], dvc[2], o, lse, do, H, kpm=kd, do_scale=dsc, want_amax=True)
    worst = 0.0
    for got, ref, name in ((dq, qkv[0].grad, "dq"), (dk, qkv[1].grad, "dk"), (dv, qkv[2].grad, "dv")):
        err = float((got.cpu().double() - ref).abs().max() / ref.abs().max())
        worst = max(worst, err)
        assert err < 2e-5, (name, err)
    again = ops.attention_bwd(dvc[0], dvc[1], dvc[2], o, lse, do, H, kpm=kd, do_scale=dsc)
    assert all(torch.equal(a, b) for a, b in zip((dq, dk, dv), again))
    big = max(float(t.abs().max()) for t in (dq, dk, dv))
    assert int(amax.item()) == int(torch.tensor(big, dtype=torch.float32).view(torch.int32).item())
    print(f"attention backward B={B} H={H} hd={HD} {Nq}x{Nk}: worst error {worst:.2e} of the tensor maximum")


def test_attention_backward_dropout_masks_match_the_forward():
    """with the weight dropout on, forward and fused backward must use the same keep(seed, (row, key)) stream: central differences
    of sum(out * dout) along random directions of q, k, v"""
    g = torch.Generator().manual_seed(11)
    B, H, HD, N, p, seed = 1, 4, 64, 260, 0.2, 987
    D = H * HD
    q, k, v = (torch.randn(B, N, D, generator=g).to(dev()) for _ in range(3))
    dout = torch.randn(B, N, D, generator=g).to(dev())
    o, lse = ops.attention(q, k, v, H, want_lse=True, drop_p=p, drop_seed=seed, prec="f16x3")
    dq, dk, dv = ops.attention_bwd(q, k, v, o, lse, dout, H, drop_p=p, drop_seed=seed)
    for idx, (x, gx) in enumerate(((q, dq), (k, dk), (v, dv))):
        d = torch.randn(x.shape, generator=g).to(dev())
        eps = 2e-3
        args = [q, k, v]
        args[idx] = x + eps * d
        fp = (ops.attention(*args, H, drop_p=p, drop_seed=seed, prec="f16x3").double() * dout.double()).sum()
        args[idx] = x - eps * d
        fm = (ops.attention(*args, H, drop_p=p, drop_seed=seed, prec="f16x3").double() * dout.double()).sum()
        fd, an = float((fp - fm) / (2 * eps)), float((gx.double() * d.double()).sum())
        assert abs(fd - an) <= 5e-3 * max(1.0, abs(an)), (idx, fd, an)

