"""Training path parity on the GPU: losses and EVERY parameter gradient against the golden fixtures produced by the
reference's own modules (autograd of DETRVAE + ACTPolicy.__call__), and one AdamW step against torch.optim.AdamW
with the reference's two parameter groups.  Tolerance: gradients are sums of O(1e4..1e6) fp32 products whose
order differs from ATen's; bound = 2e-3 of the tensor's max |grad| (observed ~1e-5..1e-4), losses 1e-4 relative."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from helpers import load_fixture, regenerate, sample_like  # noqa: E402
from actmi import weights as W  # noqa: E402
from actmi import lib as L  # noqa: E402
from actmi import ops  # noqa: E402
from actmi.engine import ACTEngine  # noqa: E402


def rel_err(got, exp):
    got, exp = got.detach().cpu().double(), exp.detach().cpu().double()
    return float((got - exp).abs().max() / (exp.abs().max() + 1e-30))


def _gemm_desc(**kw):
    d = L.GemmDesc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


@pytest.mark.parametrize("M,N,K", [(77, 130, 96), (1202, 64, 1202), (16, 512, 800), (300, 36, 17)])
def test_gemm_transposed_operand_forms(M, N, K):
    """dX = dY W (B stored [K][N]) and dW = dY^T X (both stored [contraction][out])."""
    g = torch.Generator().manual_seed(M + N)
    d = torch.device("cuda:0")
    A = torch.randn(M, K, generator=g).to(d)
    Bkn = torch.randn(K, N, generator=g).to(d)
    out = torch.zeros(M, N, device=d)
    Kp = (K + 3) // 4 * 4
    Apad = torch.zeros(M, Kp, device=d)
    Apad[:, :K] = A
    desc = _gemm_desc(A=Apad, lda=Kp, M=M, N=N, K=K, Bw=Bkn, ldb=N, tb=1, C=out, ldc=N, groups=1)
    if N % 4 == 0:
        L.check(L.load().actmi_op_gemm(C.byref(desc), L.current_stream_ptr()), None, "gemm NT")
        assert rel_err(out, A.double() @ Bkn.double()) < 3e-6
    # both transposed: C[m][n] = sum_k At[k][m] Bt[k][n]
    Mp = (M + 3) // 4 * 4
    At = torch.zeros(K, Mp, device=d)
    At[:, :M] = A.t()
    Np = (N + 3) // 4 * 4
    Bt = torch.zeros(K, Np, device=d)
    Bt[:, :N] = Bkn
    out2 = torch.zeros(M, N, device=d)
    desc = _gemm_desc(A=At, lda=Mp, ta=1, M=M, N=N, K=K, Bw=Bt, ldb=Np, tb=1, C=out2, ldc=N, groups=1)
    L.check(L.load().actmi_op_gemm(C.byref(desc), L.current_stream_ptr()), None, "gemm TT")
    assert rel_err(out2, A.double() @ Bkn.double()) < 3e-6
    # split-K with atomics accumulates on top of existing values
    out3 = torch.ones(M, N, device=d)
    desc = _gemm_desc(A=At, lda=Mp, ta=1, M=M, N=N, K=K, Bw=Bt, ldb=Np, tb=1, C=out3, ldc=N, groups=1, splitk=3)
    L.check(L.load().actmi_op_gemm(C.byref(desc), L.current_stream_ptr()), None, "gemm TT splitk")
    assert rel_err(out3, A.double() @ Bkn.double() + 1.0) < 3e-6


@pytest.mark.parametrize("G,B,H,W,Cin,Cout,k,stride,pad", [(2, 2, 12, 16, 8, 16, 3, 1, 1), (1, 2, 15, 20, 16, 32, 3, 2, 1),
                                                          (2, 1, 16, 24, 16, 32, 1, 2, 0), (1, 1, 30, 40, 64, 64, 3, 1, 1)])
def test_conv_dgrad_wgrad(G, B, H, W, Cin, Cout, k, stride, pad):
    g = torch.Generator().manual_seed(H * 3 + Cin)
    d = torch.device("cuda:0")
    x = torch.randn(G, B, Cin, H, W, generator=g).double().requires_grad_(True)
    w = (torch.randn(G, Cout, Cin, k, k, generator=g) / (Cin * k * k) ** 0.5).double().requires_grad_(True)
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    dy = torch.randn(G, B, Cout, Ho, Wo, generator=g).double()
    for i in range(G):
        F.conv2d(x[i], w[i], None, stride, pad).backward(dy[i])
    dy_nhwc = dy.permute(0, 1, 3, 4, 2).contiguous().float().to(d)
    x_nhwc = x.detach().permute(0, 1, 3, 4, 2).contiguous().float().to(d)
    KK = k * k
    # dgrad weights [G][Cin][(r,s,n)]
    wd = w.detach().permute(0, 2, 3, 4, 1).contiguous().float().to(d)          # [G][Cin][k][k][Cout]
    dx = torch.zeros(G, B, H, W, Cin, device=d)
    desc = _gemm_desc(mode=2, A=dy_nhwc, H=H, W=W, Cin=Cin, KH=k, KW=k, stride=stride, pad=pad, Ho=Ho, Wo=Wo,
                      img_stride=Ho * Wo * Cout, M=B * H * W, N=Cin, K=KK * Cout, Bw=wd, ldb=KK * Cout, C=dx, ldc=Cin,
                      groups=G, gA=B * Ho * Wo * Cout, gB=Cin * KK * Cout, gC=B * H * W * Cin)
    L.check(L.load().actmi_op_gemm(C.byref(desc), L.current_stream_ptr()), None, "dgrad")
    assert rel_err(dx.permute(0, 1, 4, 2, 3), x.grad) < 3e-6
    # wgrad [G][Cout][(r,s,c)] with split-K atomics
    gw = torch.zeros(G, Cout, k, k, Cin, device=d)
    desc = _gemm_desc(A=dy_nhwc, lda=Cout, ta=1, M=Cout, K=B * Ho * Wo, Bw=x_nhwc, tb=2, N=KK * Cin, H=H, W=W, Cin=Cin,
                      KH=k, KW=k, stride=stride, pad=pad, Ho=Ho, Wo=Wo, img_stride=H * W * Cin, C=gw, ldc=KK * Cin,
                      groups=G, gA=B * Ho * Wo * Cout, gB=B * H * W * Cin, gC=Cout * KK * Cin, splitk=2)
    L.check(L.load().actmi_op_gemm(C.byref(desc), L.current_stream_ptr()), None, "wgrad")
    assert rel_err(gw.permute(0, 1, 4, 2, 3), w.grad) < 3e-6


def _run_training_fixture(name, check_all):
    z, cfg = load_fixture(name)
    sd_np, inp = regenerate(z, cfg)
    B = int(z["batch"])
    eng = ACTEngine(cfg, max_batch=B, training=True)
    eng.load_state_dict(sd_np)
    eng.finalize()
    d = eng.device
    if cfg.vq:        # VQ-ACT: replay the reference's multinomial draw of the latent code (detr_vae.py:140)
        out = eng.forward_train(torch.from_numpy(inp["qpos"]).to(d), torch.from_numpy(inp["image_u8"]).to(d),
                                torch.from_numpy(inp["actions"]).to(d), torch.from_numpy(inp["is_pad"]).to(d),
                                vq_code=torch.from_numpy(z["train.vq_code"]).view(B, cfg.vq_class, cfg.vq_dim).to(d))
    else:
        out = eng.forward_train(torch.from_numpy(inp["qpos"]).to(d), torch.from_numpy(inp["image_u8"]).to(d),
                                torch.from_numpy(inp["actions"]).to(d), torch.from_numpy(inp["is_pad"]).to(d),
                                eps=torch.from_numpy(z["train.eps"]).to(d))
    for k in ("l1", "kl", "loss"):
        got, exp = float(out[k]), float(z["train." + k][0])
        print(f"{name} {k}: hip {got:.6f} ref {exp:.6f}")
        assert abs(got - exp) <= 1e-4 * max(1.0, abs(exp)), k
    assert np.abs(out["a_hat"].cpu().numpy() - z["train.a_hat"]).max() <= 1e-4
    if cfg.vq:
        assert out["mu"] is None and out["logvar"] is None and float(out["kl"]) == 0.0
        assert np.abs(out["probs"].cpu().numpy().reshape(-1) - z["train.vq_probs"].reshape(-1)).max() <= 1e-5
        assert np.array_equal(out["binaries"].cpu().numpy().reshape(-1), z["train.vq_code"].reshape(-1))
        assert abs(float(out["vq_discrepancy"]) - float(z["train.vq_discrepancy"][0])) <= 1e-5
    else:
        assert np.abs(out["mu"].cpu().numpy() - z["train.mu"]).max() <= 1e-4
        assert np.abs(out["logvar"].cpu().numpy() - z["train.logvar"]).max() <= 1e-4
    eng.zero_grad()
    eng.backward(1.0)
    names = [str(n) for n in z["grad_names"]]
    none = set(str(n) for n in z["grad_none"])
    worst = (0.0, "")
    for n, ref_l2 in zip(names, z["grad_l2"]):
        g = eng.grad(n).cpu().double()
        if n in none:
            assert float(g.abs().max()) == 0.0, n            # is_pad_head: no gradient (quirk 3)
            continue
        got = float(g.norm())
        if ref_l2 == 0.0:
            assert got == 0.0, f"{n}: dead-layer gradient must be exactly zero (quirk 1)"
            continue
        if ref_l2 < 1e-6:
            # mathematically zero (decoder layer-0 self-attention q/k/v weights on tgt = 0, quirk 2): the reference
            # shows fp noise of ~1e-8 there, this implementation is exactly zero
            assert got < 1e-6, (n, got, ref_l2)
            continue
        e = abs(got - ref_l2) / ref_l2
        worst = max(worst, (e, n))
        assert e <= 2e-3, (n, got, ref_l2)
        key = "grad." + n
        if key in z.files:
            exp = z[key].reshape(-1)
            gs = sample_like(g.float().numpy(), z)
            assert np.abs(gs - exp).max() <= 2e-3 * max(np.abs(exp).max(), 1e-6), n
    print(f"{name}: worst relative gradient-norm error {worst[0]:.2e} at {worst[1]}")
    return eng, z, cfg, sd_np, inp


@pytest.mark.parametrize("name", ["tiny", "tiny_c3", "tiny_vq"])
def test_training_step_matches_reference_gradients(name):
    _run_training_fixture(name, True)


def test_training_full_size_gradients_and_adamw():
    """C=4, 480x640, B=2 (golden full4): losses, sampled gradients, gradient norms of all 640 parameters, then one
    AdamW step against torch.optim.AdamW fed with the same gradients."""
    eng, z, cfg, sd_np, inp = _run_training_fixture("full4", False)
    keys = ["action_head.weight", "transformer.encoder.layers.1.linear1.weight", "transformer.decoder.layers.3.linear1.weight",
            "backbones.2.0.body.layer3.0.conv1.weight", "backbones.0.0.body.conv1.weight", "is_pad_head.weight",
            "encoder.layers.0.norm1.bias", "query_embed.weight"]
    lr, lr_bb, wd = 1e-5, 1e-5 * 3, 1e-4
    before = {k: torch.from_numpy(sd_np[k]).clone() for k in keys}
    grads = {k: eng.grad(k).cpu() for k in keys}
    eng.adamw_step(lr, lr_bb, wd, step=1)
    after = eng.state_dict()
    for k in keys:
        p = before[k].clone().requires_grad_(True)
        if k.startswith("is_pad_head"):
            assert torch.equal(after[k], before[k])               # grad None in the reference => untouched
            continue
        opt = torch.optim.AdamW([p], lr=lr_bb if "backbone" in k else lr, weight_decay=wd)
        p.grad = grads[k].clone()
        opt.step()
        err = float((after[k] - p.detach()).abs().max())
        assert err <= 1e-7 + 1e-6 * float(p.detach().abs().max()), (k, err)
    # dead decoder layer: zero gradient -> only the decoupled weight decay moved it
    k = "transformer.decoder.layers.3.linear1.weight"
    assert torch.allclose(after[k], before[k] * (1 - lr * wd), rtol=0, atol=1e-9)
    # the inference path sees the updated weights (derived tensors were refreshed)
    d = eng.device
    a1 = eng.forward_infer(torch.from_numpy(inp["qpos"]).to(d), torch.from_numpy(inp["image_u8"]).to(d))
    assert torch.isfinite(a1).all()


def test_dropout_epilogue_statistics_and_determinism():
    g = torch.Generator().manual_seed(0)
    d = torch.device("cuda:0")
    M, N, K, p = 512, 384, 64, 0.3
    A, Wt, b = torch.randn(M, K, generator=g).to(d), torch.randn(N, K, generator=g).to(d), torch.randn(N, generator=g).to(d)
    res = torch.randn(M, N, generator=g).to(d)
    base = ops.gemm(A, Wt, bias=b)
    y1 = ops.gemm(A, Wt, bias=b, res=res, drop_p=p, drop_seed=7)
    y2 = ops.gemm(A, Wt, bias=b, res=res, drop_p=p, drop_seed=7)
    y3 = ops.gemm(A, Wt, bias=b, res=res, drop_p=p, drop_seed=8)
    assert torch.equal(y1, y2) and not torch.equal(y1, y3)                  # pure function of (seed, element)
    z = y1 - res                                                             # dropout acts before the residual add
    dropped = (z == 0)
    frac = float(dropped.float().mean())
    assert abs(frac - p) < 0.01
    kept = ~dropped
    assert torch.allclose(z[kept], base[kept] / (1 - p), rtol=1e-5, atol=1e-5)
    # attention-weight dropout: E[out] == no-dropout output (unbiased), rows differ by seed
    B, H, Nq, Nk, hd = 2, 4, 64, 96, 16
    q, k, v = (torch.randn(B, n, H * hd, generator=g).to(d) for n in (Nq, Nk, Nk))
    o0 = ops.attention(q, k, v, H)
    acc = torch.zeros_like(o0)
    for s in range(64):
        acc += ops.attention(q, k, v, H, drop_p=0.2, drop_seed=100 + s)
    assert float(((acc / 64) - o0).abs().mean()) < 0.05 * float(o0.abs().mean()) + 0.02


def test_training_gradients_with_dropout_by_finite_differences():
    """With a fixed dropout seed the loss is a deterministic function of the weights: check the analytic backward
    (including the general decoder self-attention path that dropout activates) against central differences."""
    from actmi.config import tiny_config
    cfg = tiny_config(kl_weight=1)
    sd_np = W.generate_state_dict(cfg, seed=21)
    B, p, seed = 2, 0.25, 1234567
    eng = ACTEngine(cfg, max_batch=B, training=True)
    inp = W.generate_inputs(cfg, B, seed=9, with_actions=True)
    d = eng.device
    t = {k: torch.from_numpy(v).to(d) for k, v in inp.items()}

    def loss_of(sd):
        eng.load_state_dict(sd)
        eng.finalize()
        out = eng.forward_train(t["qpos"], t["image_u8"], t["actions"], t["is_pad"], eps=t["eps"], dropout_p=p, dropout_seed=seed)
        return float(out["loss"].double()), out

    l0, _ = loss_of(sd_np)
    l0b, _ = loss_of(sd_np)
    assert l0 == l0b                                                       # same seed, same loss
    lp, _ = loss_of(sd_np)
    eng.zero_grad()
    eng.backward(1.0)
    keys = ["transformer.decoder.layers.0.self_attn.in_proj_weight", "transformer.decoder.layers.0.self_attn.out_proj.weight",
            "transformer.decoder.layers.0.self_attn.in_proj_bias", "query_embed.weight",
            "transformer.decoder.layers.0.multihead_attn.in_proj_weight", "transformer.decoder.layers.0.linear2.weight",
            "transformer.encoder.layers.1.linear1.weight", "transformer.encoder.layers.0.self_attn.out_proj.weight",
            "encoder.layers.0.self_attn.in_proj_weight", "latent_proj.weight", "action_head.weight",
            "backbones.1.0.body.layer3.0.conv1.weight", "input_proj.weight", "transformer.decoder.layers.0.norm1.weight"]
    grads = {k: eng.grad(k).cpu().numpy().copy() for k in keys}
    # the dropped self-attention makes the q/k gradients non-zero (they are exactly zero without dropout)
    assert np.abs(grads["transformer.decoder.layers.0.self_attn.in_proj_weight"][:2 * cfg.hidden_dim]).max() > 0
    worst = 0.0
    for k in keys:
        g = grads[k]
        idx = np.unravel_index(np.argmax(np.abs(g)), g.shape)            # the most sensitive element: best signal/noise
        ga = float(g[idx])
        h = 2e-2 if abs(ga) < 0.5 else 5e-3
        sp, sm = {kk: vv.copy() for kk, vv in sd_np.items()}, {kk: vv.copy() for kk, vv in sd_np.items()}
        sp[k][idx] += h
        sm[k][idx] -= h
        fd = (loss_of(sp)[0] - loss_of(sm)[0]) / (float(sp[k][idx]) - float(sm[k][idx]))
        err = abs(fd - ga) / max(abs(ga), 1e-3)
        print(f"{k}{list(idx)}: analytic {ga:+.5f} finite-diff {fd:+.5f} rel.err {err:.3f}")
        worst = max(worst, err)
        assert err < 0.08, (k, ga, fd)


def test_vq_device_drawn_code():
    """without a given code the library draws it itself (inverse CDF on the counter-based generator): one-hot per class,
    a pure function of the seed, and distributed like the softmax probabilities (detr_vae.py:139-140)"""
    z, cfg = load_fixture("tiny_vq")
    sd_np, inp = regenerate(z, cfg)
    B = int(z["batch"])
    eng = ACTEngine(cfg, max_batch=B, training=True)
    eng.load_state_dict(sd_np)
    eng.finalize()
    d = eng.device
    args = (torch.from_numpy(inp["qpos"]).to(d), torch.from_numpy(inp["image_u8"]).to(d),
            torch.from_numpy(inp["actions"]).to(d), torch.from_numpy(inp["is_pad"]).to(d))
    a = eng.forward_train(*args, dropout_seed=7)
    b = eng.forward_train(*args, dropout_seed=7)
    assert torch.equal(a["binaries"], b["binaries"]) and torch.equal(a["a_hat"], b["a_hat"])
    assert torch.equal(a["binaries"].sum(-1), torch.ones(B, cfg.vq_class, device=d))
    assert set(a["binaries"].unique().tolist()) <= {0.0, 1.0}
    probs = a["probs"].cpu().double()
    n, counts = 400, torch.zeros(B, cfg.vq_class, cfg.vq_dim, dtype=torch.float64)
    for s in range(n):
        counts += eng.forward_train(*args, dropout_seed=1000 + s)["binaries"].cpu().double()
    freq = counts / n
    assert float((freq - probs).abs().max()) < 4.5 * float((probs * (1 - probs) / n).sqrt().max()) + 1e-3
    eng.zero_grad()
    eng.backward(1.0)                                   # the straight-through path runs with a drawn code too
    assert float(eng.grad("latent_proj.weight").abs().max()) > 0


def test_chunks_longer_than_the_episode_are_padded_not_read_out_of_bounds():
    """actions / is_pad with fewer than num_queries steps (episodes shorter than the chunk): the wrapper continues the
    dataset's zero / is_pad=True padding; the result equals the explicitly padded call (the library used to read past the
    end of the short tensors: found by the default-on loss guard)."""
    from actmi.config import tiny_config
    cfg = tiny_config()
    eng = ACTEngine(cfg, max_batch=2, training=True)
    eng.load_state_dict(W.generate_state_dict(cfg, seed=0))
    inp = W.generate_inputs(cfg, 2, seed=3, with_actions=True)
    t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}
    Q, short = cfg.num_queries, 3
    a_full, p_full = t["actions"].clone(), t["is_pad"].clone()
    a_full[:, short:] = 0
    p_full[:, short:] = True
    junk = torch.full((1 << 22,), float("nan"), device="cuda")               # poison what a stray read would hit
    ref = eng.forward_train(t["qpos"], t["image_u8"], a_full, p_full, eps=t["eps"])
    got = eng.forward_train(t["qpos"], t["image_u8"], a_full[:, :short].contiguous(), p_full[:, :short].contiguous(), eps=t["eps"])
    del junk
    for k in ("l1", "kl", "loss"):
        assert torch.isfinite(got[k]) and float(got[k]) == float(ref[k])
    with pytest.raises(ValueError):
        eng.forward_train(t["qpos"], t["image_u8"], a_full[:, :, :5].contiguous(), p_full, eps=t["eps"])


@pytest.mark.parametrize("full", [False, True])
def test_gradients_are_bitwise_repeatable(full):
    """No float atomics in the backward pass: split weight-gradient contractions store plain slices that a second kernel
    adds in split order, LayerNorm / bias-gradient partials are summed in block order, the loss in block order -- two runs of
    the same step give identical bits in every gradient (the atomic split-K was run-to-run different)."""
    from actmi.config import ACTConfig, tiny_config
    cfg = ACTConfig(camera_names=["a", "b"]) if full else tiny_config()
    B = 3
    eng = ACTEngine(cfg, max_batch=B, training=True)
    eng.load_state_dict(W.generate_state_dict(cfg, seed=0))
    eng.finalize()
    inp = W.generate_inputs(cfg, B, seed=3, with_actions=True)
    t = {k: torch.from_numpy(v).cuda() for k, v in inp.items()}

    def run():
        eng.zero_grad()
        out = eng.forward_train(t["qpos"], t["image_u8"], t["actions"], t["is_pad"], eps=t["eps"], dropout_p=0.1, dropout_seed=11)
        eng.backward(1.0)
        torch.cuda.synchronize()
        return float(out["loss"]), eng.grad_arena().clone()

    l0, g0 = run()
    l1, g1 = run()
    assert l0 == l1
    assert torch.isfinite(g0).all() and float(g0.abs().max()) > 0
    assert torch.equal(g0, g1), f"{int((g0 != g1).sum())} gradient elements differ between two runs of the same step"


@pytest.mark.parametrize("name", ["tiny", "full4"])
def test_bf16_training_mode_is_opt_in_and_close_to_the_reference(name):
    """BASELINE config 3 says bf16: ACTEngine(train_prec="bf16") forms ONE bf16 product per fp32 product in every GEMM of the
    training step (fp32 accumulation, fp32 master weights and optimizer state).  It is an opt-in speed mode: the default step
    stays fp32-grade (f16x3), and this test pins how far bf16 moves the reference-run losses and gradients -- well inside what
    bf16 training tolerates, far outside the 1e-4 inference bar (which is why inference never uses it)."""
    z, cfg = load_fixture(name)
    sd_np, inp = regenerate(z, cfg)
    B = int(z["batch"])
    eng = ACTEngine(cfg, max_batch=B, training=True, train_prec="bf16")
    eng.load_state_dict(sd_np)
    eng.finalize()
    d = eng.device
    args = [torch.from_numpy(inp[k]).to(d) for k in ("qpos", "image_u8", "actions", "is_pad")]
    out = eng.forward_train(*args, eps=torch.from_numpy(z["train.eps"]).to(d))
    for k in ("l1", "kl", "loss"):
        got, exp = float(out[k]), float(z["train." + k][0])
        print(f"bf16 {name} {k}: hip {got:.6f} ref {exp:.6f} rel {abs(got - exp) / max(abs(exp), 1e-9):.2e}")
        assert abs(got - exp) <= 3e-2 * max(1.0, abs(exp)), k
    a_err = float(np.abs(out["a_hat"].cpu().numpy() - z["train.a_hat"]).max())
    assert a_err > 1e-4, "bf16 products cannot meet the fp32 bar: if this passes at 1e-4 the mode is not active"
    assert a_err <= 0.15 * max(1.0, float(np.abs(z["train.a_hat"]).max()))
    eng.zero_grad()
    eng.backward(1.0)
    names = [str(n) for n in z["grad_names"]]
    none = set(str(n) for n in z["grad_none"])
    worst, n_checked = (0.0, ""), 0
    big = float(max(z["grad_l2"]))
    for n, ref_l2 in zip(names, z["grad_l2"]):
        if n in none or ref_l2 < 1e-6:
            continue
        got = float(eng.grad(n).double().norm())
        # bf16 rounding noise adds in quadrature: a tensor whose gradient is small next to the network's largest shows it as an
        # inflated norm (measured: 0.09 against 0.057 at a layer1 convolution of the tiny fixture), hence the absolute term
        e = abs(got - ref_l2) / (ref_l2 + 0.02 * big)
        worst = max(worst, (e, n))
        n_checked += 1
        assert e <= 0.35, (n, got, ref_l2, big)
    print(f"bf16 {name}: a_hat max err {a_err:.2e}; worst relative gradient-norm error {worst[0]:.2e} at {worst[1]} over {n_checked} tensors")
    eng.check_flags()
    # the inference path of the same handle is untouched by the training mode: still fp32-grade
    a = eng.forward_infer(args[0], args[1]).cpu().numpy()
    assert np.abs(a - z["infer.a_hat"]).max() <= 1e-4
