"""Training step of the VQ-ACT latent prior (reference detr/models/latent_model.py:8-56 driven by
train_latent_model.py:323-343, 395-404) on the library's kernels.  Kernel-level checks against torch's own autograd on the
CPU (torch IS present: these pieces are pinned by torch), the whole step against the committed fixture produced by the
reference's module (tests/golden/latent_prior_train.npz, tools/gen_golden.py) and, at the default size, against the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from actmi import ops  # noqa: E402
from actmi import weights as W  # noqa: E402
from actmi.latent_model import LatentModelTransformer, latent_model_spec  # noqa: E402

D = "cuda:0"


def rel(got, exp):
    got, exp = got.detach().cpu().double(), exp.detach().cpu().double()
    return float((got - exp).abs().max() / (exp.abs().max() + 1e-30))


def test_gelu_dropout_and_transposed_products():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3000, generator=g, dtype=torch.float64, requires_grad=True)
    dy = torch.randn(3000, generator=g, dtype=torch.float64)
    y = F.gelu(x)
    y.backward(dy)
    assert rel(ops.gelu(x.detach().float().to(D)), y) < 1e-6
    assert rel(ops.gelu_bwd(x.detach().float().to(D), dy.float().to(D)), x.grad) < 2e-6
    # dropout: the same mask on a second call (that is the backward), kept values scaled, the kept fraction near 1 - p
    v = torch.randn(200000, generator=g).to(D)
    a, b = ops.dropout(v, 0.1, 77), ops.dropout(v, 0.1, 77)
    assert torch.equal(a, b) and not torch.equal(a, ops.dropout(v, 0.1, 78))
    kept = a != 0
    assert abs(float(kept.float().mean()) - 0.9) < 5e-3 and torch.allclose(a[kept], v[kept] / 0.9)
    # linear backward through the transposed operand forms
    M, N, K = 200, 96, 64
    dY, X, Wt = torch.randn(M, N, generator=g), torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    assert rel(ops.gemm_t(dY.to(D), Wt.to(D), tb=True), dY.double() @ Wt.double()) < 2e-6
    assert rel(ops.gemm_t(dY.to(D), X.to(D), ta=True, tb=True), dY.double().T @ X.double()) < 2e-6
    acc = torch.randn(N, K, generator=g)
    got = ops.gemm_t(dY.to(D), X.to(D), ta=True, tb=True, res=acc.to(D))
    assert rel(got, dY.double().T @ X.double() + acc.double()) < 2e-6


@pytest.mark.parametrize("n,T,H,HD,causal", [(3, 32, 8, 32, True), (2, 8, 8, 32, True), (2, 64, 2, 64, True), (2, 17, 4, 16, False)])
def test_small_attention_forward_and_backward_against_autograd(n, T, H, HD, causal):
    g = torch.Generator().manual_seed(T + HD)
    Dm = H * HD
    qkv = torch.randn(n, T, 3 * Dm, generator=g, dtype=torch.float64, requires_grad=True)
    dout = torch.randn(n, T, Dm, generator=g, dtype=torch.float64)
    q, k, v = (t.reshape(n, T, H, HD).transpose(1, 2) for t in qkv.split(Dm, dim=-1))
    s = (q @ k.transpose(-1, -2)) / HD ** 0.5
    if causal:
        s = s.masked_fill(torch.triu(torch.ones(T, T, dtype=torch.bool), diagonal=1), float("-inf"))
    out = (s.softmax(-1) @ v).transpose(1, 2).reshape(n, T, Dm)
    out.backward(dout)
    got = ops.small_attention(qkv.detach().float().to(D), H, causal=causal)
    assert rel(got, out) < 2e-6
    dq = ops.small_attention_bwd(qkv.detach().float().to(D), dout.float().to(D), H, causal=causal)
    assert rel(dq, qkv.grad) < 3e-6
    # the library's MFMA attention (the eval-mode path) agrees with the small kernel
    qf = qkv.detach().float().to(D)
    big = ops.attention(qf[..., :Dm], qf[..., Dm:2 * Dm], qf[..., 2 * Dm:], H, causal=causal, split=False)
    assert rel(big, out) < 5e-6


def test_small_attention_dropout_is_consistent_between_forward_and_backward():
    """with the weight dropout on, the backward must differentiate the SAME masked function: central differences of
    sum(out * dout) along random directions"""
    g = torch.Generator().manual_seed(3)
    n, T, H, HD, p, seed = 2, 16, 4, 16, 0.25, 1234
    qkv = torch.randn(n, T, 3 * H * HD, generator=g).to(D)
    dout = torch.randn(n, T, H * HD, generator=g).to(D)
    o0 = ops.small_attention(qkv, H, drop_p=p, seed=seed)
    assert torch.equal(o0, ops.small_attention(qkv, H, drop_p=p, seed=seed))
    assert not torch.equal(o0, ops.small_attention(qkv, H, drop_p=p, seed=seed + 1))
    dq = ops.small_attention_bwd(qkv, dout, H, drop_p=p, seed=seed)
    for trial in range(3):
        d = torch.randn(qkv.shape, generator=g).to(D)
        eps = 1e-2
        fp = (ops.small_attention(qkv + eps * d, H, drop_p=p, seed=seed).double() * dout.double()).sum()
        fm = (ops.small_attention(qkv - eps * d, H, drop_p=p, seed=seed).double() * dout.double()).sum()
        fd, an = float((fp - fm) / (2 * eps)), float((dq.double() * d.double()).sum())
        assert abs(fd - an) <= 2e-3 * max(1.0, abs(an)), (fd, an)


def test_soft_cross_entropy_over_dim1_and_the_l1_metric():
    g = torch.Generator().manual_seed(5)
    B, T, V = 6, 32, 32
    x = (torch.randn(B, T, V, generator=g, dtype=torch.float64) * 3).requires_grad_(True)
    tg = F.one_hot(torch.randint(0, V, (B, T), generator=g), V).double()
    loss = F.cross_entropy(x, tg)                      # [B, C=T, d1=V]: the reference's call (train_latent_model.py:329)
    loss.backward()
    gl, gd = ops.soft_ce_dim1(x.detach().float().to(D), tg.float().to(D))
    assert abs(float(gl) - float(loss.detach())) < 1e-5 * max(1.0, float(loss.detach())) and rel(gd, x.grad) < 5e-6
    gl2, none = ops.soft_ce_dim1(x.detach().float().to(D), tg.float().to(D), want_grad=False)
    assert none is None and torch.equal(gl, gl2)
    soft = torch.rand(B, T, V, generator=g, dtype=torch.float64)
    x2 = x.detach().clone().requires_grad_(True)
    l2 = F.cross_entropy(x2, soft)
    l2.backward()
    gl3, gd3 = ops.soft_ce_dim1(x2.detach().float().to(D), soft.float().to(D))
    assert abs(float(gl3) - float(l2.detach())) < 1e-5 * float(l2.detach()) and rel(gd3, x2.grad) < 5e-6
    l1 = F.l1_loss(F.one_hot(torch.argmax(x.detach(), dim=-1), V).double(), tg)
    assert abs(float(ops.argmax_l1(x.detach().float().to(D), tg.float().to(D))) - float(l1)) < 1e-6


def test_layernorm_backward_colsum_batch_sum_and_adamw_against_torch():
    g = torch.Generator().manual_seed(6)
    M, Dm = 300, 256
    x = (torch.randn(M, Dm, generator=g, dtype=torch.float64) * 2 + 0.5).requires_grad_(True)
    w = (torch.rand(Dm, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    b = torch.randn(Dm, generator=g, dtype=torch.float64).requires_grad_(True)
    dy, add = torch.randn(M, Dm, generator=g, dtype=torch.float64), torch.randn(M, Dm, generator=g, dtype=torch.float64)
    F.layer_norm(x, (Dm,), w, b).backward(dy)
    ws = torch.empty(2 * Dm * 1024, device=D)
    dw, db = torch.zeros(Dm, device=D), torch.zeros(Dm, device=D)
    dx = ops.layernorm_bwd(x.detach().float().to(D), w.detach().float().to(D), dy.float().to(D), dw, db, ws, dx_add=add.float().to(D))
    assert rel(dx, x.grad + add) < 3e-6 and rel(dw, w.grad) < 3e-6 and rel(db, b.grad) < 3e-6
    dw2, db2 = torch.zeros(Dm, device=D), torch.zeros(Dm, device=D)
    ops.layernorm_bwd(x.detach().float().to(D), w.detach().float().to(D), dy.float().to(D), dw2, db2, ws)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)                       # ordered sums: bitwise repeatable
    out = torch.ones(Dm, device=D)
    ops.colsum(dy.float().to(D), out, ws)
    assert rel(out, dy.sum(0) + 1.0) < 3e-6
    src = torch.randn(5, 8, 64, generator=g)
    dst = torch.full((8, 64), 2.0, device=D)
    ops.sum_batch(src.to(D), dst, accumulate=True)
    assert rel(dst, src.double().sum(0) + 2.0) < 2e-6
    # torch.optim.AdamW, five steps with changing gradients
    p0 = torch.randn(5000, generator=g)
    pt = p0.clone().double().requires_grad_(True)
    opt = torch.optim.AdamW([pt], lr=3e-3)
    p, m, v = p0.clone().to(D), torch.zeros(5000, device=D), torch.zeros(5000, device=D)
    for step in range(1, 6):
        gr = torch.randn(5000, generator=g) * (10.0 ** (step - 3))
        pt.grad = gr.double()
        opt.step()
        ops.adamw(p, gr.to(D), m, v, 3e-3, 0.01, step)
    assert rel(p, pt) < 2e-6


def _fixture():
    from test_oracle_golden import _latent_prior_train_fixture
    return _latent_prior_train_fixture()


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_training_step_matches_the_reference_module(prec):
    """logits, loss (class axis = dim 1), L1 metric, every gradient and the parameters after 1 and 3 AdamW steps against the
    fixture written by the reference's Latent_Model_Transformer + torch autograd + torch.optim.AdamW (dropout off)."""
    from test_oracle_golden import check_against_prior_train_fixture
    z, sd, vq = _fixture()
    m = LatentModelTransformer(vq, vq, vq, gemm_prec=prec).load_state_dict(sd)
    m.train()
    import actmi.latent_model as LM
    old, LM.DROPOUT_RATE = LM.DROPOUT_RATE, 0.0                   # parity mode: torch's dropout stream cannot be reproduced
    try:
        opt = m.configure_optimizer(float(z["lr"]))
        x, y = torch.from_numpy(z["inputs"]), torch.from_numpy(z["labels"])
        losses, p1 = [], None
        for it in range(int(z["steps"])):
            opt.zero_grad()
            logits = m(x)
            loss = m.cross_entropy(logits, y)
            loss.backward()
            if it == 0:
                assert np.abs(logits.cpu().numpy() - z["logits"]).max() < 1e-4
                assert abs(float(loss.l1_error) - float(z["l1_error"])) < 1e-6
                grads = {k: m.grad[k].cpu().numpy() for k in m.spec}
            opt.step()
            losses.append(loss.item())
            if it == 0:
                p1 = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
        p3 = {k: v.cpu().numpy() for k, v in m.state_dict().items()}
    finally:
        LM.DROPOUT_RATE = old
    assert np.abs(np.array(losses) - z["losses"]).max() < 2e-5, (losses, z["losses"])
    check_against_prior_train_fixture(z, grads, p1, p3, rel=1e-4)
    with pytest.raises(RuntimeError, match="backward without"):
        m._backward(torch.zeros(1, device=D))


def test_default_size_step_against_the_oracle_and_repeatability():
    """vq_class = vq_dim = 32 (the reference's commands), B = 16: gradients against the float64 oracle; two identical steps are
    bitwise identical; with dropout on, the loss is differentiated through the same masks (central differences)."""
    from oracle import act_ref as R
    import actmi.latent_model as LM
    vq, B = 32, 16
    sd = W.generate_latent_model_state_dict(latent_model_spec(vq, vq, vq), 9)
    g = torch.Generator().manual_seed(1)
    y = F.one_hot(torch.randint(0, vq, (B, vq), generator=g), vq).float()
    x = torch.cat([torch.zeros_like(y)[:, [0]], y[:, :-1]], dim=1)
    ref = R.latent_model_train_step({k: torch.from_numpy(v) for k, v in sd.items()}, x, y)
    m = LatentModelTransformer(vq, vq, vq).load_state_dict(sd)
    m.train()
    old, LM.DROPOUT_RATE = LM.DROPOUT_RATE, 0.0
    try:
        runs = []
        for _ in range(2):
            m.zero_grad()
            loss = m.cross_entropy(m(x), y)
            loss.backward()
            runs.append((loss.item(), m.grads.clone()))
    finally:
        LM.DROPOUT_RATE = old
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    assert abs(runs[0][0] - float(ref["loss"])) < 2e-5
    worst = 0.0
    for k in m.spec:
        r = ref["grads"][k]
        if float(r.abs().max()) < 1e-6:                    # exactly-zero gradients of the dim-1 cross entropy (see the CPU test)
            assert float(m.grad[k].abs().max()) < 1e-6, k
            continue
        worst = max(worst, rel(m.grad[k], r))
        assert rel(m.grad[k], r) < 2e-4, (k, rel(m.grad[k], r))
    print(f"prior training step, default size: loss {runs[0][0]:.6f}, worst gradient error {worst:.2e} of the tensor's maximum")
    # dropout on (the reference's training mode): same masks in forward and backward
    m.dropout_seed = 5

    def seeded_loss():
        m._fwd_count = 41                                  # the mask stream is keyed by (dropout_seed, forward count, site)
        return m.cross_entropy(m(x), y)

    m.zero_grad()
    l0 = seeded_loss()
    l0.backward()
    assert abs(l0.item() - runs[0][0]) > 1e-4             # dropout changed the function
    gsnap = m.grads.clone()
    gen = torch.Generator().manual_seed(2)
    for key in ("attention_blocks.2.mlp.0.weight", "attention_blocks.1.attn.in_proj_weight", "input_layer.weight"):
        d = torch.randn(m.sd[key].shape, generator=gen).to(D)
        o, n_, shp = m._off[key]
        an = float((gsnap[o:o + n_].view(shp).double() * d.double()).sum())
        eps = 2e-3
        m.sd[key].add_(eps * d)
        fp = seeded_loss().item()
        m.sd[key].sub_(2 * eps * d)
        fm = seeded_loss().item()
        m.sd[key].add_(eps * d)
        fd = (fp - fm) / (2 * eps)
        assert abs(fd - an) <= 0.03 * max(abs(an), 1e-3), (key, fd, an)


def test_train_latent_model_entry_point_learns_and_writes_the_checkpoints(tmp_path):
    """train_latent_model.py main(): a VQ-ACT policy_last.ckpt in ckpt_dir, synthetic episodes; the loop must lower the validation
    loss and leave latent_model_last.ckpt where eval_bc looks for it (imitate_episodes.py:252-262)."""
    import train_latent_model as TL
    from imitate_episodes import make_policy
    args = {"eval": False, "onscreen_render": False, "ckpt_dir": str(tmp_path), "policy_class": "ACT",
            "task_name": "sim_transfer_cube_scripted", "batch_size": 4, "seed": 0, "num_epochs": 3, "lr": 2e-3, "kl_weight": 10,
            "chunk_size": 20, "hidden_dim": 128, "dim_feedforward": 256, "temporal_agg": False, "use_vq": True, "vq_class": 8,
            "vq_dim": 8, "max_batch": 4, "dataset_dir": None}
    # a small VQ-ACT policy checkpoint with the shapes main() will build
    from actmi.constants import SIM_TASK_CONFIGS
    cams = SIM_TASK_CONFIGS[args["task_name"]]["camera_names"]
    pc = {"lr": 1e-4, "num_queries": 20, "kl_weight": 10, "hidden_dim": 128, "dim_feedforward": 256, "lr_backbone": 1e-5,
          "backbone": "resnet18", "enc_layers": 4, "dec_layers": 7, "nheads": 8, "camera_names": cams, "vq": True, "vq_class": 8,
          "vq_dim": 8, "action_dim": 16, "state_dim": 14, "max_batch": 4}
    pol = make_policy("ACT", pc)
    torch.save(pol.serialize(), os.path.join(str(tmp_path), "policy_last.ckpt"))
    del pol
    best_epoch, min_val, best_sd = TL.main(args)
    assert np.isfinite(min_val) and best_epoch >= 1, (best_epoch, min_val)     # epoch 0 validates the untrained prior
    for name in ("latent_model_last.ckpt", "latent_model_best.ckpt", f"latent_model_epoch_{best_epoch}_seed_0.ckpt"):
        sd = torch.load(os.path.join(str(tmp_path), name), weights_only=True)
        assert list(sd.keys()) == list(latent_model_spec(8, 8, 8).keys())
    m = LatentModelTransformer(8, 8, 8).load_state_dict(torch.load(os.path.join(str(tmp_path), "latent_model_last.ckpt"), weights_only=True))
    codes = m.generate(3, temperature=1.0)
    assert tuple(codes.shape) == (3, 8, 8) and torch.all(codes.sum(-1) == 1)
