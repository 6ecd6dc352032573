"""The CPU oracle against the golden vectors produced by the reference's own modules
(tools/gen_golden.py).  CPU only; sized to run in well under a minute."""
import os

import numpy as np
import pytest
import torch

from helpers import load_fixture, regenerate, sample_like, torch_sd
from oracle import act_ref as R
from actmi import weights as W


@pytest.mark.parametrize("name", ["tiny", "tiny_c3", "full3"])
def test_oracle_inference_matches_reference(name):
    z, cfg = load_fixture(name)
    sd_np, inp = regenerate(z, cfg)
    sd = torch_sd(sd_np)
    image = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))
    qpos = torch.from_numpy(inp["qpos"])
    stages = {}
    with torch.no_grad():
        a_hat, is_pad_hat, _, _ = R.detrvae_forward(sd, cfg, qpos, R.normalize_image(image), p="model.", stages=stages)
    # tolerance: fp32 CPU, same op order -> differences are thread-partition noise only
    assert np.abs(a_hat.numpy() - z["infer.a_hat"]).max() < 2e-5
    assert np.abs(sample_like(is_pad_hat.numpy(), z) - z["infer.is_pad_hat"].reshape(-1)).max() < 2e-5
    for k in ("cam0_conv1", "cam0_maxpool", "cam0_layer1", "cam0_layer4", "src", "memory"):
        got = sample_like(stages[k].numpy(), z)
        exp = z["stage." + k].reshape(-1)
        assert got.shape == exp.shape, k
        assert np.abs(got - exp).max() < 5e-5 * max(1.0, np.abs(exp).max()), k


@pytest.mark.parametrize("name", ["tiny", "tiny_c3"])
def test_oracle_training_losses_and_grads(name):
    z, cfg = load_fixture(name)
    sd_np, inp = regenerate(z, cfg)
    # single thread: torch CPU autograd is not run-to-run reproducible for the 1x1/s2 conv wgrad otherwise
    nthreads = torch.get_num_threads()
    torch.set_num_threads(1)
    sd = {k: v.clone().requires_grad_(not W.is_buffer(k[len("model."):])) for k, v in torch_sd(sd_np).items()}
    image = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))
    out = R.policy_call(sd, cfg, torch.from_numpy(inp["qpos"]), image, torch.from_numpy(inp["actions"]),
                        torch.from_numpy(inp["is_pad"]), torch.from_numpy(z["train.eps"]))
    for k in ("l1", "kl", "loss"):
        assert abs(float(out[k].detach()) - float(z["train." + k][0])) < 5e-5 * max(1.0, abs(float(out[k].detach()))), k
    assert np.abs(out["mu"].detach().numpy() - z["train.mu"]).max() < 2e-5
    assert np.abs(out["a_hat"].detach().numpy() - z["train.a_hat"]).max() < 2e-5
    out["loss"].backward()
    torch.set_num_threads(nthreads)
    names = [str(n) for n in z["grad_names"]]
    l2 = z["grad_l2"]
    none = set(str(n) for n in z["grad_none"])
    for n, ref_l2 in zip(names, l2):
        g = sd["model." + n].grad
        if n in none:
            # is_pad_head never receives a gradient
            assert g is None or float(g.abs().max()) == 0.0, n
            continue
        assert g is not None, n
        got = float(g.double().norm())
        assert abs(got - ref_l2) <= 2e-4 * max(ref_l2, 1e-3), (n, got, ref_l2)
    # dead decoder layers get exactly-zero (not None) gradients (quirk 1)
    if cfg.dec_layers > 1:
        i = names.index("transformer.decoder.layers.1.linear1.weight")
        assert l2[i] == 0.0
    for k in z.files:
        if k.startswith("grad.") and name == "tiny":
            g = sd["model." + k[5:]].grad.numpy()
            exp = z[k]
            assert np.abs(g - exp).max() <= 1e-4 * max(np.abs(exp).max(), 1e-3), k


def test_full4_fixture_is_consistent():
    """Full-size fixture: only hashes / shapes on CPU here (the forward itself runs in the gpu tests)."""
    z, cfg = load_fixture("full4")
    assert cfg.num_tokens == 1202 and cfg.num_cams == 4
    assert z["infer.a_hat"].shape == (2, 100, 16)
    assert int(z["n_params"]) == 117_431_073 or int(z["n_params"]) > 117_000_000


def test_sinusoid_and_pos_table():
    z, cfg = load_fixture("tiny")
    sd_np, _ = regenerate(z, cfg)
    assert sd_np["pos_table"].shape == (1, cfg.num_queries + 2, cfg.hidden_dim)
    p = R.position_embedding_sine(3, 5, 16)
    assert p.shape == (1, 32, 3, 5)


def test_temporal_ensemble_reference_semantics():
    """Transcription check of imitate_episodes.py:402-411 on a scripted stream, incl. an exact-zero row."""
    T, Q, A = 12, 5, 16
    ens = R.TemporalEnsembleRef(T, Q, A)
    rng = np.random.default_rng(0)
    chunks = rng.standard_normal((T, 1, Q, A)).astype(np.float32)
    chunks[3, 0, 2, 7] = 0.0            # row written at t=3 for step t=5 has a zero -> not "populated"
    for t in range(T):
        raw, pop = ens.step(t, torch.from_numpy(chunks[t]))
        assert raw.dtype == torch.float64 and raw.shape == (1, A)
        rows = [r for r in range(max(0, t - Q + 1), t + 1) if not (t == 5 and r == 3)]
        w = np.exp(-0.01 * np.arange(len(rows)))
        w /= w.sum()
        exp = sum(wi * chunks[r, 0, t - r].astype(np.float64) for wi, r in zip(w, rows))
        assert np.allclose(raw.numpy()[0], exp, atol=1e-12)
        assert int(pop.sum()) == len(rows)


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference only exists in the authoring container")
def test_reference_reproduces_committed_fixture():
    """Re-run the reference's own modules and compare with the committed tiny fixture (authoring container only)."""
    import importlib.util
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(root, "tools", "gen_golden.py"))
    gg = importlib.util.module_from_spec(spec)
    saved = {k: sys.modules.get(k) for k in ("policy", "IPython", "torchvision")}
    spec.loader.exec_module(gg)
    ref = gg.import_reference()
    try:
        z, cfg = load_fixture("tiny")
        sd_np, inp = regenerate(z, cfg)
        pol = gg.build_reference_policy(ref, cfg)
        pol.model.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
        pol.eval()
        with torch.no_grad():
            a = pol(torch.from_numpy(inp["qpos"]), torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"])))
        assert np.abs(a.numpy() - z["infer.a_hat"]).max() < 1e-6
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_oracle_vq_inference_matches_reference():
    """VQ-ACT inference (detr_vae.py:155-156, policy.py:322-332 with vq_sample): fixture made by the reference's own
    DETRVAE(vq=True) through ACTPolicy.__call__ (tools/gen_golden.py, job tiny_vq)."""
    z, cfg = load_fixture("tiny_vq")
    assert cfg.vq and cfg.vq_class == 4 and cfg.vq_dim == 8
    sd_np, inp = regenerate(z, cfg)
    sd = torch_sd(sd_np)
    assert tuple(sd["model.latent_out_proj.weight"].shape) == (cfg.hidden_dim, 32)
    assert tuple(sd["model.latent_proj.weight"].shape) == (32, cfg.hidden_dim)
    image = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))
    code = torch.from_numpy(inp["vq_sample"])
    assert torch.equal(code.sum(-1), torch.ones(code.shape[:2]))            # one-hot per class
    with torch.no_grad():
        a = R.policy_call(sd, cfg, torch.from_numpy(inp["qpos"]), image, vq_sample=code)
    assert np.abs(a.numpy() - z["infer.a_hat"]).max() < 2e-5
    # the code matters: another code gives another action chunk
    with torch.no_grad():
        b = R.policy_call(sd, cfg, torch.from_numpy(inp["qpos"]), image, vq_sample=code.roll(1, dims=-1))
    assert np.abs(b.numpy() - z["infer.a_hat"]).max() > 1e-4


def test_oracle_vq_training_matches_reference():
    """VQ-ACT training (detr_vae.py:137-145, policy.py:307-318): the code the reference drew with torch.multinomial is
    replayed as an input; losses (no KL), probabilities and the straight-through gradients agree."""
    z, cfg = load_fixture("tiny_vq")
    sd_np, inp = regenerate(z, cfg)
    B = int(z["batch"])
    torch.set_num_threads(1)
    sd = {k: v.clone().requires_grad_(not W.is_buffer(k[len("model."):])) for k, v in torch_sd(sd_np).items()}
    image = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))
    code = torch.from_numpy(z["train.vq_code"]).view(B, cfg.vq_class, cfg.vq_dim)
    out = R.policy_call(sd, cfg, torch.from_numpy(inp["qpos"]), image, torch.from_numpy(inp["actions"]),
                        torch.from_numpy(inp["is_pad"]), None, vq_sample=code)
    assert abs(float(out["l1"]) - float(z["train.l1"][0])) < 2e-5 and float(out["kl"]) == 0.0
    assert abs(float(out["loss"]) - float(z["train.loss"][0])) < 2e-5
    assert abs(float(out["vq_discrepancy"]) - float(z["train.vq_discrepancy"][0])) < 1e-6
    out["loss"].backward()
    for n in ("latent_proj.weight", "latent_out_proj.weight", "encoder.layers.0.linear1.weight", "action_head.weight"):
        g = sd["model." + n].grad
        exp = z["grad." + n].reshape(-1)
        assert np.abs(sample_like(g.numpy(), z) - exp).max() <= 1e-5 * max(1.0, np.abs(exp).max()), n


def _latent_prior_fixture():
    import hashlib, os
    from actmi.latent_model import latent_model_spec
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "latent_prior.npz"))
    vq_dim, vq_class = int(z["vq_dim"]), int(z["vq_class"])
    spec = latent_model_spec(vq_dim, vq_dim, vq_class)
    sd = W.generate_latent_model_state_dict(spec, int(z["seed_w"]))
    for k in ("output_layer.weight", "attention_blocks.2.mlp.0.weight"):
        assert hashlib.sha256(np.ascontiguousarray(sd[k]).tobytes()).hexdigest() == str(z["sha:" + k]), k
    return z, sd, vq_dim, vq_class


def test_oracle_latent_prior_matches_reference():
    """Latent_Model_Transformer.forward (latent_model.py:50-56), fixture from the reference's own module in eval mode"""
    z, sd, vq_dim, vq_class = _latent_prior_fixture()
    with torch.no_grad():
        lo = R.latent_model_forward({k: torch.from_numpy(v) for k, v in sd.items()}, torch.from_numpy(z["x"]))
    assert np.abs(lo.numpy() - z["logits"]).max() < 2e-5


def _latent_prior_train_fixture():
    import os
    from actmi.latent_model import latent_model_spec
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "latent_prior_train.npz"))
    vq = int(z["vq"])
    sd = W.generate_latent_model_state_dict(latent_model_spec(vq, vq, vq), int(z["seed_w"]))
    return z, sd, vq


def check_against_prior_train_fixture(z, grads, params1, params3, rel=2e-5):
    """gradients / parameters (dicts of arrays) against the stored ones: full tensors where the fixture holds them, else the
    first 256 elements and the norm"""
    for k in [f[2:] for f in z.files if f.startswith("g:")]:
        ref = z["g:" + k]
        got = np.asarray(grads[k], dtype=np.float64)
        part = got if ref.shape == got.shape else got.reshape(-1)[:256]
        if float(np.abs(ref).max()) < 1e-6:
            # a vector added at every position moves the logits by a constant along T -- the class axis of the reference's cross
            # entropy -- so output_layer.bias and the last LayerNorm's bias have EXACTLY zero gradient: rounding noise on both sides
            assert float(np.abs(got).max()) < 1e-6, k
            continue
        scale = float(np.abs(ref).max())
        assert np.abs(part - ref).max() <= rel * scale + 1e-9, (k, float(np.abs(part - ref).max()), scale)
        assert abs(np.sqrt((got ** 2).sum()) - float(z["gnorm:" + k])) <= rel * float(z["gnorm:" + k]) + 1e-9, k
        for tag, prm in (("p1:", params1), ("p3:", params3)):
            if prm is None:
                continue
            ref = z[tag + k]
            got = np.asarray(prm[k], dtype=np.float64)
            part = got if ref.shape == got.shape else got.reshape(-1)[:256]
            # AdamW's first steps move every weight by ~lr * g / (|g| + eps) whatever the gradient's size: absolute bound in units
            # of lr, over the elements whose gradient is clear of eps = 1e-8 (below it the update amplifies rounding noise of g)
            clear = np.abs(z["g:" + k]) > 1e-5 * max(float(np.abs(z["g:" + k]).max()), 1e-30)
            assert clear.mean() > 0.5, k
            assert np.abs(part - ref)[clear].max() <= 0.02 * float(z["lr"]), (tag, k, float(np.abs(part - ref)[clear].max()))


def test_oracle_latent_prior_training_step_matches_reference():
    """train_latent_model.py:323-343, 395-404 on the reference's own module: loss (class axis = dim 1), L1 metric, every
    gradient, and the parameters after 1 and 3 AdamW steps"""
    z, sd, vq = _latent_prior_train_fixture()
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    x, y = torch.from_numpy(z["inputs"]), torch.from_numpy(z["labels"])
    r1 = R.latent_model_train_step(tsd, x, y, lr=float(z["lr"]), steps=1)
    r3 = R.latent_model_train_step(tsd, x, y, lr=float(z["lr"]), steps=int(z["steps"]))
    assert np.abs(r1["logits"].numpy() - z["logits"]).max() < 2e-5
    assert abs(float(r1["loss"]) - float(z["losses"][0])) < 1e-5 and abs(float(r1["l1_error"]) - float(z["l1_error"])) < 1e-6
    assert np.abs(np.array(r3["losses"]) - z["losses"]).max() < 1e-5
    check_against_prior_train_fixture(z, {k: v.numpy() for k, v in r1["grads"].items()},
                                      {k: v.numpy() for k, v in r1["params_after"].items()},
                                      {k: v.numpy() for k, v in r3["params_after"].items()})

