"""End-to-end parity of the HIP inference path: golden fixtures produced by the reference's own modules
(tests/golden, tools/gen_golden.py) and the CPU oracle on other batch sizes.
Bar (BASELINE.json north_star): |a_hat - reference| <= 1e-4 absolute, fp32."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_fixture, regenerate, sample_like, torch_sd  # noqa: E402
from actmi import weights as W  # noqa: E402
from actmi.engine import ACTEngine  # noqa: E402

ATOL = 1e-4


def _engine(cfg, sd_np, max_batch, prec=None):
    eng = ACTEngine(cfg, max_batch=max_batch, gemm_prec=prec)
    eng.load_state_dict(sd_np)
    eng.finalize()
    return eng


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
@pytest.mark.parametrize("name", ["tiny", "tiny_c3", "full3", "full4"])
def test_forward_matches_reference_golden(name, prec):
    z, cfg = load_fixture(name)
    sd_np, inp = regenerate(z, cfg, with_actions=True)
    B = int(z["batch"])
    eng = _engine(cfg, sd_np, B, prec)
    d = eng.device
    qpos = torch.from_numpy(inp["qpos"]).to(d)
    img_u8 = torch.from_numpy(inp["image_u8"]).to(d)
    a = eng.forward_infer(qpos, img_u8).cpu().numpy()
    err = np.abs(a - z["infer.a_hat"]).max()
    print(f"{name} [{prec}]: max|a_hat - ref| = {err:.3e}")
    assert err <= ATOL
    # the reference's own input contract (f32 NCHW in [0,1]) gives the same result
    img_f32 = torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"])).to(d)
    a2 = eng.forward_infer(qpos, img_f32).cpu().numpy()
    assert np.abs(a2 - z["infer.a_hat"]).max() <= ATOL
    # hs of decoder layer 0
    hs = eng.debug_tensor("hs").cpu().numpy()
    assert np.abs(sample_like(hs, z) - z["infer.hs"].reshape(-1)).max() <= 2e-4
    # stage activations (camera 0 is the first B images of the camera-major maps)
    fh, fw = cfg.feat_hw
    for stage in ["conv1", "maxpool", "layer1", "layer4", "src"]:
        eng.debug_stop_after(stage)
        eng.forward_infer(qpos, img_u8)
        t = eng.debug_tensor(stage).cpu()
        key = "stage." + (stage if stage == "src" else "cam0_" + stage)
        exp = z[key].reshape(-1)
        if stage == "src":
            got = t.view(B, cfg.num_tokens, cfg.hidden_dim).permute(1, 0, 2)         # reference is [N,B,D]
        else:
            Cc = t.numel() // (cfg.num_cams * B)
            full = t.view(cfg.num_cams, B, -1)
            got0 = full[0]
            # recover H,W,C of the map
            ref_shape = {"conv1": (cfg.image_h // 2, cfg.image_w // 2, cfg.base_width)}
            ch = {"conv1": cfg.base_width, "maxpool": cfg.base_width, "layer1": cfg.base_width,
                  "layer4": 8 * cfg.base_width}[stage]
            hw = got0.shape[1] // ch
            # H/W from the conv arithmetic
            def down(x, k, s, p):
                return (x + 2 * p - k) // s + 1
            h, w = down(cfg.image_h, 7, 2, 3), down(cfg.image_w, 7, 2, 3)
            if stage != "conv1":
                h, w = down(h, 3, 2, 1), down(w, 3, 2, 1)
            if stage == "layer4":
                h, w = fh, fw
            assert h * w == hw
            got = got0.view(B, h, w, ch).permute(0, 3, 1, 2)                           # NCHW like the reference hook
        g = sample_like(got.contiguous().numpy(), z)
        tol = 1e-4 * max(1.0, float(np.abs(exp).max()))
        assert np.abs(g - exp).max() <= tol, stage
    eng.debug_stop_after("")


def test_forward_batch_sizes_against_oracle():
    """Other batch sizes than the fixtures', checked against the CPU oracle (same seeded inputs)."""
    from oracle import act_ref as R
    from actmi.config import tiny_config
    cfg = tiny_config(camera_names=["a", "b", "c"], image_h=96, image_w=128)
    sd_np = W.generate_state_dict(cfg, seed=11)
    eng = _engine(cfg, sd_np, 5)
    sd = torch_sd(sd_np)
    for B in (1, 5, 3):
        inp = W.generate_inputs(cfg, B, seed=100 + B)
        with torch.no_grad():
            exp = R.policy_call(sd, cfg, torch.from_numpy(inp["qpos"]),
                                torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))).numpy()
        got = eng.forward_infer(torch.from_numpy(inp["qpos"]).cuda(), torch.from_numpy(inp["image_u8"]).cuda()).cpu().numpy()
        assert np.abs(got - exp).max() <= ATOL


@pytest.mark.parametrize("cams,h,w", [(["only"], 64, 64), (["a", "b", "c", "d", "e"], 64, 96)])
def test_forward_other_camera_counts(cams, h, w):
    """one camera and five cameras (the fixtures cover 2, 3 and 4): token layout 2 + C*fh*fw, per-camera weights"""
    from oracle import act_ref as R
    from actmi.config import tiny_config
    cfg = tiny_config(camera_names=cams, image_h=h, image_w=w)
    sd_np = W.generate_state_dict(cfg, seed=21)
    eng = _engine(cfg, sd_np, 2)
    inp = W.generate_inputs(cfg, 2, seed=8)
    with torch.no_grad():
        exp = R.policy_call(torch_sd(sd_np), cfg, torch.from_numpy(inp["qpos"]),
                            torch.from_numpy(W.u8_nhwc_to_f32_nchw(inp["image_u8"]))).numpy()
    got = eng.forward_infer(torch.from_numpy(inp["qpos"]).cuda(), torch.from_numpy(inp["image_u8"]).cuda()).cpu().numpy()
    assert np.abs(got - exp).max() <= ATOL


def test_batch_independence_and_determinism():
    """Size-independent properties at the benchmark configuration (C=4, 480x640, B=8): a sample's output does not
    depend on its batch neighbours (FrozenBN => no batch statistics) and repeated runs are bit-identical."""
    from actmi.config import ACTConfig
    cfg = ACTConfig()
    sd_np = W.generate_state_dict(cfg, seed=0)
    eng = _engine(cfg, sd_np, 8)
    inp = W.generate_inputs(cfg, 8, seed=5)
    q = torch.from_numpy(inp["qpos"]).cuda()
    im = torch.from_numpy(inp["image_u8"]).cuda()
    a8 = eng.forward_infer(q, im).clone()
    a8b = eng.forward_infer(q, im).clone()
    assert torch.equal(a8, a8b)
    a1 = eng.forward_infer(q[3:4].contiguous(), im[3:4].contiguous())
    assert float((a1[0] - a8[3]).abs().max()) <= 2e-5
    assert torch.isfinite(a8).all()


def test_f16x3_agrees_with_native_fp32_at_benchmark_size():
    """The benchmark configuration (C=4, 480x640, B=8) has no golden fixture (the oracle takes minutes there): pin the
    default arithmetic (fp32 products from three fp16 MFMA products) against the native fp32 MFMA path on the same
    weights and inputs -- the same bound as run-to-run summation-order noise of the fp32 path itself."""
    from actmi.config import ACTConfig
    cfg = ACTConfig()
    sd_np = W.generate_state_dict(cfg, seed=0)
    inp = W.generate_inputs(cfg, 8, seed=6)
    q = torch.from_numpy(inp["qpos"]).cuda()
    im = torch.from_numpy(inp["image_u8"]).cuda()
    outs = {}
    for prec in ("f16x3", "f32"):
        eng = _engine(cfg, sd_np, 8, prec)
        outs[prec] = eng.forward_infer(q, im).clone()
        del eng
    d = float((outs["f16x3"] - outs["f32"]).abs().max())
    print(f"benchmark size: max|a_hat(f16x3) - a_hat(f32)| = {d:.3e}, |a_hat| max {float(outs['f32'].abs().max()):.3f}")
    assert d <= 3e-5


def test_single_query_split_contraction_matches_unsplit(monkeypatch):
    """B = 1 (the reference's own rollout batch, imitate_episodes.py:390-399): layer3/4 convolutions and the K = 3200
    FFN products run with their contraction split over the grid + a fixed-order combine pass; the result agrees with the
    unsplit launches (ACTMI_FWD_SPLITK=0) to summation-order noise and is bitwise repeatable."""
    from actmi.config import ACTConfig
    cfg = ACTConfig()
    sd_np = W.generate_state_dict(cfg, seed=0)
    inp = W.generate_inputs(cfg, 1, seed=9)
    q = torch.from_numpy(inp["qpos"]).cuda()
    im = torch.from_numpy(inp["image_u8"]).cuda()
    eng = _engine(cfg, sd_np, 1, "f16x3")
    a = eng.forward_infer(q, im).clone()
    b = eng.forward_infer(q, im).clone()
    assert torch.equal(a, b)
    del eng
    monkeypatch.setenv("ACTMI_FWD_SPLITK", "0")
    eng = _engine(cfg, sd_np, 1, "f16x3")
    c = eng.forward_infer(q, im).clone()
    d = float((a - c).abs().max())
    print(f"B=1: max|a_hat(split) - a_hat(unsplit)| = {d:.3e}")
    assert 0.0 < d <= 3e-5                    # > 0: the split path did run


@pytest.mark.parametrize("cams,batch", [(["a", "b"], 3), (["top"], 1)])
def test_stem_with_fused_vertical_pool_is_bit_identical(monkeypatch, cams, batch):
    """Inference lets conv1 write max over conv rows (2a-1, 2a, 2a+1) and finishes the 3x3/s2 pool row-wise; the same
    maxima as conv1 -> max pool, so a_hat must not change by a single bit (full-size images: 5 column strips, several
    row segments with a halo step each)."""
    from actmi.config import ACTConfig
    cfg = ACTConfig(camera_names=cams, enc_layers=1, dim_feedforward=256)
    sd_np = W.generate_state_dict(cfg, seed=4)
    inp = W.generate_inputs(cfg, batch, seed=8)
    q = torch.from_numpy(inp["qpos"]).cuda()
    im = torch.from_numpy(inp["image_u8"]).cuda()
    eng = _engine(cfg, sd_np, batch, "f16x3")
    a = eng.forward_infer(q, im).clone()
    del eng
    monkeypatch.setenv("ACTMI_CONV1_VPOOL", "0")
    eng = _engine(cfg, sd_np, batch, "f16x3")
    b = eng.forward_infer(q, im).clone()
    assert torch.isfinite(a).all() and torch.equal(a, b)


def test_state_dict_round_trip_and_errors():
    from actmi.config import tiny_config
    cfg = tiny_config()
    sd_np = W.generate_state_dict(cfg, seed=2)
    eng = ACTEngine(cfg, max_batch=2)
    eng.load_state_dict({"model." + k: v for k, v in sd_np.items()}, prefix="model.")
    back = eng.state_dict(prefix="model.")
    assert list(back.keys()) == ["model." + k for k in sd_np]
    for k, v in sd_np.items():
        assert np.array_equal(back["model." + k].numpy(), v)
    with pytest.raises(RuntimeError):
        eng.load_state_dict({"bogus": np.zeros(3, np.float32)})
    with pytest.raises(ValueError):
        eng.forward_infer(torch.zeros(3, cfg.state_dim).cuda(),
                          torch.zeros(3, cfg.num_cams, cfg.image_h, cfg.image_w, 3, dtype=torch.uint8).cuda())


def test_hipgraph_replay_equals_eager():
    """The forward allocates nothing and never syncs: captured into a hipGraph it must give bit-identical results."""
    from actmi.config import tiny_config
    from actmi import ops
    cfg = tiny_config(camera_names=["a", "b", "c"])
    eng = _engine(cfg, W.generate_state_dict(cfg, seed=3), 4)
    ens_g = ops.TemporalEnsemble(4, cfg.num_queries, cfg.action_dim, 0.01, eng.device)
    ens_e = ops.TemporalEnsemble(4, cfg.num_queries, cfg.action_dim, 0.01, eng.device)
    replay = eng.capture_infer(4, with_ensemble=ens_g)
    for t in range(3):
        inp = W.generate_inputs(cfg, 4, seed=50 + t)
        q, im = torch.from_numpy(inp["qpos"]).cuda(), torch.from_numpy(inp["image_u8"]).cuda()
        a_g, raw_g = replay(q, im)
        a_g, raw_g = a_g.clone(), raw_g.clone()
        a_e = eng.forward_infer(q, im).clone()
        raw_e = ens_e.step(a_e).clone()
        assert torch.equal(a_g, a_e)
        assert torch.equal(raw_g, raw_e)


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_vq_inference_matches_reference_golden(prec):
    """VQ-ACT: ACTPolicy.__call__(qpos, image, vq_sample=code) against the reference's DETRVAE(vq=True) fixture."""
    z, cfg = load_fixture("tiny_vq")
    sd_np, inp = regenerate(z, cfg)
    B = int(z["batch"])
    eng = _engine(cfg, sd_np, B, prec)
    d = eng.device
    qpos = torch.from_numpy(inp["qpos"]).to(d)
    img = torch.from_numpy(inp["image_u8"]).to(d)
    code = torch.from_numpy(inp["vq_sample"]).to(d)
    a = eng.forward_infer(qpos, img, vq_sample=code).cpu().numpy()
    err = np.abs(a - z["infer.a_hat"]).max()
    print(f"tiny_vq [{prec}]: max|a_hat - ref| = {err:.3e}")
    assert err <= ATOL
    with pytest.raises(ValueError):
        eng.forward_infer(qpos, img)                      # a vq policy needs its code
    other = eng.forward_infer(qpos, img, vq_sample=code.roll(1, dims=-1)).cpu().numpy()
    assert np.abs(other - z["infer.a_hat"]).max() > 1e-4


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_latent_prior_matches_reference_golden(prec):
    """VQ-ACT prior (latent_model.py:35-72) on the library's kernels: logits against the reference module's, and the
    structural contract of generate()."""
    from test_oracle_golden import _latent_prior_fixture
    from actmi.latent_model import LatentModelTransformer
    z, sd, vq_dim, vq_class = _latent_prior_fixture()
    m = LatentModelTransformer(vq_dim, vq_dim, vq_class, gemm_prec=prec).load_state_dict(sd)
    logits = m(torch.from_numpy(z["x"])).cpu().numpy()
    err = np.abs(logits - z["logits"]).max()
    print(f"latent prior [{prec}]: max|logits - ref| = {err:.3e}")
    assert err <= 1e-4
    # causality: logits at position t do not depend on later inputs
    x2 = torch.from_numpy(z["x"]).clone()
    x2[:, -1] = 0.0
    assert np.array_equal(m(x2).cpu().numpy()[:, :-1], logits[:, :-1])
    # generate(): seq_len one-hot codes per sample, reproducible in the seed, first step = the draw from forward(zeros)
    a = m.generate(5, temperature=1.0, seed=3)
    assert tuple(a.shape) == (5, vq_class, vq_dim) and torch.equal(a.sum(-1), torch.ones(5, vq_class, device=a.device))
    assert torch.equal(a, m.generate(5, temperature=1.0, seed=3)) and not torch.equal(a, m.generate(5, temperature=1.0, seed=4))
    cold = m.generate(5, temperature=1e-3, seed=9)               # near-greedy: argmax of the step logits
    step0 = m(torch.zeros(5, 1, vq_dim))[:, -1]
    assert torch.equal(cold[:, 0].argmax(-1), step0.argmax(-1))


def test_sample_onehot_distribution():
    from actmi import ops
    g = torch.Generator().manual_seed(2)
    logits = torch.randn(4, 6, generator=g).cuda()
    p = torch.softmax(logits.double() / 0.7, -1).cpu()
    n, cnt = 2000, torch.zeros(4, 6, dtype=torch.float64)
    for s in range(n):
        cnt += ops.sample_onehot(logits, temperature=0.7, seed=s).cpu().double()
    assert float((cnt / n - p).abs().max()) < 4.5 * float((p * (1 - p) / n).sqrt().max())
    code, probs = ops.sample_onehot(logits, temperature=0.7, seed=1, want_probs=True)
    assert float((probs.cpu().double() - p).abs().max()) < 1e-6 and torch.equal(code.sum(-1), torch.ones(4).cuda())
