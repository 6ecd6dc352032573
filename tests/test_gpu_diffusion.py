"""DiffusionPolicy inference path (reference policy.py:20-241).  Kernel-level checks against torch's own
GroupNorm / Mish / conv1d / conv_transpose1d / softmax (torch IS present, so these pieces are pinned by torch), and the
whole policy against oracle/diffusion_ref.py -- a restatement of robomimic / diffusers from their published definitions:
PARITY UNPINNED against the reference itself (neither package is importable offline; no fixture of the reference covers it)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from actmi import ops  # noqa: E402
from actmi.diffusion import DiffusionNet, diffusion_state_dict_spec, generate_diffusion_state_dict  # noqa: E402

D = "cuda:0"


def rel(got, exp):
    got, exp = got.detach().cpu().double(), exp.detach().cpu().double()
    return float((got - exp).abs().max() / (exp.abs().max() + 1e-30))


@pytest.mark.parametrize("act", [None, "relu", "mish"])
def test_groupnorm_forms(act):
    g = torch.Generator().manual_seed(1)
    n, P, C, G = 3, 77, 64, 4
    x, res = torch.randn(n, P, C, generator=g) * 3 + 1, torch.randn(n, P, C, generator=g)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    fs, fb = torch.randn(n, C, generator=g), torch.randn(n, C, generator=g)
    gn = F.group_norm(x.double().permute(0, 2, 1), G, w.double(), b.double(), 1e-5).permute(0, 2, 1)
    a = {None: lambda v: v, "relu": F.relu, "mish": F.mish}[act]
    assert rel(ops.groupnorm(x.to(D), w.to(D), b.to(D), G, act=act), a(gn)) < 2e-6
    assert rel(ops.groupnorm(x.to(D), w.to(D), b.to(D), G, act=act, res=res.to(D)), a(gn + res.double())) < 2e-6
    exp = a(gn) * fs.double()[:, None] + fb.double()[:, None] + res.double()
    got = ops.groupnorm(x.to(D), w.to(D), b.to(D), G, act=act, res=res.to(D), res_after=True, film=(fs.to(D), fb.to(D)))
    assert rel(got, exp) < 2e-6


@pytest.mark.parametrize("n,H,W,C", [(3, 60, 80, 64), (2, 33, 47, 128), (5, 120, 160, 64)])
def test_groupnorm_large_maps(n, H, W, C):
    """the chunked-statistics path of the ResNet trunk's maps (>= 16k values per sample and group): an offset far from zero
    (no cancellation in the variance), residual + ReLU, written into a caller-provided view; bitwise repeatable."""
    g = torch.Generator().manual_seed(n + H)
    G = C // 16
    x = torch.randn(n, H, W, C, generator=g) * 2 + 30.0
    res = torch.randn(n, H, W, C, generator=g)
    w, b = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    gn = F.group_norm(x.double().permute(0, 3, 1, 2), G, w.double(), b.double(), 1e-5).permute(0, 2, 3, 1)
    out = torch.empty(2, n, H, W, C, device=D)
    got = ops.groupnorm(x.to(D), w.to(D), b.to(D), G, act="relu", res=res.to(D), out=out[1])
    assert got.data_ptr() == out[1].data_ptr()
    assert rel(got, F.relu(gn + res.double())) < 3e-6
    assert rel(ops.groupnorm(x.to(D), w.to(D), b.to(D), G), gn) < 3e-6
    assert torch.equal(got, ops.groupnorm(x.to(D), w.to(D), b.to(D), G, act="relu", res=res.to(D)))


def test_spatial_softmax_and_ddim_step_and_mish():
    g = torch.Generator().manual_seed(2)
    n, H, W, K = 2, 15, 20, 32
    lg = torch.randn(n, H * W, K, generator=g) * 4
    att = F.softmax(lg.double().permute(0, 2, 1), dim=-1)                           # [n,K,HW]
    px, py = np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H))
    exp = torch.stack([(att * torch.from_numpy(px.reshape(-1))).sum(-1), (att * torch.from_numpy(py.reshape(-1))).sum(-1)], -1)
    assert float((ops.spatial_softmax(lg.to(D), H, W).cpu().double() - exp).abs().max()) < 1e-6
    x, e = torch.randn(1000, generator=g) * 2, torch.randn(1000, generator=g)
    a_t, a_p = 0.3, 0.7
    x0 = ((x.double() - (1 - a_t) ** 0.5 * e.double()) / a_t ** 0.5).clamp(-1, 1)
    exp = a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * e.double()
    assert float((ops.ddim_step(x.to(D).clone(), e.to(D), a_t, a_p).cpu().double() - exp).abs().max()) < 1e-6
    v = torch.linspace(-30, 30, 601)
    assert float((ops.mish(v.to(D)).cpu().double() - F.mish(v.double())).abs().max()) < 1e-5


@pytest.mark.parametrize("k,stride,pad,T", [(5, 1, 2, 16), (3, 2, 1, 16), (1, 1, 0, 8), (3, 1, 1, 7)])
def test_conv1d_through_unfold_and_gemm(k, stride, pad, T):
    g = torch.Generator().manual_seed(3)
    B, Ci, Co = 3, 32, 48
    x, w, b = torch.randn(B, T, Ci, generator=g), torch.randn(Co, Ci, k, generator=g) / (Ci * k) ** 0.5, torch.randn(Co, generator=g)
    exp = F.conv1d(x.double().permute(0, 2, 1), w.double(), b.double(), stride=stride, padding=pad).permute(0, 2, 1)
    cols = ops.unfold1d(x.to(D), k, stride, pad)
    got = ops.gemm(cols.reshape(-1, k * Ci), w.permute(0, 2, 1).reshape(Co, -1).contiguous().to(D), bias=b.to(D), prec="f16x3")
    assert rel(got.reshape(B, -1, Co), exp) < 2e-6


def test_conv_transpose1d_through_unfold_and_gemm():
    g = torch.Generator().manual_seed(4)
    B, T, Cc = 2, 8, 64
    x, w, b = torch.randn(B, T, Cc, generator=g), torch.randn(Cc, Cc, 4, generator=g) / 16, torch.randn(Cc, generator=g)
    exp = F.conv_transpose1d(x.double().permute(0, 2, 1), w.double(), b.double(), stride=2, padding=1).permute(0, 2, 1)
    cols = ops.unfold1d(x.to(D), 4, 2, 1, transposed=True)
    assert cols.shape[1] == 2 * T
    got = ops.gemm(cols.reshape(-1, 4 * Cc), w.permute(1, 2, 0).reshape(Cc, -1).contiguous().to(D), bias=b.to(D), prec="f16x3")
    assert rel(got.reshape(B, 2 * T, Cc), exp) < 2e-6


@pytest.mark.parametrize("prec", ["f16x3", "f32"])
def test_policy_against_the_restated_oracle(prec):
    from oracle import diffusion_ref as R
    from actmi import weights as W
    cams, B, T, H, Wd = ["a", "b"], 2, 16, 64, 96
    spec = diffusion_state_dict_spec(cams)
    sd = generate_diffusion_state_dict(spec, seed=5)
    net = DiffusionNet(cams, prediction_horizon=T, gemm_prec=prec)
    net.load_state_dict(sd)
    img = W.rand_u8(7, "img", (B, len(cams), H, Wd, 3))
    qpos = W.normal(7, "qpos", B * 14).reshape(B, 14).astype(np.float32)
    noise = W.normal(7, "noise", B * T * 16).reshape(B, T, 16).astype(np.float32)
    tsd = {k: torch.from_numpy(v) for k, v in sd.items()}
    img_f = torch.from_numpy(img).permute(0, 1, 4, 2, 3).double().div(255.0).float()
    with torch.no_grad():
        cond_ref = R.obs_features(tsd, len(cams), torch.from_numpy(qpos), img_f)
        eps_ref = R.unet(tsd, torch.from_numpy(noise), 45, cond_ref)
        out_ref = R.policy_call(tsd, len(cams), torch.from_numpy(qpos), img_f, torch.from_numpy(noise))
    cond = net.obs_cond(torch.from_numpy(qpos).to(D), torch.from_numpy(img).to(D))
    e_cond = float((cond.cpu() - cond_ref).abs().max())
    eps = net.unet(torch.from_numpy(noise).to(D), 45, cond)
    e_eps = float((eps.cpu() - eps_ref).abs().max())
    out = net.forward_infer(torch.from_numpy(qpos).to(D), torch.from_numpy(img).to(D), noise=torch.from_numpy(noise))
    e_out = float((out.cpu() - out_ref).abs().max())
    print(f"diffusion [{prec}]: obs_cond {e_cond:.2e}, one UNet pass {e_eps:.2e} (|eps| max {float(eps_ref.abs().max()):.2f}), "
          f"10 DDIM steps {e_out:.2e}")
    assert e_cond <= 1e-4 and e_eps <= 1e-4 * max(1.0, float(eps_ref.abs().max())) and e_out <= 1e-3
    assert torch.isfinite(out).all() and tuple(out.shape) == (B, T, 16)


def test_policy_wrapper_surface():
    from policy import DiffusionPolicy
    cfg = {"lr": 1e-4, "camera_names": ["top"], "action_dim": 16, "observation_horizon": 1, "action_horizon": 8,
           "prediction_horizon": 16, "num_queries": 16, "num_inference_timesteps": 10, "ema_power": 0.75, "vq": False}
    pol = DiffusionPolicy(cfg)
    q = torch.zeros(1, 14, device=D)
    img = torch.rand(1, 1, 3, 64, 96, device=D)
    noise = torch.randn(1, 16, 16)
    a = pol(q, img, noise=noise)
    assert tuple(a.shape) == (1, 16, 16) and float(a.abs().max()) <= 1.0 + 1e-6
    st = pol.serialize()
    pol2 = DiffusionPolicy(cfg, init_seed=9)
    assert repr(pol2.deserialize(st)) == "<All keys matched successfully>"
    assert torch.equal(pol2(q, img, noise=noise), a)
    with pytest.raises(NotImplementedError):
        pol(q, img, actions=torch.zeros(1, 16, 16, device=D), is_pad=torch.zeros(1, 16, dtype=torch.bool, device=D))


def test_whole_query_graph_replay_equals_eager_launches():
    """DiffusionNet.capture_infer: observation trunk + all DDIM steps as ONE hipGraph (the step loop was
    issued op by op from Python).  Replays are bit-identical to eager launches, for fresh inputs too, in both image formats."""
    from actmi import weights as W
    cams, B, T, H, Wd = ["a", "b"], 3, 16, 64, 96
    net = DiffusionNet(cams, prediction_horizon=T)
    net.load_state_dict(generate_diffusion_state_dict(diffusion_state_dict_spec(cams), seed=8))
    g = torch.Generator().manual_seed(0)
    for fmt in ("u8", "f32"):
        img0 = torch.from_numpy(W.rand_u8(3, "gimg", (B, len(cams), H, Wd, 3))).to(D)
        if fmt == "f32":
            img0 = img0.permute(0, 1, 4, 2, 3).float().div(255.0).contiguous()
        replay = net.capture_infer(B, img0)
        for trial in range(3):
            qpos = torch.randn(B, 14, generator=g).to(D)
            noise = torch.randn(B, T, 16, generator=g).to(D)
            img = img0 if trial == 0 else (img0.flip(0).contiguous())
            exp = net.forward_infer(qpos, img, noise=noise)
            got = replay(qpos, img, noise=noise)
            assert torch.equal(got, exp), (fmt, trial)
        r1 = replay(qpos, img).clone()                    # no noise given: a fresh Gaussian start per call
        r2 = replay(qpos, img).clone()
        assert not torch.equal(r1, r2) and torch.isfinite(r1).all()
