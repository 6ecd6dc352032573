"""Host logic of the Diffusion branches of imitate_episodes.py on CPU: the config dict, the action
de-normalisation, the eval-time centre crop + resize and "no ensembling" follow the reference's Diffusion path
(imitate_episodes.py:95-106, 214-224, 290-293, 374, 417-423); plus train_bc's in-loop evaluation and pretrain load
(:548-550, :590-596).  Test doubles stand in for the GPU policies (the product policies have no CPU fallback)."""
import os

import numpy as np
import pytest
import torch

import imitate_episodes as IE
from actmi.envs import SyntheticEnv

Q, A, T, S = 4, 16, 6, 14
CAMS = ["top", "left_wrist"]


def _args(policy_class, **over):
    a = {"task_name": "sim_transfer_cube_scripted", "policy_class": policy_class, "ckpt_dir": "/tmp/x", "batch_size": 8, "seed": 0,
         "num_steps": 10, "lr": 1e-4, "kl_weight": 10, "chunk_size": 32, "hidden_dim": 512, "dim_feedforward": 3200,
         "temporal_agg": False, "eval_every": 500, "validate_every": 500, "save_every": 500}
    a.update(over)
    return a


def test_build_config_diffusion_matches_the_reference_dict():
    cfg = IE.build_config(_args("Diffusion"))
    cams = cfg["camera_names"]
    # the literal of reference imitate_episodes.py:97-106 with chunk_size = 32
    assert cfg["policy_config"] == {"lr": 1e-4, "camera_names": cams, "action_dim": 16, "observation_horizon": 1, "action_horizon": 8,
                                    "prediction_horizon": 32, "num_queries": 32, "num_inference_timesteps": 10, "ema_power": 0.75,
                                    "vq": False}
    assert cfg["policy_class"] == "Diffusion"
    act = IE.build_config(_args("ACT"))["policy_config"]
    assert act["num_queries"] == 32 and act["enc_layers"] == 4 and act["dec_layers"] == 7 and act["nheads"] == 8
    with pytest.raises(NotImplementedError):
        IE.build_config(_args("CNNMLP"))
    assert IE.build_config(_args("ACT", load_pretrain=True, pretrain_ckpt_path="/p.ckpt"))["load_pretrain"] is True


def test_post_process_by_policy_class():
    stats = {"action_mean": np.full(A, 2.0), "action_std": np.full(A, 3.0), "action_min": np.full(A, -4.0), "action_max": np.full(A, 6.0)}
    a = np.linspace(-1, 1, A)
    assert np.allclose(IE.make_post_process("ACT", stats)(a), a * 3.0 + 2.0)
    # reference :291: ((a + 1) / 2) * (max - min) + min
    assert np.allclose(IE.make_post_process("Diffusion", stats)(a), ((a + 1) / 2) * 10.0 - 4.0)
    assert IE.make_post_process("Diffusion", stats)(np.array([-1.0]))[0] == -4.0 and IE.make_post_process("Diffusion", stats)(np.array([1.0]))[0] == 6.0


def test_center_crop_resize_geometry():
    H, W = 480, 640
    img = torch.zeros(2, 3, 3, H, W)
    # reference :218-219: rows int(H*0.025) .. int(H*0.975), columns likewise
    r0, r1, c0, c1 = int(H * 0.05 / 2), int(H * 1.95 / 2), int(W * 0.05 / 2), int(W * 1.95 / 2)
    assert (r0, r1, c0, c1) == (12, 468, 16, 624)
    img[..., r0:r1, c0:c1] = 1.0                      # exactly the crop window is ones: the resized crop is all ones
    out = IE.center_crop_resize(img)
    assert out.shape == img.shape and float((out - 1).abs().max()) < 1e-6
    img2 = torch.ones_like(img)
    img2[..., r0:r1, c0:c1] = 0.0                     # everything OUTSIDE the window: none of it survives
    assert float(IE.center_crop_resize(img2).abs().max()) < 1e-6
    # a horizontal ramp stays a ramp between the window's end values (bilinear, align_corners=False)
    ramp = torch.linspace(0, 1, W).view(1, 1, 1, 1, W).expand(1, 1, 1, H, W).contiguous()
    o = IE.center_crop_resize(ramp)[0, 0, 0, 100]
    assert float(o[0]) == pytest.approx(c0 / (W - 1), abs=2e-3) and float(o[-1]) == pytest.approx((c1 - 1) / (W - 1), abs=2e-3)
    assert bool((o[1:] >= o[:-1] - 1e-7).all())


class StubDiffusion:
    """records what eval_bc hands to the policy"""

    def __init__(self):
        self.images, self.n = [], 0

    def __call__(self, qpos, image):
        self.images.append(image)
        self.n += 1
        E = qpos.shape[0]
        return (torch.arange(Q, dtype=torch.float32).view(1, Q, 1) * 0.1 + self.n).expand(E, Q, A).contiguous()

    def eval(self):
        return self


def _diff_config(tmp, temporal_agg):
    return {"ckpt_dir": str(tmp), "state_dim": S, "policy_class": "Diffusion",
            "policy_config": {"num_queries": Q, "action_dim": A}, "camera_names": CAMS, "episode_len": T,
            "task_name": "sim_transfer_cube_scripted", "temporal_agg": temporal_agg}


@pytest.mark.parametrize("temporal_agg", [False, True])
def test_eval_bc_diffusion_branch(tmp_path, temporal_agg):
    stats = {"qpos_mean": np.zeros(S), "qpos_std": np.ones(S), "action_mean": np.zeros(A), "action_std": np.ones(A),
             "action_min": np.full(A, -2.0), "action_max": np.full(A, 2.0)}
    pol, tr = StubDiffusion(), []

    def boom(E):
        raise AssertionError("the Diffusion branch never ensembles (imitate_episodes.py:417-423)")

    IE.eval_bc(_diff_config(tmp_path, temporal_agg), "policy_last.ckpt", num_rollouts=2, policy=pol, ensemble_factory=boom,
               env_factory=lambda pose, idx: SyntheticEnv(CAMS, pose, height=40, width=60, seed=idx), stats=stats, verbose=False,
               trace=tr)
    # queried every step with temporal_agg (query_frequency 1, first action of each fresh chunk), else once per chunk
    assert pol.n == (T if temporal_agg else (T + Q - 1) // Q)
    raws = [raw[0, 0] for _, _, raw in tr]
    if temporal_agg:
        assert np.allclose(raws, [float(i + 1) for i in range(T)])
    else:
        assert np.allclose(raws, [float(t // Q + 1) + 0.1 * (t % Q) for t in range(T)])
    # the policy saw f32 [E, cams, 3, H, W] in [0, 1]: the centre-cropped, resized frames (reference :374, :214-224)
    im = pol.images[0]
    assert im.dtype == torch.float32 and tuple(im.shape) == (2, len(CAMS), 3, 40, 60)
    assert 0.0 <= float(im.min()) and float(im.max()) <= 1.0


class StubTrainPolicy:
    """CPU double of ACTPolicy for train_bc's control flow"""
    saved = []

    def __init__(self):
        self.w = torch.zeros(3)
        self.model = type("M", (), {"check_flags": lambda self: None, "device": "cpu"})()

    def __call__(self, qpos, image, actions=None, is_pad=None):
        class L(torch.Tensor):
            def backward(self_inner):
                pass
        loss = torch.tensor(float(1.0 / (1.0 + self.w[0]))).as_subclass(L)
        return {"l1": loss, "kl": torch.tensor(0.0), "loss": loss}

    def cuda(self): return self
    def eval(self): return self
    def train(self, mode=True): return self
    def serialize(self): return {"model.w": self.w.clone()}
    def deserialize(self, sd): self.w = sd["model.w"].clone(); return "<All keys matched successfully>"

    def configure_optimizers(self):
        pol = self

        class O:
            def zero_grad(self): pass
            def step(self): pol.w += 1.0
        return O()


def test_train_bc_in_loop_eval_and_pretrain_load(tmp_path, monkeypatch):
    evals = []
    monkeypatch.setattr(IE, "make_policy", lambda pc, cfg, device=None: StubTrainPolicy())
    monkeypatch.setattr(IE, "eval_bc", lambda config, ckpt_name, save_episode=True, num_rollouts=50, **kw:
                        (evals.append((ckpt_name, num_rollouts, os.path.isfile(os.path.join(config["ckpt_dir"], ckpt_name)))) or (0.5, 1.0)))
    import actmi.data as D
    monkeypatch.setattr(D, "DevicePrefetcher", lambda it: it)
    pre = tmp_path / "pre.ckpt"
    torch.save({"model.w": torch.full((3,), 7.0)}, pre)
    data = [(torch.zeros(1), torch.zeros(1), torch.zeros(1), torch.zeros(1))]
    cfg = {"num_steps": 6, "ckpt_dir": str(tmp_path / "ck"), "seed": 3, "policy_class": "ACT", "policy_config": {"seed": 0},
           "eval_every": 2, "validate_every": 3, "save_every": 100, "load_pretrain": True, "pretrain_ckpt_path": str(pre)}
    logged = []
    best = IE.train_bc(data, data, cfg, log=lambda d, step: logged.append((step, d)))
    # reference :590-596: at step 2, 4, 6 -- first save policy_step_{step}_seed_{seed}.ckpt, then eval_bc on it, 10 rollouts
    assert evals == [(f"policy_step_{s}_seed_3.ckpt", 10, True) for s in (2, 4, 6)]
    assert [(s, d) for s, d in logged if "success" in d] == [(s, {"success": 0.5}) for s in (2, 4, 6)]
    # the pretrain checkpoint was loaded before training (weights start at 7, :548-550)
    assert float(torch.load(tmp_path / "ck" / "policy_step_2_seed_3.ckpt", weights_only=True)["model.w"][0]) == 9.0
    assert best[0] in (0, 3, 6)
    cfg2 = dict(cfg, pretrain_ckpt_path=str(tmp_path / "missing.ckpt"))
    with pytest.raises(FileNotFoundError):
        IE.train_bc(data, data, cfg2)
    evals.clear()
    IE.train_bc(data, data, dict(cfg, eval_every=0, load_pretrain=False))          # 0 disables (the reference would divide by zero)
    assert evals == []
