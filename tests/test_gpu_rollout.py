"""Batched eval rollout on the GPU (ACTPolicy over libactmi + HIP ensemble) against a reference-style sequential
loop on CPU (oracle policy call with batch 1 + the reference's ensemble transcription), same SyntheticEnv."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import imitate_episodes as IE  # noqa: E402
from actmi.config import tiny_config  # noqa: E402
from actmi.envs import SyntheticEnv  # noqa: E402
from actmi.sim_utils import draw_episode_poses  # noqa: E402
from actmi import weights as W  # noqa: E402


def test_batched_rollout_matches_sequential_oracle(tmp_path):
    from oracle import act_ref as R
    from policy import ACTPolicy
    cfg = tiny_config()
    T, n = 10, 3
    pc = {"lr": 1e-5, "num_queries": cfg.num_queries, "kl_weight": 10, "hidden_dim": cfg.hidden_dim,
          "dim_feedforward": cfg.dim_feedforward, "lr_backbone": 1e-5, "backbone": "resnet18", "enc_layers": cfg.enc_layers,
          "dec_layers": cfg.dec_layers, "nheads": cfg.nheads, "camera_names": cfg.camera_names, "vq": False,
          "vq_class": None, "vq_dim": None, "action_dim": 16, "no_encoder": False, "state_dim": 14,
          "image_h": cfg.image_h, "image_w": cfg.image_w, "base_width": cfg.base_width, "max_batch": 4}
    config = {"ckpt_dir": str(tmp_path), "state_dim": 14, "policy_class": "ACT", "policy_config": pc,
              "camera_names": cfg.camera_names, "episode_len": T, "task_name": "sim_transfer_cube_scripted", "temporal_agg": True}
    policy = ACTPolicy(pc, init_seed=4)
    # the drop-in round-trips the reference's checkpoint format
    sd = policy.serialize()
    assert all(k.startswith("model.") for k in sd)
    torch.save(sd, tmp_path / "policy_last.ckpt")
    print(policy.deserialize(torch.load(tmp_path / "policy_last.ckpt", weights_only=True)))
    env_factory = lambda pose, idx: SyntheticEnv(cfg.camera_names, pose, cfg.image_h, cfg.image_w, seed=idx)  # noqa: E731
    trace = []
    res = IE.eval_bc(config, "policy_last.ckpt", num_rollouts=n, policy=policy, env_factory=env_factory, verbose=False,
                     trace=trace)
    assert 0.0 <= res[0] <= 1.0
    got = {(i, t): row for ids, t, raw in trace for i, row in zip(ids, raw)}
    # sequential reference-style loop, batch 1, CPU oracle
    poses = draw_episode_poses(config["task_name"], n, 1000)
    sdt = {k: v for k, v in sd.items()}
    worst = 0.0
    for i in range(n):
        env = env_factory(poses[i], i)
        ts = env.reset()
        ens = R.TemporalEnsembleRef(T, cfg.num_queries, 16)
        for t in range(T):
            qpos = torch.from_numpy(np.array(ts.observation["qpos"])).float().unsqueeze(0)
            img = np.stack([ts.observation["images"][c] for c in cfg.camera_names])[None]
            with torch.no_grad():
                all_actions = R.policy_call(sdt, cfg, qpos, R.get_image_from_u8(img))
            raw, _ = ens.step(t, all_actions)
            raw = raw.numpy()[0]
            worst = max(worst, float(np.abs(got[(i, t)] - raw).max()))
            assert got[(i, t)].dtype == np.float64
            ts = env.step(raw[:-2])
    print(f"rollout: max|raw_action - sequential oracle| = {worst:.3e}")
    assert worst <= 1e-4
