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
    # the default above ran the three episodes as two ping-pong groups (2 + 1, the policy query of one beside the simulator
    # steps of the other); one lock-step group gives the same trajectories (other batch sizes pick other tile shapes: 3e-5)
    assert any(len(ids) < n for ids, _, _ in trace)
    trace1 = []
    res1 = IE.eval_bc(config, "policy_last.ckpt", num_rollouts=n, policy=policy, env_factory=env_factory, verbose=False,
                      trace=trace1, pipeline_groups=1)
    assert all(len(ids) == n for ids, _, _ in trace1) and res1 == res
    got1 = {(i, t): row for ids, t, raw in trace1 for i, row in zip(ids, raw)}
    assert got1.keys() == got.keys() and max(float(np.abs(got1[k] - got[k]).max()) for k in got) <= 3e-5


@pytest.mark.filterwarnings("ignore::DeprecationWarning")        # torch's own pin_memory helper
def test_training_loop_on_episode_files(tmp_path):
    """imitate_episodes.main (training branch) over episode files in the reference's key layout: load_data ->
    pinned staging -> DevicePrefetcher -> forward/backward/AdamW; checkpoints and dataset_stats.pkl are written."""
    import os
    import pickle
    rng = np.random.default_rng(0)
    data = tmp_path / "data"
    data.mkdir()
    cams = ["top", "left_wrist", "right_wrist"]                    # sim_transfer_cube_scripted (constants.py)
    for i in range(4):
        T = 5 + i
        ep = {"/observations/qpos": rng.standard_normal((T, 14)).astype(np.float32),
              "/observations/qvel": rng.standard_normal((T, 14)).astype(np.float32),
              "/action": rng.standard_normal((T, 16)).astype(np.float32), "attrs_sim": np.array(True)}
        for c in cams:
            ep[f"/observations/images/{c}"] = rng.integers(0, 256, (T, 480, 640, 3), dtype=np.uint8)
        np.savez(data / f"episode_{i}.npz", **ep)
    ck = tmp_path / "ck"
    args = dict(eval=False, ckpt_dir=str(ck), policy_class="ACT", task_name="sim_transfer_cube_scripted", batch_size=2,
                seed=0, num_steps=2, lr=1e-5, kl_weight=10, chunk_size=100, hidden_dim=512, dim_feedforward=3200,
                temporal_agg=False, eval_every=1000, validate_every=2, save_every=1000, dataset_dir=str(data),
                num_rollouts=1, max_batch=2)
    IE.main(args)
    assert os.path.isfile(ck / "policy_last.ckpt") and os.path.isfile(ck / "policy_best.ckpt")
    with open(ck / "dataset_stats.pkl", "rb") as f:
        stats = pickle.load(f)                       # written by this program a moment ago
    assert stats["action_mean"].shape == (16,) and stats["qpos_std"].shape == (14,)
    sd = torch.load(ck / "policy_last.ckpt", weights_only=True)
    assert all(torch.isfinite(v).all() for v in sd.values())
