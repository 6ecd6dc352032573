"""Host logic of the batched / sharded eval rollouts on CPU (no GPU): stub policy + the oracle's ensemble as test
doubles; world_size-2 ``gloo`` run must reproduce the single-process result exactly."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import imitate_episodes as IE
from actmi.envs import SyntheticEnv
from actmi.sim_utils import draw_episode_poses, sample_box_pose, sample_insertion_pose, set_seed, shard_range
from oracle.act_ref import TemporalEnsembleRef

Q, A, T, S = 4, 16, 12, 14
CAMS = ["top", "left_wrist"]


class StubPolicy:
    """Deterministic stand-in for ACTPolicy on CPU (the product policy needs the GPU and has no fallback)."""

    def __call__(self, qpos, image):
        E = qpos.shape[0]
        base = qpos.float().mean(dim=1, keepdim=True) + image.float().mean(dim=(1, 2, 3, 4)).view(E, 1) / 255.0
        steps = torch.arange(Q, dtype=torch.float32).view(1, Q, 1) * 0.01
        dims = torch.arange(A, dtype=torch.float32).view(1, 1, A) * 0.001
        return base.view(E, 1, 1) + steps + dims + 0.5


class OracleEnsemble:
    """E independent reference-semantics ensembles (full [T,T+Q,A] buffers as in imitate_episodes.py:339)."""

    def __init__(self, E):
        self.refs = [TemporalEnsembleRef(T, Q, A) for _ in range(E)]
        self.t = 0

    def step(self, all_actions):
        out = torch.cat([r.step(self.t, all_actions[e:e + 1])[0] for e, r in enumerate(self.refs)], dim=0)
        self.t += 1
        return out


def _config(tmp, temporal_agg=True):
    return {"ckpt_dir": str(tmp), "state_dim": S, "policy_class": "ACT", "policy_config": {"num_queries": Q, "action_dim": A},
            "camera_names": CAMS, "episode_len": T, "task_name": "sim_transfer_cube_scripted", "temporal_agg": temporal_agg}


def _env_factory(pose, idx):
    return SyntheticEnv(CAMS, pose, height=8, width=12, seed=idx)


def _run(tmp, num_rollouts, temporal_agg=True, max_parallel=None, trace=None):
    return IE.eval_bc(_config(tmp, temporal_agg), "policy_last.ckpt", num_rollouts=num_rollouts, policy=StubPolicy(),
                      ensemble_factory=OracleEnsemble, env_factory=_env_factory, max_parallel=max_parallel, verbose=False,
                      trace=trace)


def test_pose_draw_order_matches_reference_loop():
    # reference: set_seed(1000) once, then one sample_box_pose() per rollout (imitate_episodes.py:229,324-327)
    set_seed(1000)
    ref = [sample_box_pose() for _ in range(7)]
    got = draw_episode_poses("sim_transfer_cube_scripted", 7, 1000)
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    set_seed(1000)
    ref = [np.concatenate(sample_insertion_pose()) for _ in range(5)]
    got = draw_episode_poses("sim_insertion_scripted", 5, 1000)
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    assert got[0].shape == (14,)


def test_shard_range_partitions():
    for n in (0, 1, 7, 50, 400):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_batched_equals_sequential(tmp_path):
    """Lock-step batching must not change any episode: batch of 5 == five batches of 1 (reference order)."""
    tr_b, tr_s = [], []
    r_b = _run(tmp_path / "b", 5, max_parallel=5, trace=tr_b)
    r_s = _run(tmp_path / "s", 5, max_parallel=1, trace=tr_s)
    assert r_b == r_s
    a = {(i, t): row for ids, t, raw in tr_b for i, row in zip(ids, raw)}
    b = {(i, t): row for ids, t, raw in tr_s for i, row in zip(ids, raw)}
    assert a.keys() == b.keys()
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert next(iter(a.values())).dtype == np.float64          # raw_action is float64 as in the reference
    txt = (tmp_path / "b" / "result_policy_last.txt").read_text()
    assert "Success rate" in txt and "Reward >= 0: 5/5" in txt


def test_no_temporal_agg_queries_every_chunk(tmp_path):
    tr = []
    _run(tmp_path, 2, temporal_agg=False, trace=tr)
    assert len(tr) == T and tr[0][2].dtype == np.float32


def _worker(rank, world, port, tmp, n):
    os.environ.update({"RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_RANK": str(rank),
                       "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    for p in (os.path.dirname(os.path.abspath(__file__)), ):
        sys.path.insert(0, p)
    res = _run(os.path.join(tmp, "dist"), n)
    if rank == 0:
        np.save(os.path.join(tmp, "dist_result.npy"), np.array(res))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_gloo_world2_matches_single_process(tmp_path):
    n = 7                                           # ragged: 4 + 3 episodes
    single = _run(tmp_path / "single", n)
    port = 29600 + (os.getpid() % 200)
    mp.spawn(_worker, args=(2, port, str(tmp_path), n), nprocs=2, join=True)
    distres = np.load(tmp_path / "dist_result.npy")
    assert tuple(distres) == single
    a = (tmp_path / "single" / "result_policy_last.txt").read_text()
    b = (tmp_path / "dist" / "result_policy_last.txt").read_text()
    assert a == b                                   # identical per-episode returns / highest rewards
