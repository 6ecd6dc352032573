"""Environment side of the eval rollouts.

The reference steps dm_control/MuJoCo envs (sim_env.py:20-116) on the host; those packages are absent offline, so
throughput and plumbing runs use ``SyntheticEnv``: same interface (``reset()/step(action)`` returning a timestep
with ``.observation['qpos']``, ``.observation['images'][cam]`` HWC uint8 and ``.reward``; ``.task.max_reward``), a
deterministic function of (pose, step).  ``make_sim_env`` wraps the real simulator (``SimEnvAdapter``) whenever a module
with the reference's ``sim_env`` interface is importable.
"""
import importlib
import os
import threading
import types

import numpy as np


class _TimeStep:
    __slots__ = ("observation", "reward")

    def __init__(self, observation, reward):
        self.observation, self.reward = observation, reward


class SyntheticEnv:
    """Pseudo-dynamics: qpos follows the commanded target with a first-order lag; reward rises when the
    commanded trajectory stays close to a pose-dependent goal.  Frames are cheap deterministic patterns."""

    def __init__(self, camera_names, pose, height=480, width=640, state_dim=14, max_reward=4, seed=0):
        self.camera_names = list(camera_names)
        self.h, self.w, self.S = height, width, state_dim
        self.task = types.SimpleNamespace(max_reward=max_reward)
        self.pose = np.asarray(pose, dtype=np.float64)
        self.rng = np.random.default_rng(seed)
        base = self.rng.integers(0, 256, size=(len(self.camera_names), height, width, 3), dtype=np.uint8)
        self._frames = base
        self.t = 0
        self.qpos = np.zeros(state_dim)
        self.goal = np.resize(self.pose, state_dim) * 0.5

    def _obs(self):
        shift = self.t % 7
        images = {c: np.roll(self._frames[i], shift, axis=1) for i, c in enumerate(self.camera_names)}
        return {"qpos": self.qpos.copy(), "qvel": np.zeros(self.S), "images": images}

    def reset(self):
        self.t = 0
        self.qpos = np.resize(self.pose, self.S) * 0.1
        return _TimeStep(self._obs(), 0)

    def step(self, action):
        a = np.asarray(action, dtype=np.float64)[: self.S]
        self.qpos = 0.9 * self.qpos + 0.1 * a
        self.t += 1
        err = float(np.abs(self.qpos - self.goal).mean())
        reward = int(min(self.task.max_reward, max(0, self.task.max_reward - int(err * 4))))
        return _TimeStep(self._obs(), reward)


class SimEnvAdapter:
    """The reference's simulator behind the batched rollout loop.

    ``mod`` is a module with the reference's ``sim_env`` interface (sim_env.py:18-52): ``make_sim_env(task_name)`` returning a
    dm_control ``control.Environment`` and the module global ``BOX_POSE = [None]`` that the task's ``initialize_episode``
    reads during ``reset()``.  The reference sets ``BOX_POSE[0]`` right before ``env.reset()`` (imitate_episodes.py:324-329);
    here the pose was pre-drawn in the reference's RNG order and is installed under a lock, because the E envs of a batch are
    reset from host threads and ``BOX_POSE`` is one global per module."""

    _lock = threading.Lock()

    def __init__(self, mod, task_name, pose):
        self.mod = mod
        self.pose = np.asarray(pose, dtype=np.float64)
        self.env = mod.make_sim_env(task_name)
        self.task = self.env.task                       # .max_reward (imitate_episodes.py:316)

    def reset(self):
        with SimEnvAdapter._lock:
            self.mod.BOX_POSE[0] = self.pose
            return self.env.reset()

    def step(self, action):
        return self.env.step(action)


_warned_synthetic = [False]


def load_sim_env_module(name=None):
    """Import the user-supplied simulator module (default ``sim_env``, override with ACTMI_SIM_ENV_MODULE): the reference's
    own sim_env.py on sys.path works as is.  Returns None when it (or dm_control / mujoco underneath it) is not importable."""
    name = name or os.environ.get("ACTMI_SIM_ENV_MODULE", "sim_env")
    try:
        mod = importlib.import_module(name)
    except Exception:
        return None
    if not (hasattr(mod, "make_sim_env") and hasattr(mod, "BOX_POSE")):
        return None
    return mod


def make_sim_env(task_name, camera_names, pose, seed=0, height=480, width=640, synthetic=None):
    """reference sim_env.py:20-52 + the BOX_POSE hand-off of imitate_episodes.py:324-327 when a ``sim_env`` module is
    importable; otherwise (or with ``synthetic=True`` / ACTMI_SYNTHETIC_ENV=1) the ``SyntheticEnv`` stand-in, announced once:
    its rewards are pseudo-dynamics, so success rates from it are plumbing / throughput figures, not task results."""
    if synthetic is None:
        synthetic = os.environ.get("ACTMI_SYNTHETIC_ENV") == "1"
    mod = None if synthetic else load_sim_env_module()
    if mod is not None:
        return SimEnvAdapter(mod, task_name, pose)
    if not _warned_synthetic[0]:
        _warned_synthetic[0] = True
        why = "requested" if synthetic else "no importable sim_env module (dm_control / mujoco absent)"
        print(f"[actmi.envs] SyntheticEnv stand-in ({why}): rewards are pseudo-dynamics -- throughput / plumbing only, "
              "NOT task success rates")
    return SyntheticEnv(camera_names, pose, height, width, seed=seed)


class SyntheticDataset:
    """Batches in the reference's training contract (utils.py:71-174, forward_pass imitate_episodes.py:529-532):
    image (u8 NHWC here; the f32 NCHW /255 form is produced on request), qpos, actions [B,Q,A] z-scored zero-padded,
    is_pad [B,Q]."""

    def __init__(self, cfg, batch_size, num_batches, seed=0, f32_images=False):
        self.cfg, self.B, self.n, self.seed, self.f32 = cfg, batch_size, num_batches, seed, f32_images

    def __len__(self):
        return self.n

    def __iter__(self):
        import torch
        from .weights import generate_inputs, u8_nhwc_to_f32_nchw
        for i in range(self.n):
            inp = generate_inputs(self.cfg, self.B, seed=self.seed * 100003 + i, with_actions=True)
            img = torch.from_numpy(u8_nhwc_to_f32_nchw(inp["image_u8"])) if self.f32 else torch.from_numpy(inp["image_u8"])
            yield img, torch.from_numpy(inp["qpos"]), torch.from_numpy(inp["actions"]), torch.from_numpy(inp["is_pad"])
