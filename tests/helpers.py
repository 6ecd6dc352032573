"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

from actmi.config import ACTConfig
from actmi import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    cfg = ACTConfig(**json.loads(str(z["config_json"]))).validate()
    return z, cfg


def regenerate(z, cfg, with_actions=True):
    """Weights and inputs are regenerated from seeds; the fixture's hashes prove they are the same bytes."""
    import hashlib
    sd = W.generate_state_dict(cfg, int(z["seed_w"]))
    inp = W.generate_inputs(cfg, int(z["batch"]), int(z["seed_in"]), with_actions=with_actions)
    for k in z.files:
        if k.startswith("sha:"):
            name = k[4:]
            a = inp[name] if name in inp else sd[name]
            assert hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest() == str(z[k]), f"regenerated {name} differs"
    return sd, inp


def sample_like(a, z):
    m = int(z["sample_max_elems"])
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return W.fixture_sample(a, m) if m else a.reshape(-1)


def torch_sd(sd_np, prefix="model."):
    return {prefix + k: torch.from_numpy(v) for k, v in sd_np.items()}
