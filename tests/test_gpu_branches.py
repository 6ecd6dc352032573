"""The forward as two concurrent branches (camera halves in the ResNet trunk, batch halves in the transformer)
against the same forward issued as one branch: same kernels on disjoint ranges, so the outputs must agree to the last bit --
for odd camera counts and odd batches too, eagerly and as a replayed hipGraph, and for 3 / 4 branches."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from helpers import load_fixture, regenerate  # noqa: E402
from actmi import weights as W  # noqa: E402
from actmi.engine import ACTEngine  # noqa: E402


def _engine(cfg, sd_np, max_batch, env):
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        eng = ACTEngine(cfg, max_batch=max_batch)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    eng.load_state_dict(sd_np)
    eng.finalize()
    return eng


@pytest.mark.parametrize("name,batches", [("tiny", [1, 2, 5]), ("tiny_c3", [1, 3, 4]), ("full4", [2])])
def test_branches_bit_identical(name, batches):
    z, cfg = load_fixture(name)
    sd_np, _ = regenerate(z, cfg, with_actions=False)
    Bmax = max(batches)
    single = _engine(cfg, sd_np, Bmax, {"ACTMI_CAM_PIPE": "0"})
    engines = {"two": _engine(cfg, sd_np, Bmax, {"ACTMI_CAM_PIPE": "1"}),
               "four": _engine(cfg, sd_np, Bmax, {"ACTMI_CAM_PIPE": "1", "ACTMI_BRANCHES": "4"})}
    d = single.device
    for B in batches:
        inp = W.generate_inputs(cfg, B, seed=100 + B)
        qpos, img = torch.from_numpy(inp["qpos"]).to(d), torch.from_numpy(inp["image_u8"]).to(d)
        ref = single.forward_infer(qpos, img).clone()
        assert torch.isfinite(ref).all()
        for tag, eng in engines.items():
            got = eng.forward_infer(qpos, img)
            assert torch.equal(got, ref), f"{name} B={B} {tag}: max diff {float((got - ref).abs().max()):.3e}"
    # the replayed graph (branches = parallel branches of the graph) gives the same bits, twice in a row
    B = batches[-1]
    inp = W.generate_inputs(cfg, B, seed=100 + B)
    qpos, img = torch.from_numpy(inp["qpos"]).to(d), torch.from_numpy(inp["image_u8"]).to(d)
    ref = single.forward_infer(qpos, img).clone()
    eng = _engine(cfg, sd_np, B, {"ACTMI_CAM_PIPE": "1"})
    replay = eng.capture_infer(B)
    for _ in range(2):
        out = replay(qpos, img)
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    eng.check_flags()
