"""engine.InferPipeline: double-buffered, host-fed policy queries (the copy of step t + 1 beside the transformer of step t) must give
exactly what one-at-a-time eager steps give on the same frames -- actions AND the temporal ensemble's running state."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from actmi import ops  # noqa: E402
from actmi import weights as W  # noqa: E402
from actmi.config import tiny_config  # noqa: E402
from actmi.engine import ACTEngine, InferPipeline  # noqa: E402


@pytest.mark.parametrize("B", [1, 3])
def test_pipelined_host_fed_steps_equal_sequential_eager_steps(B):
    cfg = tiny_config()
    eng = ACTEngine(cfg, max_batch=B)
    eng.load_state_dict(W.generate_state_dict(cfg, seed=4))
    eng.finalize()
    d = eng.device
    T = 7
    frames = [W.generate_inputs(cfg, B, seed=100 + t) for t in range(T)]
    # reference: eager, device-resident inputs, its own ensemble
    ens_e = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, d)
    exp = []
    for f in frames:
        a = eng.forward_infer(torch.from_numpy(f["qpos"]).to(d), torch.from_numpy(f["image_u8"]).to(d))
        exp.append((a.clone(), ens_e.step(a).clone()))
    # pipeline: pinned host frames; the copy of frame t + 1 is handed over with step t and runs beside its transformer
    host = [(torch.from_numpy(f["qpos"]).pin_memory(), torch.from_numpy(f["image_u8"]).pin_memory()) for f in frames]
    ens_p = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, d)
    pipe = InferPipeline(eng, B, with_ensemble=ens_p)
    pipe.feed(*host[0])
    for t in range(T):
        a_hat, raw = pipe.step(next_inputs=host[t + 1] if t + 1 < T else None)
        assert torch.equal(a_hat, exp[t][0]), f"step {t}: a_hat differs"
        assert torch.equal(raw, exp[t][1]), f"step {t}: ensembled action differs"
    with pytest.raises(RuntimeError, match="no inputs were fed"):
        pipe.step()
    # the engine is back to whole-step forwards after the phase captures
    f = frames[0]
    a = eng.forward_infer(torch.from_numpy(f["qpos"]).to(d), torch.from_numpy(f["image_u8"]).to(d))
    assert torch.equal(a, exp[0][0])
    eng.check_flags()
