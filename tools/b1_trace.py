#!/usr/bin/env python3
"""A few graph replays of the B=1 policy query for `rocprofv3 --kernel-trace`: the timeline of the reference's own rollout mode
(imitate_episodes.py:390-399) -- which kernels sit on the critical path and where the device idles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "act-plus-plus_amd")):
    sys.path.insert(0, p)
import torch
from actmi import ops, weights as W
from actmi.config import ACTConfig
from actmi.engine import ACTEngine
dev = torch.device("cuda", 0)
cfg = ACTConfig(); B = int(os.environ.get("B", "1"))
eng = ACTEngine(cfg, max_batch=B, device=str(dev)); eng.load_state_dict(W.generate_state_dict(cfg, seed=0)); eng.finalize()
ens = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, dev)
replay = eng.capture_infer(B, with_ensemble=ens)
inp = W.generate_inputs(cfg, B, seed=1)
replay.static[0].copy_(torch.from_numpy(inp["qpos"])); replay.static[1].copy_(torch.from_numpy(inp["image_u8"]))
torch.cuda.synchronize(dev)
for i in range(int(os.environ.get("N", "12"))):
    replay(replay.static[0], replay.static[1])
torch.cuda.synchronize(dev)
print("done")
