#!/usr/bin/env python3
"""A few InferPipeline steps for `rocprofv3 --kernel-trace --memory-copy-trace`: where the next frame's copy sits on the timeline."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "act-plus-plus_amd")):
    sys.path.insert(0, p)
import torch
from actmi import ops, weights as W
from actmi.config import ACTConfig
from actmi.engine import ACTEngine, InferPipeline
dev = torch.device("cuda", 0)
cfg = ACTConfig(); B = 8
eng = ACTEngine(cfg, max_batch=B, device=str(dev)); eng.load_state_dict(W.generate_state_dict(cfg, seed=0)); eng.finalize()
ens = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, dev)
pipe = InferPipeline(eng, B, with_ensemble=ens)
inp = W.generate_inputs(cfg, B, seed=1)
qh = torch.from_numpy(inp["qpos"]).pin_memory(); ih = torch.from_numpy(inp["image_u8"]).pin_memory()
n = int(os.environ.get("N", "8"))
pipe.feed(qh, ih)
for i in range(n):
    pipe.step(next_inputs=(qh, ih) if i + 1 < n else None)
torch.cuda.synchronize(dev)
print("done")
