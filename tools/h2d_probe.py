#!/usr/bin/env python3
"""Where the with_h2d leg loses time: the step alone / the step as trunk graph + transformer graph with the next frame's copy beside the transformer
(engine.InferPipeline) / the copy released beside the START of the step (the stem) / the copy alone.  One JSON line."""
import json, os, sys, time
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
plain = eng.capture_infer(B, with_ensemble=ens)              # the step alone
pipe = InferPipeline(eng, B, with_ensemble=ens)              # trunk graph | event | transformer graph, the next frame's H2D copy beside the latter
inp = W.generate_inputs(cfg, B, seed=1)
qh = torch.from_numpy(inp["qpos"]).pin_memory(); ih = torch.from_numpy(inp["image_u8"]).pin_memory()
plain.static[0].copy_(qh); plain.static[1].copy_(ih)
torch.cuda.synchronize(dev)
N = 60
def t(fn):
    fn(3); torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); fn(N); torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / N * 1e3
def same(n):
    for _ in range(n): plain(plain.static[0], plain.static[1])
def piped(n):
    pipe.feed(qh, ih)
    for i in range(n): pipe.step(next_inputs=(qh, ih) if i + 1 < n else None)
def split_nocopy(n):
    for i in range(n):
        (q, im, _), trunk, rest = pipe.slots[i & 1]
        trunk(q, im); rest(q, im)
stage = [torch.empty_like(plain.static[1]) for _ in range(2)]
cs = torch.cuda.Stream(device=dev); evc = [torch.cuda.Event() for _ in range(2)]; evd = [torch.cuda.Event() for _ in range(2)]
def beside_start(n):
    """the round's first pipeline: the copy of frame t + 1 on a copy stream, released when step t STARTS (beside the stem)"""
    cur = torch.cuda.current_stream(dev)
    for i in range(n):
        k = i & 1
        cs.wait_stream(cur)
        with torch.cuda.stream(cs):
            stage[k].copy_(ih, non_blocking=True); evc[k].record(cs)
        plain(plain.static[0], plain.static[1])
        cur.wait_event(evc[k])
def copy_only(n):
    for i in range(n):
        stage[i & 1].copy_(ih, non_blocking=True)
for st, _, _ in pipe.slots:
    st[0].copy_(qh); st[1].copy_(ih)
out = {"step_alone_ms": t(same), "two_graphs_no_copy_ms": t(split_nocopy), "pipelined_copy_beside_transformer_ms": t(piped), "copy_beside_step_start_ms": t(beside_start),
       "h2d_copy_alone_ms": t(copy_only), "copy_stream_trials_ms": pipe.copy_stream_trials,
       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES")}
print(json.dumps(out))
