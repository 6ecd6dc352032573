#!/usr/bin/env python3
"""Where the with_h2d leg loses its 5-9 %: same graph back to back / two graphs alternating / alternating + H2D copies /
one graph + staged D2D.  One JSON line."""
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
pipe = InferPipeline(eng, B, with_ensemble=ens)
inp = W.generate_inputs(cfg, B, seed=1)
qh = torch.from_numpy(inp["qpos"]).pin_memory(); ih = torch.from_numpy(inp["image_u8"]).pin_memory()
for s in pipe.slots:
    s.static[0].copy_(qh); s.static[1].copy_(ih)
torch.cuda.synchronize(dev)
N = 60
def t(fn):
    fn(3); torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); fn(N); torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / N * 1e3
def same(n):
    s = pipe.slots[0]
    for _ in range(n): s(s.static[0], s.static[1])
def alt(n):
    for i in range(n):
        s = pipe.slots[i & 1]; s(s.static[0], s.static[1])
def piped(n):
    pipe.feed(qh, ih)
    for i in range(n):
        if i + 1 < n: pipe.feed(qh, ih)
        pipe.step()
stage = [torch.empty_like(pipe.slots[0].static[1]) for _ in range(2)]
cs = torch.cuda.Stream(device=dev); evc = [torch.cuda.Event() for _ in range(2)]; evd = [torch.cuda.Event() for _ in range(2)]
def staged(n):
    s = pipe.slots[0]; cur = torch.cuda.current_stream(dev)
    def feed(k, first):
        if not first: cs.wait_event(evd[k])
        with torch.cuda.stream(cs):
            stage[k].copy_(ih, non_blocking=True); evc[k].record(cs)
    feed(0, True)
    for i in range(n):
        k = i & 1
        if i + 1 < n: feed(k ^ 1, i < 1)
        cur.wait_event(evc[k])
        s.static[1].copy_(stage[k], non_blocking=True)      # D2D into the one graph's input
        evd[k].record(cur)
        s(s.static[0], s.static[1])
def copy_only(n):
    for i in range(n):
        stage[i & 1].copy_(ih, non_blocking=True)
out = {"same_graph_ms": t(same), "alternating_graphs_ms": t(alt), "pipelined_h2d_ms": t(piped), "staged_d2d_one_graph_ms": t(staged),
       "h2d_copy_alone_ms": t(copy_only)}
print(json.dumps(out))
