#!/usr/bin/env python3
"""Is the B=8 step better as two INDEPENDENT half-batch steps on two streams (free to drift out of phase: one half's HBM-bound
trunk beside the other half's transformer) than as one graph whose two branches run the same phase side by side?  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "act-plus-plus_amd")):
    sys.path.insert(0, p)
import torch
from actmi import ops, weights as W
from actmi.config import ACTConfig
from actmi.engine import ACTEngine
dev = torch.device("cuda", 0)
cfg = ACTConfig()
sd = W.generate_state_dict(cfg, seed=0)

def rig(B, cam_pipe):
    os.environ["ACTMI_CAM_PIPE"] = "1" if cam_pipe else "0"
    eng = ACTEngine(cfg, max_batch=B, device=str(dev)); eng.load_state_dict(sd); eng.finalize()
    ens = ops.TemporalEnsemble(B, cfg.num_queries, cfg.action_dim, 0.01, dev)
    g = eng.capture_infer(B, with_ensemble=ens)
    inp = W.generate_inputs(cfg, B, seed=1)
    g.static[0].copy_(torch.from_numpy(inp["qpos"])); g.static[1].copy_(torch.from_numpy(inp["image_u8"]))
    return eng, g

N = 80
def timeit(fn):
    fn(5); torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); fn(N); torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / N * 1e3

out = {}
e8, g8 = rig(8, True)
out["one_graph_two_branches_b8_ms"] = timeit(lambda n: [g8(g8.static[0], g8.static[1]) for _ in range(n)])
e8s, g8s = rig(8, False)
out["one_graph_one_branch_b8_ms"] = timeit(lambda n: [g8s(g8s.static[0], g8s.static[1]) for _ in range(n)])
s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for name, pipe in (("two_streams_of_b4_one_branch_each_ms", False), ("two_streams_of_b4_two_branches_each_ms", True)):
    ea, ga = rig(4, pipe); eb, gb = rig(4, pipe)
    def both(n, stagger=False):
        cur = torch.cuda.current_stream(dev)
        s1.wait_stream(cur); s2.wait_stream(cur)
        for i in range(n):
            with torch.cuda.stream(s1): ga(ga.static[0], ga.static[1])
            with torch.cuda.stream(s2): gb(gb.static[0], gb.static[1])
        cur.wait_stream(s1); cur.wait_stream(s2)
    out[name] = timeit(both)
    del ea, ga, eb, gb
print(json.dumps(out))
