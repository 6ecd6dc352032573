#!/usr/bin/env python3
"""DiffusionPolicy B=32, 3 cameras 480x640: eager vs graph replay, trunk vs UNet split (one line of JSON)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "act-plus-plus_amd")):
    sys.path.insert(0, p)
import torch
from actmi import weights as W
from actmi.diffusion import DiffusionNet, generate_diffusion_state_dict
dev = torch.device("cuda", 0)
cams = ["top", "left_wrist", "right_wrist"]
B = 32
net = DiffusionNet(cams, prediction_horizon=32, num_inference_timesteps=10, device=str(dev))
net.load_state_dict(generate_diffusion_state_dict(net.spec, seed=0))
img = torch.from_numpy(W.rand_u8(3, "dimg", (B, len(cams), 480, 640, 3))).to(dev)
qpos = torch.zeros((B, 14), device=dev)
noise = torch.randn((B, 32, 16), device=dev)
def timeit(fn, n):
    fn(); torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / n * 1e3
out = {}
out["eager_ms"] = timeit(lambda: net.forward_infer(qpos, img, noise=noise), 3)
out["trunk_ms"] = timeit(lambda: net.obs_cond(qpos, img), 3)
cond = net.obs_cond(qpos, img)
out["unet_pass_ms"] = timeit(lambda: net.unet(noise, 45, cond), 10)
rp = net.capture_infer(B, img)
out["graph_ms"] = timeit(lambda: rp(qpos, img, noise=noise), 9)
print(json.dumps(out))
