"""Per-kernel time of one DiffusionPolicy query batch (library profiler)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import torch
from actmi import lib as L, weights as W
from actmi.diffusion import DiffusionNet, generate_diffusion_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cams = ["top", "left_wrist", "right_wrist"]
net = DiffusionNet(cams, prediction_horizon=32)
net.load_state_dict(generate_diffusion_state_dict(net.spec, 0))
img = torch.from_numpy(W.rand_u8(3, "dimg", (B, 3, 480, 640, 3))).cuda()
q = torch.zeros(B, 14, device="cuda"); noise = torch.randn(B, 32, 16, device="cuda")
net.forward_infer(q, img, noise=noise); torch.cuda.synchronize()
t0 = time.perf_counter(); cond = net.obs_cond(q, img); torch.cuda.synchronize(); t1 = time.perf_counter()
net.unet(noise, 45, cond); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"obs_cond {1e3*(t1-t0):.1f} ms, one unet pass {1e3*(t2-t1):.2f} ms (wall, incl. python)")
L.profile_enable(True)
net.forward_infer(q, img, noise=noise); torch.cuda.synchronize()
L.profile_enable(False)
prof = sorted(L.profile_report(), key=lambda p: -p["ms"])
tot = sum(p["ms"] for p in prof)
print(f"profiled kernel time {tot:.1f} ms")
for p in prof[:14]:
    print(f"  {p['name'][:70]:70s} x{p['count']:4d} {p['ms']:8.2f} ms")
