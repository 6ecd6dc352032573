import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import torch
from actmi import ops
dev = torch.device("cuda:0")
def bench(B, H, Nq, Nk, hd, shared=False):
    D = H * hd
    q = torch.randn((Nq, D) if shared else (B, Nq, D), device=dev)
    kv = torch.randn(B, Nk, 2 * D, device=dev)
    f = lambda: ops.attention(q, kv[..., :D], kv[..., D:], H, q_shared=shared)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"B={B} H={H} Nq={Nq} Nk={Nk} hd={hd}: {ms*1e3:8.1f} us  {4.0*B*H*Nq*Nk*hd/ms/1e9:6.1f} TF")
bench(8, 8, 1202, 1202, 64)
bench(8, 8, 100, 1202, 64, shared=True)
bench(50, 8, 1202, 1202, 64)
bench(1, 8, 1202, 1202, 64)
