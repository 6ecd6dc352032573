#!/usr/bin/env python3
"""Attention micro-benchmark on the ACT shapes (B=8), both precisions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import torch
from actmi import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for (Nq, Nk, shared, tag) in [(1202, 1202, False, "encoder self-attn"), (100, 1202, True, "decoder cross-attn")]:
    q = torch.randn((Nq, 512) if shared else (B, Nq, 512), device=dev)
    kv = torch.randn(B, Nk, 1024, device=dev)
    for prec in ("f32", "f16x3"):
        for split in (True, False):
            ms = timeit(lambda: ops.attention(q, kv[..., :512], kv[..., 512:], 8, q_shared=shared, prec=prec, split=split))
            fl = 4.0 * B * 8 * Nq * Nk * 64
            print(f"{tag:20s} {prec:6s} split-KV={split!s:5s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")
