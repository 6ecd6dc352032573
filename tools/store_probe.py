#!/usr/bin/env python3
"""How fast can the chip take plain output stores?  (orientation for the GEMM epilogue: a 9616x3200 fp32 tile set)"""
import torch, time
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
for mb in (32, 123, 492):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    ms = timeit(lambda: x.fill_(1.0))
    ms2 = timeit(lambda: y.copy_(x))
    ms3 = timeit(lambda: torch.add(x, 1.0, out=y))
    print(f"{mb:4d} MB: fill {ms*1e3:7.1f} us = {mb/1024/ms*1e3/1e3:5.2f} TB/s written | copy {ms2*1e3:7.1f} us = {2*mb/1024/ms2:5.2f} TB/s r+w | add {ms3*1e3:7.1f} us")
