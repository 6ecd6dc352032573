#!/usr/bin/env python3
"""Timing experiment for the direct layer1 convolution: which phase costs what (phases switched off via debug bits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import ctypes as C, torch
from actmi import ops, lib as L
dev = torch.device("cuda:0")
G, B, H, W = 4, 8, 120, 160
x = torch.randn(G, B, H, W, 64, device=dev); w = torch.randn(G, 64, 3, 3, 64, device=dev) / 24
sc = torch.ones(G, 64, device=dev); bi = torch.zeros(G, 64, device=dev); res = torch.randn_like(x); out = torch.empty_like(x)
w16 = ops.split16(w, 256.0)
lib = L.load()
def run(bits):
    rc = lib.actmi_op_conv3x3_c64(x.data_ptr(), w16.data_ptr(), 256.0, sc.data_ptr(), bi.data_ptr(), res.data_ptr(), out.data_ptr(),
                                  G, B, H, W, 1 | (bits << 8), L.current_stream_ptr())
    assert rc == 0
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
for bits, tag in [(0, "full"), (1, "no patch loads"), (2, "no tap loop"), (4, "no epilogue"), (8, "no weight prefetch beyond 4"), (7, "nothing but launch+staging writes"), (3, "no loads no taps"), (6, "patch only")]:
    print(f"{tag:36s} {timeit(lambda: run(bits)):8.1f} us")
