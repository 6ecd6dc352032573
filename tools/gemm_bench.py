#!/usr/bin/env python3
"""Micro-benchmark of the fp32 MFMA GEMM / implicit-conv kernel on the shapes of the ACT forward (B=8)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import torch
from actmi import ops

dev = torch.device("cuda:0")
VENDOR = os.environ.get("GEMM_BENCH_VENDOR") == "1"
SPLITW = os.environ.get("GEMM_BENCH_SPLITW") == "1"     # weights pre-split (needs ACTMI_GEMM_PREC=f16x3)
torch.backends.cuda.matmul.allow_tf32 = False

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def lin(M, N, K, tag):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    if SPLITW: W = ops.split16(W)
    ms = timeit(lambda: ops.gemm(A, W, bias=b, out=out, w_split=SPLITW))
    line = f"{tag:28s} M={M:6d} N={N:5d} K={K:5d}  {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF"
    if VENDOR:      # the vendor library (rocBLAS / hipBLASLt behind torch.addmm), same shape, for orientation only
        ms2 = timeit(lambda: torch.addmm(b, A, W.t(), out=out))
        line += f"   | torch.addmm {ms2*1e3:8.1f} us {2*M*N*K/ms2/1e9:7.1f} TF"
    print(line)

def conv(G, B, H, W_, Cin, Cout, k, s, p, tag):
    x = torch.randn(G, B, H, W_, Cin, device=dev); w = torch.randn(G, Cout, k, k, Cin, device=dev)
    sc = torch.rand(G, Cout, device=dev); bi = torch.rand(G, Cout, device=dev)
    Ho, Wo = (H + 2*p - k)//s + 1, (W_ + 2*p - k)//s + 1
    if SPLITW: w = ops.split16(w)
    ms = timeit(lambda: ops.conv2d_nhwc(x, w, sc, bi, None, True, s, p, w_split=SPLITW))
    fl = 2.0 * G * B * Ho * Wo * Cout * k * k * Cin
    print(f"{tag:28s} M={B*Ho*Wo:6d} N={Cout:5d} K={k*k*Cin:5d}  {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TF")

if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    conv(4, B, 120, 160, 64, 64, 3, 1, 1, "layer1 3x3")
    conv(4, B, 120, 160, 64, 128, 3, 2, 1, "layer2.0 3x3 s2")
    conv(4, B, 60, 80, 128, 128, 3, 1, 1, "layer2 3x3")
    conv(4, B, 30, 40, 256, 256, 3, 1, 1, "layer3 3x3")
    conv(4, B, 15, 20, 512, 512, 3, 1, 1, "layer4 3x3")
    lin(B * 1202, 1536, 512, "enc qkv")
    lin(B * 1202, 512, 512, "enc out_proj")
    lin(B * 1202, 3200, 512, "enc linear1")
    lin(B * 1202, 512, 3200, "enc linear2")
    lin(B * 1202, 1024, 512, "dec kv")
    lin(B * 100, 3200, 512, "dec linear1")
    lin(4096, 4096, 4096, "square 4096")
