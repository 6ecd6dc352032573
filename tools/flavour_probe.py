#!/usr/bin/env python3
"""The same row-major GEMM shapes through the hot (unmasked) loop flavour and through the general one (an operand pre-scale of 1.0
forces it): what the training step's dynamic operand scales cost in the main loop.  One JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "act-plus-plus_amd")):
    sys.path.insert(0, p)
import torch
from actmi import ops
d = "cuda:0"
out = {}
for name, (M, N, K) in {"ffn1_b64": (76928, 3200, 512), "outproj_b64": (76928, 512, 512), "ffn2_b64": (76928, 512, 3200),
                        "ffn1_b8": (9616, 3200, 512)}.items():
    A = torch.randn(M, K, device=d); W = torch.randn(N, K, device=d) * 0.05
    W16 = ops.split16(W, 256.0)
    C = torch.empty(M, N, device=d)
    one = torch.ones(1, device=d)
    def run(**kw):
        for _ in range(3): ops.gemm(A, W16, out=C, prec="f16x3", w_split=256.0, **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): ops.gemm(A, W16, out=C, prec="f16x3", w_split=256.0, **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        return {"us": dt * 1e6, "tflops": 2.0 * M * N * K / dt / 1e12}
    out[name] = {"hot": run(), "general_a_scale_1": run(a_scale=1.0), "general_a_scale_dev": run(a_scale_dev=one)}
    Wf = W.clone()
    def run2(**kw):
        for _ in range(3): ops.gemm(A, Wf, out=C, prec="f16x3", **kw)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): ops.gemm(A, Wf, out=C, prec="f16x3", **kw)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        return {"us": dt * 1e6, "tflops": 2.0 * M * N * K / dt / 1e12}
    out[name]["unsplit_weights_b_scale_256"] = run2(b_scale=256.0)
print(json.dumps(out))
