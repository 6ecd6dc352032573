"""Microbenchmark of gemm16 against the f16x3 form of actmi_op_gemm on the ACT shapes of the benchmark batch (B = 8, C = 4).
python tools/gemm16_bench.py [iters]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import torch  # noqa: E402
from actmi import ops  # noqa: E402

D = "cuda:0"
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
HOT = os.environ.get("G16_HOT") == "1"              # diagnostics: zero row strides -> every operand row is one resident line


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3          # us


rows = []
# plain GEMMs of the encoder at M = 9616
for name, M, N, K in [("ffn1", 9616, 3200, 512), ("ffn2", 9616, 512, 3200), ("qkv", 9616, 1536, 512), ("out_proj", 9616, 512, 512),
                      ("kv_dec", 9616, 1024, 512), ("sq4096", 4096, 4096, 4096)]:
    A, W = torch.randn(M, K, device=D), torch.randn(N, K, device=D) * K ** -0.5
    b = torch.randn(N, device=D)
    A16, W16, W4 = ops.split16v2(A, 16.0), ops.split16v2(W, 256.0), ops.split16(W, 256.0)
    out16 = torch.empty(M, N, device=D)
    outf = torch.empty(M, N, device=D)
    r = {"name": name, "M": M, "N": N, "K": K, "gflop": 2.0 * M * N * K / 1e9}
    r["old_us"] = timeit(lambda: ops.gemm(A, W4, bias=b, prec="f16x3", w_split=256.0, out=outf))
    for bm in (128, 256, 512):
        r[f"g16_{bm}_us"] = timeit(lambda: ops.gemm16(A16, W16, alpha=1 / 4096.0, bias=b, out_scale=16.0, bm=bm, out=out16,
                                                      _ld_override=(0, 0) if HOT else None))
    rows.append(r)
# convolutions of layer2-4 (4 cameras as groups)
for name, B, H, W_, Cin, Cout, k, s, p in [("l2_3x3", 8, 60, 80, 128, 128, 3, 1, 1), ("l2_s2", 8, 120, 160, 64, 128, 3, 2, 1),
                                            ("l3_3x3", 8, 30, 40, 256, 256, 3, 1, 1), ("l4_3x3", 8, 15, 20, 512, 512, 3, 1, 1),
                                            ("l3_s2", 8, 60, 80, 128, 256, 3, 2, 1), ("l2_ds", 8, 120, 160, 64, 128, 1, 2, 0)]:
    G = 4
    x = torch.randn(G, B, H, W_, Cin, device=D)
    w = torch.randn(G, Cout, k, k, Cin, device=D) * (k * k * Cin) ** -0.5
    sc, bi = torch.rand(G, Cout, device=D) + 0.5, torch.randn(G, Cout, device=D)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W_ + 2 * p - k) // s + 1
    x16, w16, w4 = ops.split16v2(x, 16.0), ops.split16v2(w, 256.0), ops.split16(w, 256.0)
    out16 = torch.empty(G, B, Ho, Wo, Cout, device=D)
    r = {"name": name, "M": B * Ho * Wo, "N": Cout, "K": k * k * Cin, "groups": G, "gflop": 2.0 * G * B * Ho * Wo * Cout * k * k * Cin / 1e9}
    r["old_us"] = timeit(lambda: ops.conv2d_nhwc(x, w4, sc, bi, relu=True, stride=s, pad=p, prec="f16x3", w_split=256.0))
    for bm in (128, 256, 512):
        r[f"g16_{bm}_us"] = timeit(lambda: ops.gemm16(x16, w16, alpha=1 / 4096.0, scale=sc, bias=bi, relu=True, out_scale=16.0, bm=bm,
                                                      conv=dict(stride=s, pad=p), out=out16))
    rows.append(r)
for r in rows:
    for k in ("old_us", "g16_128_us", "g16_256_us", "g16_512_us"):
        r[k.replace("_us", "_tf")] = r["gflop"] / r[k] * 1e3 / 1e3
    print(f"{r['name']:9s} M={r['M']:6d} N={r['N']:5d} K={r['K']:5d}  old {r['old_us']:7.1f} us {r['old_tf']:6.1f} TF | "
          f"g16/128 {r['g16_128_us']:7.1f} us {r['g16_128_tf']:6.1f} TF | g16/256 {r['g16_256_us']:7.1f} us {r['g16_256_tf']:6.1f} TF | "
          f"g16/256x256 {r['g16_512_us']:7.1f} us {r['g16_512_tf']:6.1f} TF", flush=True)
json.dump(rows, open(os.path.join(ROOT, "gpurun_out", os.environ.get("G16_OUT", "gemm16_bench.json")), "w"), indent=1)
