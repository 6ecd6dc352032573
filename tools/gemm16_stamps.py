"""Where one workgroup of gemm16 spends its cycles (diagnostic): shader-clock stamps of workgroup 0."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import torch
from actmi import ops
D = "cuda:0"
for name, M, N, K, bm in [("ffn1", 9616, 3200, 512, 256), ("ffn1", 9616, 3200, 512, 128), ("sq4096", 4096, 4096, 4096, 256)]:
    A, W = torch.randn(M, K, device=D), torch.randn(N, K, device=D) * K ** -0.5
    A16, W16 = ops.split16v2(A, 16.0), ops.split16v2(W, 256.0)
    out = torch.empty(M, N, device=D)
    st = torch.zeros(128, dtype=torch.int64, device=D)
    for _ in range(3):
        ops.gemm16(A16, W16, alpha=1 / 4096.0, out_scale=16.0, bm=bm, out=out, stamps=st)
    torch.cuda.synchronize()
    s = st.cpu().tolist()
    ticks = s[63]                       # 100 MHz real-time ticks between stamp[1] and the end
    last = max(i for i in range(2, 60) if s[i])
    clk = (s[last] - s[1]) / (ticks / 100.0) if ticks else 0.0     # MHz
    print(f"{name} bm={bm}: clock ~{clk:.0f} MHz; prologue {s[1]-s[0]} cyc")
    names = ["L start", "dma issued", "ds_read issued", "vmcnt done", "lgkm done", "barrier passed", "mfma issued", "vmcnt done"]
    for g in range(2):
        base = 64 + 32 * g
        t0 = s[64]                      # both groups relative to group 0's L(4) start
        print(f"   group {g} steps 4,5: " + " | ".join(f"{names[k % 8]} +{s[base + k] - t0}" for k in range(16)))
    prev = s[1]
    for i in range(2, min(last, 5) + 1):
        print(f"   {'kloop' if i % 2 == 0 else 'epilogue'} tile {(i-2)//2}: {s[i]-prev} cyc")
        prev = s[i]
