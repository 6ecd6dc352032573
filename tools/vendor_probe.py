#!/usr/bin/env python3
"""Run the vendor fp32 GEMM (torch.addmm) on the ACT shapes so that rocprofv3 shows which library kernel it picks."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device("cuda:0")
for (M, N, K) in [(9616, 1536, 512), (9616, 3200, 512), (9616, 512, 3200), (4096, 4096, 4096)]:
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(5): torch.addmm(b, A, W.t(), out=out)
torch.cuda.synchronize()
