import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import torch
from actmi import ops
dev = torch.device("cuda:0")
def lin(M, N, K):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(5): ops.gemm(A, W, bias=b, out=out)
    torch.cuda.synchronize()
lin(4096, 4096, 4096)
lin(9616, 1536, 512)
lin(9616, 512, 512)
