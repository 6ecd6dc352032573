#!/usr/bin/env python3
"""Block timeline of the GEMM kernel from in-kernel s_memtime stamps + HW_ID (diagnostic path only)."""
import os, sys, ctypes as C, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "act-plus-plus_amd"))
import numpy as np, torch
from actmi import lib as L
dev = torch.device("cuda:0")
def run(M, N, K, tag):
    A = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev); b = torch.randn(N, device=dev)
    out = torch.empty(M, N, device=dev)
    st = torch.zeros(8192 * 4, dtype=torch.int64, device=dev)
    d = L.GemmDesc()
    d.A, d.lda, d.Bw, d.ldb, d.bias, d.C, d.ldc = A.data_ptr(), K, W.data_ptr(), K, b.data_ptr(), out.data_ptr(), N
    d.M, d.N, d.K, d.groups = M, N, K, 1
    lib = L.load()
    for i in range(3):
        d.stamps = st.data_ptr() if i == 2 else None
        L.check(lib.actmi_op_gemm(C.byref(d), L.current_stream_ptr()), None, "gemm")
    torch.cuda.synchronize()
    s = st.view(-1, 4).cpu().numpy()
    nb = int((s[:, 3] > 0).sum()); s = s[:nb]
    t0 = s[:, 0].min()
    start, loop_end, end = s[:, 0] - t0, s[:, 2] - t0, s[:, 3] - t0
    hw = s[:, 1] & 0xFFFFFFFF; xcc = (s[:, 1] >> 32) & 0xF
    cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xF))
    print(f"== {tag}: {nb} blocks, span {end.max()} cycles; loop {np.mean(loop_end-start):.0f} epilogue {np.mean(end-loop_end):.0f}; distinct CUs {len(set(cu.tolist()))}")
    bycu = collections.defaultdict(list)
    for i in range(nb): bycu[int(cu[i])].append(i)
    for c, ids in list(bycu.items())[:3]:
        ids = sorted(ids, key=lambda i: start[i])
        print("  cu", hex(c), [(i, int(start[i]), int(loop_end[i]), int(end[i])) for i in ids[:5]])
    idle = []
    for c, ids in bycu.items():
        ev = sorted([(start[i], loop_end[i]) for i in ids])
        covered = 0; cur_s, cur_e = None, None
        for a, b_ in ev:
            if cur_e is None or a > cur_e:
                if cur_e is not None: covered += cur_e - cur_s
                cur_s, cur_e = a, b_
            else: cur_e = max(cur_e, b_)
        covered += cur_e - cur_s
        idle.append(1 - covered / end.max())
    print(f"  mean fraction of the kernel span with NO block in its main loop on a CU: {np.mean(idle):.3f}")
run(9616, 1536, 512, "qkv")
run(9616, 3200, 512, "linear1")
run(4096, 4096, 4096, "square")
