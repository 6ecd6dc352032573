#!/usr/bin/env python3
"""Per-kernel averages of every counter found in the rocprofv3 --pmc passes under <dir>/p*/ (tools/conv_pmc.sh).
Prints, for the kernels that matter, counter value per launch and a few derived ratios."""
import csv, glob, os, re, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def short(n):
    n = re.sub(r"^void\s+", "", n).replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*\)$", "", n).strip().replace(" ", "")
    return n


tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(sys.argv[1], "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        c = r["Counter_Name"]
        tot[k][c] += float(r["Counter_Value"])
        cnt[k][c] += 1
want = [k for k in tot if re.search(r"gemm_f32_kernel<128,128|conv3x3_c64|attn_f16x3|conv1_f16x3|layernorm", k)]
for k in sorted(want):
    print("==", k)
    avg = {c: tot[k][c] / max(cnt[k][c], 1) for c in tot[k]}
    for c in sorted(avg):
        print(f"   {c:34s} {avg[c]:16.1f}   (launches {cnt[k][c]})")
    wc = avg.get("SQ_WAVE_CYCLES")
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS",
                  "SQ_ACTIVE_INST_VMEM", "SQ_INST_CYCLES_VMEM"):
            if c in avg:
                print(f"   ratio {c}/SQ_WAVE_CYCLES = {avg[c] / wc:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg and "SQ_BUSY_CYCLES" in avg:
        print(f"   ratio MFMA_BUSY/SQ_BUSY_CYCLES = {avg['SQ_VALU_MFMA_BUSY_CYCLES'] / avg['SQ_BUSY_CYCLES']:.3f}")
    if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
        print(f"   L2 hit rate = {avg['TCC_HIT_sum'] / (avg['TCC_HIT_sum'] + avg['TCC_MISS_sum'] + 1e-9):.3f}")
    if "SQ_LDS_BANK_CONFLICT" in avg and "SQ_LDS_IDX_ACTIVE" in avg:
        print(f"   LDS conflict share = {avg['SQ_LDS_BANK_CONFLICT'] / (avg['SQ_LDS_IDX_ACTIVE'] + 1e-9):.3f}")
