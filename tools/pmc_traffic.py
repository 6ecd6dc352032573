#!/usr/bin/env python3
"""Per-launch HBM traffic of every kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), with the gfx950
correction of MI355X_MICROARCH.md (FETCH_SIZE under-reports wide coalesced reads by exactly 2x; both counters are KiB).
Kernel names are rewritten to the short names bench.py's in-library profiler uses, so that bench.py can look the
dominant kernel up.  The output records what it was measured on: the content hash of the kernel sources
(actmi.buildinfo.kernel_source_sha16, computed where this script runs -- run it on the tree that produced the CSVs) and, when
given, the commit.  usage: pmc_traffic.py <fetch_counter_csv> <write_counter_csv> <out.json> [note] [commit]"""
import csv, json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "act-plus-plus_amd"))
from actmi.buildinfo import kernel_source_sha16
from collections import defaultdict


def short_name(n):
    n = re.sub(r"^void\s+", "", n)
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*\)$", "", n).strip()
    n = n.replace(" ", "")
    m = re.match(r"gemm_f32_kernel<(\d+),(\d+),(\d+),(\d+),(\d+),(\d+),(\d+),(\d+)((?:,\d+)*)>", n)
    if m:       # <BM,BN,WM,WN,AMODE,BMODE,PREC,BSPLIT[,UNMASKED]> -> the profiler's name
        g = m.groups()
        xs = ",xs" if g[8].split(",")[2:3] == ["1"] else ""          # trailing <..., UNMASKED, XS>
        return "gemm_%s_kernel<%s%s>" % ("f16x3" if g[6] == "1" else "f32", ",".join(g[:6]), xs)
    return n


def collect(path, counter):
    """per kernel AND per (kernel, workgroups): the grid size tells the launch shapes of one kernel apart (layer2 / 3 / 4)"""
    tot, cnt = defaultdict(float), defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            k = short_name(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"]); cnt[k] += 1
            wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            tot[(k, wgs)] += float(r["Counter_Value"]); cnt[(k, wgs)] += 1
    return tot, cnt


ft, fc = collect(sys.argv[1], "FETCH_SIZE")
wt, wc = collect(sys.argv[2], "WRITE_SIZE")
out = {"note": sys.argv[4] if len(sys.argv) > 4 else "", "commit": sys.argv[5] if len(sys.argv) > 5 else None,
       "kernel_source_sha16": kernel_source_sha16(), "kernels": {}}
out["per_shape"] = []
for k in ft:
    f = ft[k] / max(fc[k], 1) * 1024 * 2
    w = wt.get(k, 0.0) / max(wc.get(k, 0), 1) * 1024
    rec = {"launches": fc[k], "fetch_bytes_per_launch": f, "write_bytes_per_launch": w, "traffic_bytes_per_launch": f + w}
    if isinstance(k, tuple):
        if k[0].startswith("gemm_") or k[0].startswith("conv") or k[0].startswith("attn"):
            out["per_shape"].append(dict(rec, kernel=k[0], workgroups=k[1]))
    else:
        out["kernels"][k] = rec
out["per_shape"].sort(key=lambda r: -r["traffic_bytes_per_launch"] * r["launches"])
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("kernels:", len(out["kernels"]))
