"""Identity of the kernel sources a measurement was taken on (no git on the GPU box: a content hash instead)."""
import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(os.path.dirname(_HERE), "csrc")
_HEADER = os.path.join(os.path.dirname(os.path.dirname(_HERE)), "include", "actmi.h")


def kernel_source_sha16() -> str:
    """sha256 (first 16 hex digits) over csrc/*.hip, csrc/*.h and include/actmi.h in name order: two runs with the same value
    ran the same kernels.  profiles/r0N_traffic.json records it; bench.py compares it with the live tree (traffic_stale)."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_CSRC, "*.hip")) + glob.glob(os.path.join(_CSRC, "*.h"))) + [_HEADER]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]
