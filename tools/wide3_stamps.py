"""Diagnostic (build with -DTQ_WIDE_STAMPS): in-kernel time stamps of k_hf_w, root block and last block, C4."""
import sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
import os
f = P.pruned_chain_qp() if os.environ.get("TQ_STAMPS_CASE") == "pruned" else P.random_clipping_qp()      # (pruned: build with -DTQ_STAMP_BLOCK=<a block of interest>)
g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
o = dict(f.opts)
for _ in range(5):
    r = g.solve(**o)
buf = np.zeros(160, dtype=np.uint64)
capi.lib().tqgpu_get_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), len(buf))
names = ["start", "C in LDS", "image built", "records in", "p0 chain", "p0 ahead", "p1 chain", "p1 ahead", "p2 chain", "p2 ahead", "p3 chain", "-", "schur posted", "prepared", "-", "-"] + [f"p{p} {w}" for p in range(4) for w in ("loaded", "chained", "stored")]
for base, what in ((32, "last block (first workgroup)"), (0, "block TQ_STAMP_BLOCK (default: the root)")):
    cyc, wall = buf[2 * base:2 * base + 56:2].astype(np.int64), buf[2 * base + 1:2 * base + 57:2].astype(np.int64)
    print(what)
    order = sorted(range(len(names)), key=lambda i: cyc[i])
    for i in order:
        n = names[i]
        if cyc[i]:
            print(f"  {n:13s} cycles +{cyc[i]-cyc[0]:8d}  wall +{(wall[i]-wall[0])*10:7d} ns")
print("root start after last-block start: %d ns" % ((int(buf[1]) - int(buf[65])) * 10))
g.close()
