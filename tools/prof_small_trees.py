"""Diagnostic: phase times of the single-workgroup kernel (g_persist, TREEQP_AMD_STAMPS=1) on the small single trees of
tools/single_trees.py."""
import os, sys, ctypes as C
from pathlib import Path
import numpy as np
os.environ["TREEQP_AMD_STAMPS"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
cases = [("thesis example", P.thesis_example()), ("irregular clipping", P.irregular_clipping_qp()), ("random shape seed 5", P.random_shape_qp(5)),
         ("spring mass C1 (single workgroup)", None)]
names = ["init + first sweep", "G grad + termination", "H hessian", "F backward", "F forward", "L line search"]
for label, f in cases:
    if f is None:
        continue
    g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
    o = dict(f.opts) if getattr(f, "opts", None) else {}
    for _ in range(5):
        r = g.solve(**o)
    print(f"{label}: {len(f.nk)} nodes, path {g.path}; status {r['status']}, {r['iter']} iterations, {r['ls_total']} trials, {r['device_time']*1e6:.1f} us")
    if g.path != 3:
        g.close(); continue
    buf = np.zeros(12, dtype=np.uint64)
    capi.lib().tqgpu_get_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 12)
    for i, n in enumerate(names):
        print(f"  {n:24s} {int(buf[2 * i]) * 0.01:9.1f} us   ({int(buf[2 * i]) * 0.01 / max(r['iter'], 1):7.1f} per iteration)")
    g.close()
