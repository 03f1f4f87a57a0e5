"""Diagnostic: phase times of the single-workgroup kernel on a C5-class pruned tree (TREEQP_AMD_STAMPS=1)."""
import sys, os, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
f = P.pruned_chain_qp()
g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
for _ in range(5):
    r = g.solve(**f.opts)
print("path", g.path, r["status"], r["iter"], r["ls_total"], f"{r['device_time']*1e6:.1f} us")
buf = np.zeros(12, dtype=np.uint64)
capi.lib().tqgpu_get_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 12)
names = ["init + first sweep", "G grad + termination", "H hessian", "F backward", "F forward", "L line search"]
for i, n in enumerate(names):
    print(f"  {n:24s} {int(buf[2 * i]) * 0.01:8.2f} us")
