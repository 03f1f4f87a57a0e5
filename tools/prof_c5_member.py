"""Diagnostic: phase times of the single-workgroup kernel (g_persist) on the slowest tree of BASELINE config C5, run as a member of a
small batch launch (TREEQP_AMD_STAMPS=1; alone such a tree goes to the three-launch family)."""
import os, sys, ctypes as C
from pathlib import Path
import numpy as np
os.environ["TREEQP_AMD_STAMPS"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
seed = 7 + (int(sys.argv[1]) if len(sys.argv) > 1 else 38)
f = P.pruned_chain_qp(seed=seed)
ms = [capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0) for _ in range(2)]
for _ in range(3):
    rs = capi.solve_batch(ms, **f.opts)
r = rs[0]
nk = np.asarray(f.nk)
print(f"{len(nk)} nodes, {int((nk > 0).sum())} blocks, paths {[m.path for m in ms]}; status {r['status']}, {r['iter']} iterations, {r['ls_total']} trials, {r['device_time']*1e6:.0f} us")
buf = np.zeros(12, dtype=np.uint64)
capi.lib().tqgpu_get_stamps(ms[0].h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 12)
names = ["init + first sweep", "G grad + termination", "H hessian", "F backward", "F forward", "L line search"]
for i, n in enumerate(names):
    print(f"  {n:24s} {int(buf[2 * i]) * 0.01:9.1f} us   ({int(buf[2 * i]) * 0.01 / max(r['iter'], 1):7.1f} per iteration)")
