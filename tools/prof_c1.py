"""Diagnostic: run the C1 (spring-mass example tree) solve a few times (for rocprofv3 --kernel-trace --stats)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
p = P.spring_mass()
nk = p.nk()
nx = np.full(p.Nn, p.nx, dtype=np.int32)
nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
g = capi.TqGpu(nk, nx, nu).upload(qp.flat(), p.lambda0)
for _ in range(30):
    r = g.solve()
print(r)
import ctypes as C, os
if os.environ.get("TREEQP_AMD_STAMPS"):
    buf = np.zeros(12, dtype=np.uint64)
    capi.lib().tqgpu_get_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 12)
    names = ["init + first sweep", "G grad + termination", "H hessian", "F backward", "F forward", "L line search"]
    for i, n in enumerate(names):
        print(f"  {n:24s} {int(buf[2 * i]) * 0.01:8.2f} us")
