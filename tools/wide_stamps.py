"""Diagnostic (build with TQ_DEFS=-DTQ_WIDE_STAMPS): in-kernel time stamps of the root block's k_factor_w."""
import sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
f = P.random_clipping_qp()
g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
o = dict(f.opts); o["maxIter"] = 1
for _ in range(3):
    r = g.solve(**o)
buf = np.zeros(64, dtype=np.uint64)
capi.lib().tqgpu_get_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), len(buf))
cyc, wall = buf[0:32:2].astype(np.int64), buf[1:32:2].astype(np.int64)
names = ["start", "loaded", "p0 regs", "p0 done", "t0 done", "p1 regs", "p1 done", "t1 done", "p2 regs", "p2 done", "t2 done", "p3 regs", "p3 done", "t3 done", "factored", "end"]
for i, n in enumerate(names):
    print(f"{n:10s} cycles +{cyc[i]-cyc[0]:8d}  wall +{(wall[i]-wall[0])*10:7d} ns")
g.close()
