"""Diagnostic (build with -DTQ_WIDE_STAMPS -DTQ_STAMP_BLOCK=-1): when every block of k_hf_w started, had its children's records, posted
its own and ended -- per tree level, microseconds after the first workgroup's start (C4)."""
import sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
f = P.random_clipping_qp()
g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
o = dict(f.opts)
for _ in range(5):
    r = g.solve(**o)
Np = int(np.sum(np.asarray(f.nk) > 0))
buf = np.zeros(4 * Np, dtype=np.uint64)
L = capi.lib()
L.tqgpu_debug_block_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
n = L.tqgpu_debug_block_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), Np)
t = buf.reshape(Np, 4).astype(np.int64)
t0 = t[:, 0].min()
us = (t - t0) / 100.0
nk = np.asarray(f.nk)
first, w, lvl = 0, 1, 0
print("k_sgp: start | C staged | stage done | gradient done" if __import__("os").environ.get("TQ_STAMPS_OF_SGP") else "k_hf_w: start | records in | posted | end", "-- min/max per level, us after the first workgroup started")
while first < Np:
    sl = us[first:first + w]
    print(f"{lvl:5d} {w:6d} | {sl[:,0].min():7.1f} {sl[:,0].max():7.1f} | {sl[:,1].min():7.1f} {sl[:,1].max():7.1f} | {sl[:,2].min():7.1f} {sl[:,2].max():7.1f} | {sl[:,3].min():7.1f} {sl[:,3].max():7.1f}   own work (records in -> posted) median {np.median(sl[:,2]-sl[:,1]):5.1f}, lifetime median {np.median(sl[:,3]-sl[:,0]):5.1f}")
    first += w; w *= int(nk[0]); lvl += 1
g.close()
