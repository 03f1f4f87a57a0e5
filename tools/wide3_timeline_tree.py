"""Diagnostic (build with -DTQ_WIDE_STAMPS -DTQ_STAMP_BLOCK=-1): per-block time stamps of k_hf_w on an irregular tree -- for every
tree level: when its blocks had their children's records and when they posted their own; own work by block dimension."""
import sys, ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
f = P.pruned_chain_qp()
g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
o = dict(f.opts)
o["maxIter"] = 1
for _ in range(5):
    r = g.solve(**o)
nk = np.asarray(f.nk); nx = np.asarray(f.nx)
Nn = len(nk)
parents = [i for i in range(Nn) if nk[i] > 0]
Np = len(parents)
assert parents == list(range(Np)), "BFS numbering: parents first"
kid0 = np.concatenate([[1], 1 + np.cumsum(nk)[:-1]])
level = np.zeros(Nn, dtype=int); dad = np.zeros(Nn, dtype=int)
for i in range(Nn):
    for k in range(kid0[i], kid0[i] + nk[i]):
        level[k] = level[i] + 1; dad[k] = i
d = np.array([sum(nx[kid0[i]:kid0[i] + nk[i]]) for i in range(Np)])
buf = np.zeros(4 * Np, dtype=np.uint64)
L = capi.lib()
L.tqgpu_debug_block_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
L.tqgpu_debug_block_stamps(g.h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), Np)
t = buf.reshape(Np, 4).astype(np.int64)
us = (t - t[:, 0].min()) / 100.0
print(f"{Nn} nodes, {Np} blocks, {level.max()} levels below the root; block dimensions {sorted(set(d.tolist()))}")
import os
sgp = bool(os.environ.get("TQ_STAMPS_OF_SGP"))
print("k_sgp: level blocks | start | C staged | stage done | gradient done" if sgp else "level blocks | start | records in | posted | end", "  (min .. max, us after the first workgroup's start)")
for l in range(level[:Np].max(), -1, -1):
    s = us[level[:Np] == l]
    print(f"{l:5d} {len(s):6d} | {s[:,0].min():6.1f} {s[:,0].max():6.1f} | {s[:,1].min():6.1f} {s[:,1].max():6.1f} | {s[:,2].min():6.1f} {s[:,2].max():6.1f} | {s[:,3].min():6.1f} {s[:,3].max():6.1f}")
if sgp:
    print(f"last workgroup done {us[:, 3].max():.1f} us after the first one started; median lifetime {np.median(us[:, 3] - us[:, 0]):.1f} us, C staged after {np.median(us[:, 1] - us[:, 0]):.1f}, stage after {np.median(us[:, 2] - us[:, 0]):.1f}")
    g.close(); sys.exit(0)
own = us[:, 2] - us[:, 1]
for dd in sorted(set(d.tolist())):
    m = d == dd
    print(f"d = {dd:3d}: {m.sum():4d} blocks, records in -> posted median {np.median(own[m]):5.2f} us (min {own[m].min():5.2f}, max {own[m].max():5.2f})")
ho = [us[i, 1] - max(us[k, 2] for k in range(kid0[i], kid0[i] + nk[i]) if k < Np) for i in range(Np) if any(k < Np for k in range(kid0[i], kid0[i] + nk[i]))]
print(f"hand-over (last child's post -> records in): median {np.median(ho):5.2f} us, min {min(ho):5.2f}, max {max(ho):5.2f}")
g.close()
