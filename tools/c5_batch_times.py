"""Diagnostic: BASELINE config C5 as ONE batch launch (one workgroup per tree): every tree's own clock (launch start to its verdict),
iterations and trials -- which trees the batch waits for."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
fs = [P.pruned_chain_qp(seed=7 + i) for i in range(n)]
ms = [capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0) for f in fs]
for _ in range(3):
    rs = capi.solve_batch(ms, **fs[0].opts)
t0 = time.perf_counter()
for _ in range(5):
    rs = capi.solve_batch(ms, **fs[0].opts)
wall = (time.perf_counter() - t0) / 5
a = np.array([(len(f.nk), r["iter"], r["ls_total"], r["device_time"] * 1e6) for f, r in zip(fs, rs)])
print(f"batch of {n}: {wall*1e3:.2f} ms per call; per-tree clock: max {a[:,3].max():.0f} us, mean {a[:,3].mean():.0f}, median {np.median(a[:,3]):.0f}; paths {sorted(set(m.path for m in ms))}")
X = np.stack([a[:, 1] * a[:, 0], (a[:, 2]) * a[:, 0], a[:, 1], np.ones(n)], axis=1)
coef, *_ = np.linalg.lstsq(X, a[:, 3], rcond=None)
print(f"fit: {coef[0]*1e3:.1f} ns per node and iteration + {coef[1]*1e3:.1f} ns per node and trial + {coef[2]:.1f} us per iteration + {coef[3]:.0f} us")
for k in np.argsort(-a[:, 3])[:8]:
    print(f"  tree {k:3d}: {int(a[k,0]):4d} nodes {int(a[k,1]):3d} iterations {int(a[k,2]):4d} trials {a[k,3]:8.0f} us")
for m in ms:
    m.close()

# where a call's time goes: the C entry point alone (prebuilt argument arrays) against the Python wrapper
import ctypes as C
ms = [capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0) for f in fs]
o = capi.GpuOpts(maxIter=100, termCondition=2, stationarityTolerance=1e-8, regType=2, regTol=1e-6, regValue=1e-6, lineSearchMaxIter=50, lineSearchGamma=0.1,
                 lineSearchBeta=0.6, lineSearchRestartTrigger=-1, profile=0, checkLastActiveSet=1)
for k, v in fs[0].opts.items():
    setattr(o, k, v)
for m in ms:
    m.event_timing(False)                 # (what bench.py's timed region does)
arr = (C.c_void_p * n)(*[m.h for m in ms])
res = (capi.GpuResult * n)()
L = capi.lib()
for _ in range(3):
    L.tqgpu_solve_batch(arr, n, C.byref(o), res)
t0 = time.perf_counter()
for _ in range(5):
    L.tqgpu_solve_batch(arr, n, C.byref(o), res)
tc = (time.perf_counter() - t0) / 5
print(f"C entry point alone: {tc*1e3:.2f} ms per call; slowest tree's own clock {max(r.device_time for r in res)*1e3:.2f} ms")
