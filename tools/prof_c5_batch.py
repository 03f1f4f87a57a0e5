"""Diagnostic: phase times inside g_persist_batch (one workgroup per tree) for a batch of C5-class trees (TREEQP_AMD_STAMPS=1)."""
import sys, os, ctypes as C, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("TREEQP_AMD_STAMPS", "1")
from treeqp_amd import capi, problems as P
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fs = [P.pruned_chain_qp(seed=7 + i) for i in range(n)]
ms = [capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0) for f in fs]
for _ in range(3):
    t0 = time.perf_counter()
    rs = capi.solve_batch(ms, **fs[0].opts)
    dt = time.perf_counter() - t0
print(f"{n} trees: {dt * 1e3:.2f} ms per batch; iterations {[r['iter'] for r in rs[:8]]} trials {[r['ls_total'] for r in rs[:8]]}")
names = ["init + first sweep", "G grad + termination", "H hessian", "F backward", "F forward", "L line search"]
for m_i in (0, 1):
    buf = np.zeros(12, dtype=np.uint64)
    capi.lib().tqgpu_get_stamps(ms[m_i].h, buf.ctypes.data_as(C.POINTER(C.c_ulonglong)), 12)
    print(f"member {m_i}: nodes {len(fs[m_i].nk)} path {ms[m_i].path}")
    for i, nm in enumerate(names):
        print(f"  {nm:24s} {int(buf[2 * i]) * 0.01:9.1f} us")
its = np.array([r["iter"] for r in rs]); lss = np.array([r["ls_total"] for r in rs]); nn = np.array([len(f.nk) for f in fs])
dts = np.array([r["device_time"] for r in rs]) * 1e3
order = np.argsort(-dts)[:8]
print("slowest members (ms, nodes, iterations, trials):", [(round(float(dts[i]), 2), int(nn[i]), int(its[i]), int(lss[i])) for i in order])
print(f"sum of iterations {its.sum()}, of trials {lss.sum()}, median member {np.median(dts):.2f} ms, max {dts.max():.2f} ms")
