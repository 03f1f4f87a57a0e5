"""Pass period of the persistent kernel without in-kernel instrumentation: device time of a solve stopped after
k Newton iterations (maxIter = k), for trees of different depth.  t(2) - t(1) = one full pass (G + H, backward sweep,
forward sweep, trial sweep); fitting over the depths separates the cost of a tree level from the cost of a tier
boundary (hand-over between workgroups).  Usage: python tools/period.py [md] [reps]   (TREEQP_AMD_LIB selects a build)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P

md = int(sys.argv[1]) if len(sys.argv) > 1 else 2
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = []
for Nr in ([3, 4, 5, 6, 7, 8, 9, 10, 11] if md == 2 else [2, 3, 4, 5, 6]):
    p = P.linear_chain(md, Nr, Nr)
    nk = p.nk()
    nx = np.full(p.Nn, p.nx, dtype=np.int32)
    nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
    g = capi.TqGpu(nk, nx, nu).upload(qp.flat(), p.lambda0)
    if g.path != 2:
        print(f"Nr={Nr}: path {g.path}, skipped")
        g.close()
        continue
    t = {}
    for k in (1, 2, 3):
        for _ in range(20):
            r = g.solve(maxIter=k)
        for _ in range(reps):
            r = g.solve(maxIter=k)
        t[k] = float(np.median(g.device_times(reps))) * 1e6
        it = r["iter"]
    rows.append((Nr, p.Nn, t[1], t[2], t[3], it))
    print(f"Nr={Nr:2d} nodes={p.Nn:5d}  t(1)={t[1]:7.1f} us  t(2)={t[2]:7.1f}  t(3)={t[3]:7.1f}   pass = {t[2] - t[1]:6.2f} us  (t3-t2 = {t[3] - t[2]:6.2f}, iterations at maxIter=3: {it})", flush=True)
    g.close()
# fixed cost of a launch: a solve that is converged at its first termination test (tolerance 1e30): state load, first
# sweep, G + H, verdict, write-back
for Nr in ([3, 6, 9, 11] if md == 2 else [2, 4]):
    p = P.linear_chain(md, Nr, Nr)
    nk = p.nk()
    nx = np.full(p.Nn, p.nx, dtype=np.int32)
    nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
    g = capi.TqGpu(nk, nx, nu).upload(qp.flat(), p.lambda0)
    ink = []
    for _ in range(20 + reps):
        r = g.solve(stationarityTolerance=1e30)
        ink.append(r["device_time"])
    print(f"Nr={Nr:2d}: converged-at-once solve {float(np.median(g.device_times(reps))) * 1e6:6.1f} us by HIP events, {float(np.median(ink[20:])) * 1e6:6.1f} us in-kernel "
          f"(top workgroup: start -> verdict) (iter {r['iter']}, status {r['status']})")
    g.close()
