"""Diagnostic: device time of a solve as a function of the iteration cap (fixed per-solve cost vs per-iteration cost)."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P

Nr = int(sys.argv[1]) if len(sys.argv) > 1 else 9
p = P.linear_chain(2, Nr, Nr)
nk = p.nk()
nx = np.full(p.Nn, p.nx, dtype=np.int32)
nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
g = capi.TqGpu(nk, nx, nu).upload(qp.flat(), p.lambda0)
print("path", g.path)
for mi in (1, 2, 3, 4, 10):
    ts = []
    for _ in range(30):
        r = g.solve(maxIter=mi)
        ts.append(r["device_time"])
    ts = np.array(ts[5:]) * 1e6
    print(f"maxIter={mi:3d}: iter={r['iter']} status={r['status']} device {np.median(ts):8.2f} us (min {ts.min():.2f})")
