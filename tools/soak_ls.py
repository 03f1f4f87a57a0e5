"""Diagnostic: backtracking-heavy solves repeated many times on the persistent path (the in-kernel batched line search must give
the same verdict every time)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti
t0 = time.time()
p = P.spring_mass(xmax1=0.2)
qp = product_qp_from_lti(capi, p, eliminate_x0=True)
flat = qp.flat()
g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
r0 = g.solve(); s0 = g.solution()
bad = 0
for i in range(300):
    r = g.solve()
    bad += (r["status"], r["iter"], r["ls_total"]) != (r0["status"], r0["iter"], r0["ls_total"])
s1 = g.solution()
print(f"spring-mass x0-eliminated: {r0['iter']} iterations, {r0['ls_total']} trials, 300 repeats, bad {bad}, drift {max(float(np.max(np.abs(s0[k]-s1[k]))) for k in ('x','u','lam')):.1e}, {time.time()-t0:.1f} s", flush=True)
g.close()
for c in [(1, 8, 3, 2, 5, 5), (4, 4, 1, 3, 2, 6), (9, 4, 3, 2, 3, 9), (10, 8, 1, 2, 6, 6)]:
    f = P.random_uniform_tree_qp(*c)
    g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), None)
    r0 = g.solve(); s0 = g.solution(); bad = 0
    for i in range(500):
        r = g.solve()
        bad += (r["status"], r["iter"], r["ls_total"]) != (r0["status"], r0["iter"], r0["ls_total"])
    s1 = g.solution()
    print(f"{f.name}: {r0['iter']} iterations, {r0['ls_total']} trials, 500 repeats, bad {bad}, drift {max(float(np.max(np.abs(s0[k]-s1[k]))) for k in ('x','u','lam')):.1e}", flush=True)
    g.close()
