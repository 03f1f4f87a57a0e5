"""Diagnostic: more than 65535 launches on ONE mirror (the launch number in the hand-over tags is 16 bits wide and wraps)."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti
for name, p, reps in (("C1", P.spring_mass(), 70000), ("C2 small", P.linear_chain(2, 5, 5), 70000)):
    qp = product_qp_from_lti(capi, p)
    flat = qp.flat()
    g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
    r0 = g.solve(); s0 = g.solution()
    bad = 0
    t0 = time.time()
    for i in range(reps):
        r = g.solve()
        bad += (r["status"] != 0) or (r["iter"] != r0["iter"])
    s1 = g.solution()
    import numpy as np
    print(f"{name}: {reps} solves on one mirror, bad {bad}, solution drift {max(float(np.max(np.abs(s0[k]-s1[k]))) for k in ('x','u','lam')):.1e}, {time.time()-t0:.1f} s", flush=True)
    g.close()
