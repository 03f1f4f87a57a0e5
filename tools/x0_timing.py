"""Diagnostic: x0-eliminated trees (the form every MPC caller solves) on the persistent path, incl. the heavy-backtracking spring-mass case."""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti
for name, p in (("spring_mass x0-eliminated xmax1=0.2", P.spring_mass(xmax1=0.2)), ("C2 x0-eliminated", P.linear_chain(2, 9, 9))):
    qp = product_qp_from_lti(capi, p, eliminate_x0=True)
    flat = qp.flat()
    g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
    for _ in range(3): r = g.solve()
    t0 = time.perf_counter(); n = 20
    for _ in range(n): r = g.solve()
    w = (time.perf_counter() - t0) / n
    print(name, "path", g.path, "iter", r["iter"], "ls", r["ls_total"], "launches", r["n_launches"], f"wall {w*1e6:.1f} us -> {r['iter']/w:.0f} it/s")
    g.close()
