"""Diagnostic: the drop-in API (treeqp_tdunes_solve) per-call cost: staging (interface) vs solver time."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti

for name, p in (("C1", P.spring_mass()), ("C2", P.linear_chain(2, 9, 9))):
    qp = product_qp_from_lti(capi, p)
    s = capi.TdunesSolver(qp)
    s.set_dual_initialization(p.lambda0)
    for _ in range(3):
        s.set_dual_initialization(p.lambda0); st = s.solve()
    tot, sol, itf = [], [], []
    for _ in range(30):
        s.set_dual_initialization(p.lambda0)
        st = s.solve()
        i = qp.info
        tot.append(i["total_time"]); sol.append(i["solver_time"]); itf.append(i["interface_time"])
    print(f"{name}: status {st} iter {qp.info['iter']}: total {min(tot)*1e6:.1f} us = solver {min(sol)*1e6:.1f} + interface {min(itf)*1e6:.1f} us")
    s.destroy()
