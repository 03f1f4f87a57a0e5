"""Diagnostic: smallest shapes on every path (one block, one level, Nr = Nh = 1, chains only)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle.oracle_py as orc
from helpers import oracle_flat_from_lti
cases = [("chain 2,1,1", lambda: P.linear_chain(2, 1, 1)), ("chain 2,1,2 (multistage)", lambda: P.linear_chain(2, 1, 2)), ("chain 2,1,9", lambda: P.linear_chain(2, 1, 9)),
         ("chain 2,2,2", lambda: P.linear_chain(2, 2, 2)), ("spring 3,1,1", lambda: P.spring_mass(md=3, Nr=1, Nh=1)), ("spring 3,1,2", lambda: P.spring_mass(md=3, Nr=1, Nh=2)),
         ("chain md=4 1,1", lambda: P.linear_chain(4, 1, 1, nm=2)), ("chain 2,3,20 long chains", lambda: P.linear_chain(2, 3, 20)), ("spring 3,2,30", lambda: P.spring_mass(md=3, Nr=2, Nh=30))]
bad = 0
for name, mk in cases:
    p = mk()
    flat = oracle_flat_from_lti(orc, p)
    ref = orc.solve(flat, lambda0=p.lambda0)
    g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
    r = g.solve(); sol = g.solution()
    err = max(float(np.max(np.abs(sol[k] - ref[k])) / max(1.0, float(np.max(np.abs(ref[k]))))) for k in ("x", "u", "lam"))
    ok = (r["status"], r["iter"], r["ls_total"]) == (ref["status"], ref["iter"], ref["ls_total"]) and err < 1e-8
    bad += not ok
    print(f"{name}: nodes {p.Nn} path {g.path} gpu {r['status']}/{r['iter']}/{r['ls_total']} ref {ref['status']}/{ref['iter']}/{ref['ls_total']} err {err:.1e} launches {r['n_launches']} {'OK' if ok else 'MISMATCH'}")
    g.close()
print("mismatches:", bad)
