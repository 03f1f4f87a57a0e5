import sys, os
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle.oracle_py as orc
from helpers import oracle_flat_from_lti
p = P.linear_chain(2, 8, 8, ubound=0.2)
flat = oracle_flat_from_lti(orc, p)
rng = np.random.Generator(np.random.PCG64(3))
lam0 = 30.0 * rng.standard_normal(len(p.lambda0))
for o in (dict(maxIter=1), dict(maxIter=1, lineSearchMaxIter=1), dict(maxIter=1, regType=1, regValue=1e-6), dict(maxIter=1, regType=0)):
    ref = orc.solve(flat, orc.default_opts(**o), lam0)
    out = {}
    for path in ("auto", "generic"):
        os.environ["TREEQP_AMD_PATH"] = path
        g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, lam0)
        os.environ.pop("TREEQP_AMD_PATH")
        r = g.solve(**o); sol = g.solution()
        out[path] = (r, sol, g.path)
        g.close()
    def e(a, b, k): return float(np.max(np.abs(a[k] - b[k])) / max(1.0, float(np.max(np.abs(b[k])))))
    ra, sa, pa = out["auto"]; rg, sg, pg = out["generic"]
    print(o, "ref", ref["status"], ref["iter"], ref["ls_total"], ref["n_regularized"], "| persistent", pa, ra["status"], ra["iter"], ra["ls_total"], "| generic", pg, rg["status"], rg["iter"], rg["ls_total"])
    for k in ("x", "u", "lam"):
        print(f"    {k}: persistent-ref {e(sa, ref, k):.2e} generic-ref {e(sg, ref, k):.2e} persistent-generic {e(sa, sg, k):.2e}")
