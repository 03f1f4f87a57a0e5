import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi as gpu, problems as P
import oracle_py as orc
from helpers import oracle_flat_from_lti
p = P.linear_chain(2, 6, 6, ubound=0.1)
flat = oracle_flat_from_lti(orc, p)
rng = np.random.Generator(np.random.PCG64(2))
lam0 = 10.0 * rng.standard_normal(len(p.lambda0))
for path in ("auto", "tiered"):
    os.environ["TREEQP_AMD_PATH"] = path
    g = gpu.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, lam0)
    for ls in (1, 2, 3, 5):
        r = g.solve(maxIter=1, lineSearchMaxIter=ls)
        sol = g.solution()
        print(path, ls, {k: r[k] for k in ("status", "iter", "ls_total", "n_launches", "last_fval")}, "|x|", float(np.abs(sol["x"]).sum()), "|lam|", float(np.abs(sol["lam"]).sum()), "|dlam|", float(np.abs(sol.get("dlam", np.zeros(1))).sum()))
    g.close()
