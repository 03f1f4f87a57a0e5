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
for o in (dict(maxIter=1, lineSearchMaxIter=1), dict(maxIter=1, lineSearchMaxIter=1, regTol=1e-4), dict(maxIter=1, lineSearchMaxIter=1, regTol=1e-8)):
    out = {}
    for path in ("auto", "generic"):
        os.environ["TREEQP_AMD_PATH"] = path
        g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, lam0)
        os.environ.pop("TREEQP_AMD_PATH")
        r = g.solve(**o); out[path] = g.solution(); g.close()
    da, dg = out["auto"]["dlam"], out["generic"]["dlam"]
    nb = len(da) // 16
    diff = np.abs(da - dg).reshape(nb, 16).max(axis=1)
    mag = np.abs(dg).reshape(nb, 16).max(axis=1)
    bad = np.where(diff > 1e-6 * np.maximum(1.0, mag))[0]
    print(o, "blocks", nb, "differing", len(bad), "first", bad[:20], "levels", sorted(set(int(np.floor(np.log2(b + 1))) for b in bad)))
    for b in bad[:3]:
        print("   block", b, "generic", np.array2string(dg[16*b:16*b+16], precision=3), "\n        persistent", np.array2string(da[16*b:16*b+16], precision=3))
