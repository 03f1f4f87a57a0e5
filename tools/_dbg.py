import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
import numpy as np
import oracle_py as orc
from treeqp_amd import capi, problems as P
for reg, tol in ((0, 1e-6), (1, 1e-6), (2, 1e-3)):
  for f in (P.pruned_chain_qp(), P.pruned_chain_qp(Nh=6, seed=5)):
    opts = dict(f.opts); opts.update(regType=reg, regValue=1e-6, regTol=tol)
    ref = orc.solve(f.as_dict(), orc.default_opts(**opts), lambda0=f.lambda0)
    out = [("oracle", ref["status"], ref["iter"], ref["ls_total"])]
    for name, env in (("auto", {}), ("w3", {"TREEQP_AMD_PATH": "generic"}), ("old", {"TREEQP_AMD_PATH": "generic", "TREEQP_AMD_NO_WIDE3": "1"})):
        for k, v in env.items(): os.environ[k] = v
        g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
        for k in env: os.environ.pop(k)
        r = g.solve(**opts)
        sol = g.solution()
        err = max(float(np.max(np.abs(sol[k] - ref[k]))) for k in ("x", "u", "lam"))
        out.append((name, g.path, r["status"], r["iter"], r["ls_total"], f"{err:.1e}"))
        g.close()
    print(reg, len(f.nk), out)
