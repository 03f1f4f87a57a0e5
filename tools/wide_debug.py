"""Diagnostic: the wide (workgroup-per-block) kernels on random unconstrained trees of several block sizes."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle.oracle_py as orc
cases = ((10, 4, 3, 3), (16, 4, 3, 3), (20, 10, 3, 3), (21, 5, 3, 3), (20, 10, 3, 5), (20, 10, 3, 6), (20, 10, 3, 7), (20, 10, 3, 8))
for nx, nu, md, lv in cases:
    f = P.random_clipping_qp(nx=nx, nu=nu, md=md, levels=lv)
    ref = orc.solve(f.as_dict(), orc.default_opts(**f.opts))
    g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
    its = [g.solve(**f.opts)['iter'] for _ in range(3)]
    r = g.solve(**f.opts)
    sol = g.solution()
    err = {k: float(np.max(np.abs(sol[k] - ref[k])) / max(1.0, float(np.max(np.abs(ref[k]))))) for k in ("x", "u", "lam")}
    print(f"nx {nx} nu {nu} md {md} levels {lv}: d {md*nx} R {md*nx+1+nx} path {g.path} iter {its} {r['iter']} (ref {ref['iter']}) err {err}")
    g.close()
