"""Diagnostic: device time of a C2 solve cut off after k = 1, 2, 3 Newton iterations -> per-iteration cost and fixed cost per launch."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti
p = P.linear_chain(2, 9, 9)
qp = product_qp_from_lti(capi, p); flat = qp.flat()
g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
T = {}
for k in (1, 2, 3, 100):
    for _ in range(20):
        r = g.solve(maxIter=k)
    n = 200
    for _ in range(n):
        r = g.solve(maxIter=k)
    T[k] = float(g.device_times(n).mean()) * 1e6
    print(f"maxIter {k}: status {r['status']} iter {r['iter']} device {T[k]:.1f} us")
print(f"per iteration {T[2]-T[1]:.1f} / {T[3]-T[2]:.1f} us; fixed (launch to first iteration + verdict) {T[1]-(T[2]-T[1]):.1f} us; converged solve adds {T[100]-T[3]:.1f} us for the final termination pass")
