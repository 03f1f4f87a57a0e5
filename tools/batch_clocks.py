"""Diagnostic: where a batch of C2 trees loses time against one tree -- the top workgroup's own clock (start -> verdict, `device_time` of every
member) against the wall clock of the step, for 1 .. 7 trees per launch."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti
p = P.linear_chain(2, 9, 9)
qp = product_qp_from_lti(capi, p)
flat = qp.flat()
for B in (1, 2, 3, 4, 7):
    ms = [capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0) for _ in range(B)]
    for _ in range(30):
        rs = capi.solve_batch(ms) if B > 1 else [ms[0].solve()]
    clocks, walls = [], []
    for _ in range(200):
        t0 = time.perf_counter()
        rs = capi.solve_batch(ms) if B > 1 else [ms[0].solve()]
        walls.append(time.perf_counter() - t0)
        clocks.append([r["device_time"] for r in rs])
    c = np.array(clocks) * 1e6
    print(f"{B} trees: wall per step median {np.median(walls)*1e6:6.1f} us; top workgroup start -> verdict: median over steps of the per-step min {np.median(c.min(axis=1)):6.1f}, of the max {np.median(c.max(axis=1)):6.1f} us")
    for m in ms:
        m.close()
