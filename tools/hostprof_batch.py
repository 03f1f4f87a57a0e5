"""Diagnostic (build with TQ_HOSTPROF=1): host-side split of a solve -- before the launch, the launch call, launch -> verdict -- for single solves
(tqgpu_solve_n) and for batch calls of 1 and 7 C2 trees (tqgpu_solve_batch_n)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti
p = P.linear_chain(2, 9, 9)
flat = product_qp_from_lti(capi, p).flat()
ms = [capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0) for _ in range(7)]
for m in ms: m.event_timing(False)
for label, fn in (("solve_n", lambda: ms[0].solve_n(400)), ("batch of 1", lambda: capi.solve_batch_n(ms[:1], 400)), ("batch of 7", lambda: capi.solve_batch_n(ms, 400))):
    fn()
    t0 = time.perf_counter(); fn(); dt = (time.perf_counter() - t0) / 400
    print(f"{label}: {1e6 * dt:.1f} us per call", file=sys.stderr, flush=True)
