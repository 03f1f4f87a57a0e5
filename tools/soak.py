"""Diagnostic: many back-to-back solves on the persistent path (single, batched, different start points); every verdict must be
status 0 with the expected iteration count -- a lost hand-over would show up as a timeout (UNKNOWN_ERROR) or a wrong count."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t0 = time.time()
for name, p, reps in (("C2", P.linear_chain(2, 9, 9), n), ("C3", P.linear_chain(2, 11, 11), n // 4), ("C1", P.spring_mass(), n)):
    qp = product_qp_from_lti(capi, p)
    flat = qp.flat()
    g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
    r0 = g.solve()
    bad = 0
    for i in range(reps):
        r = g.solve()
        bad += (r["status"] != 0) or (r["iter"] != r0["iter"]) or (r["n_launches"] != 1)
    print(f"{name}: {reps} solves, iter {r0['iter']}, bad {bad}, path {g.path}, elapsed {time.time()-t0:.1f} s", flush=True)
    if name == "C2":
        ms = [g] + [capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0) for _ in range(2)]
        badb = 0
        for i in range(reps // 4):
            rs = capi.solve_batch(ms)
            badb += any(r["status"] != 0 or r["iter"] != r0["iter"] for r in rs)
        print(f"{name}: {reps // 4} batched steps of 3 trees, bad {badb}", flush=True)
        # random warm starts (backtracking now and then)
        rng = np.random.default_rng(0)
        badw = 0
        for i in range(2000):
            g.set_lambda(0.5 * rng.standard_normal(len(p.lambda0)))
            r = g.solve()
            badw += r["status"] != 0
        print(f"{name}: 2000 random starts, not converged {badw}", flush=True)
        for m in ms[1:]:
            m.close()
    g.close()
