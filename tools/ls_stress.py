"""Diagnostic: the in-kernel line search of the persistent path against the oracle, over far starts and option corners
(trial limit exhausted, restart trigger, iteration limit inside a batch of trials, beta / gamma)."""
import sys, itertools
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle.oracle_py as orc
from helpers import oracle_flat_from_lti

cases = [("chain_2_6_6_u0.1", lambda: P.linear_chain(2, 6, 6, ubound=0.1)),
         ("mstage_3_2_7", lambda: P.spring_mass(md=3, Nr=2, Nh=7)),
         ("chain_2_8_8", lambda: P.linear_chain(2, 8, 8, ubound=0.2))]
optsets = [dict(), dict(lineSearchMaxIter=3), dict(lineSearchMaxIter=5, lineSearchRestartTrigger=2), dict(lineSearchBeta=0.3),
           dict(lineSearchBeta=0.9, lineSearchMaxIter=40), dict(lineSearchGamma=0.4), dict(maxIter=3), dict(maxIter=1),
           dict(lineSearchMaxIter=1), dict(termCondition=1), dict(regType=1, regValue=1e-7), dict(lineSearchMaxIter=9, maxIter=6)]
bad = 0; n = 0; skipped = 0
for name, mk in cases:
    p = mk()
    flat = oracle_flat_from_lti(orc, p)
    g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
    assert g.path == 2, (name, g.path)
    for seed, scale in itertools.product(range(6), (0.3, 1.0, 3.0, 10.0)):
        rng = np.random.Generator(np.random.PCG64(seed))
        lam0 = scale * rng.standard_normal(len(p.lambda0))
        for o in optsets:
            ref = orc.solve(flat, orc.default_opts(**o), lam0)
            if ref["status"] != 0:
                skipped += 1
                continue          # the far start defeats the method itself (singular dual Hessian, chaotic trajectory): no parity to check
            g.set_lambda(lam0)
            r = g.solve(**o)
            sol = g.solution()
            n += 1
            same = (r["status"] == ref["status"] and r["iter"] == ref["iter"] and r["ls_total"] == ref["ls_total"])
            err = max(float(np.max(np.abs(sol[k] - ref[k])) / max(1.0, float(np.max(np.abs(ref[k])))) ) for k in ("x", "u", "lam"))
            if not same or not (err < 1e-7):
                bad += 1
                print(f"MISMATCH {name} seed {seed} scale {scale} opts {o}: gpu {r['status']}/{r['iter']}/{r['ls_total']} launches {r['n_launches']} ref {ref['status']}/{ref['iter']}/{ref['ls_total']} err {err:.2e}")
    g.close()
print(f"{n} solves compared, {skipped} skipped (oracle did not converge), {bad} mismatches")
