"""checkLastActiveSet == 2 (the kernel variant that keeps factor data) against the default kernel: device time per solve,
result differences, iteration counts (tools; GPU box).  With a TQ_REUSE_DEBUG build `ls_last` of the result carries the number
of passes in which the TOP workgroup kept its factors."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P


def flat_of(p, eliminate_x0=False):
    nk = p.nk(); nx = np.full(p.Nn, p.nx, dtype=np.int32); nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
    if eliminate_x0: qp.eliminate_x0()
    return qp.flat()


CASES = [("C2", P.linear_chain(2, 9, 9), False, None, {}),
         ("C1", P.spring_mass(), False, None, {}),
         ("C1 x0 elim. xmax1=0.2 (58 it)", P.spring_mass(xmax1=0.2), True, None, {}),
         ("C2 far start beta 0.9", P.linear_chain(2, 9, 9), False, 3.0, dict(lineSearchBeta=0.9, lineSearchMaxIter=40)),
         ("chain 2,6,6 far start", P.linear_chain(2, 6, 6, ubound=0.1), False, 3.0, dict(lineSearchBeta=0.9, lineSearchMaxIter=40)),
         ("mstage 3,2,7 far start", P.spring_mass(md=3, Nr=2, Nh=7), False, 3.0, {})]
for name, p, elim, scale, o in CASES:
    f = flat_of(p, elim)
    lam0 = p.lambda0 if scale is None else scale * np.random.Generator(np.random.PCG64(0)).standard_normal(len(p.lambda0))
    if elim: lam0 = np.zeros(int(np.sum(f["nx"])) - int(f["nx"][0]))
    sols = {}
    for mode in (1, 2):
        g = capi.TqGpu(f["nk"], f["nx"], f["nu"]).upload(f, lam0)
        ts = []
        for i in range(60):
            g.set_lambda(lam0)
            r = g.solve(checkLastActiveSet=mode, **o)
            if i >= 20: ts.append(r["device_time"])
        sols[mode] = g.solution()
        print(f"{name:32s} checkLastActiveSet={mode}: {np.median(ts) * 1e6:8.1f} us  status {r['status']} iter {r['iter']} ls {r['ls_total']}  (ls_last {r['ls_last']})")
        g.close()
    print(f"{'':32s} max |x2 - x1| = {np.max(np.abs(sols[2]['x'] - sols[1]['x'])):.2e}, |lam2 - lam1| = {np.max(np.abs(sols[2]['lam'] - sols[1]['lam'])):.2e}")
