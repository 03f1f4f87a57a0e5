"""Diagnostic: uniform trees whose dual blocks are wider than 16 rows (nx * md > 16): the models of the reference's benchmark
sweep (linear_chain nm = 4 / 8, md = 2 .. 4) on the launch-per-level path with the MFMA block kernels, next to the CPU oracle."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle.oracle_py as orc
from helpers import oracle_flat_from_lti
cases = [("nm=8 (nx=16,nu=7) md=2 Nr=Nh=9", lambda: P.linear_chain(2, 9, 9, nm=8, nu=7)),
         ("nm=4 (nx=8,nu=3) md=3 Nr=Nh=6", lambda: P.linear_chain(3, 6, 6)),
         ("nm=4 (nx=8,nu=3) md=4 Nr=Nh=5", lambda: P.linear_chain(4, 5, 5)),
         ("nm=8 (nx=16,nu=7) md=3 Nr=Nh=5", lambda: P.linear_chain(3, 5, 5, nm=8, nu=7)),
         ("nm=4 md=3 Nr=2 Nh=20 (multistage)", lambda: P.linear_chain(3, 2, 20))]
for name, mk in cases:
    p = mk()
    flat = oracle_flat_from_lti(orc, p)
    ref = min((orc.solve(flat, lambda0=p.lambda0, traces=False) for _ in range(3)), key=lambda r: r["solver_time"])
    g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
    for _ in range(3):
        r = g.solve()
    n = 20
    for _ in range(n):
        r = g.solve()
    dt = float(g.device_times(n).mean())
    sol = g.solution()
    err = max(float(np.max(np.abs(sol[k] - ref[k])) / max(1.0, float(np.max(np.abs(ref[k]))))) for k in ("x", "u", "lam"))
    print(f"{name}: nodes {p.Nn} d {p.nx * p.md} path {g.path} iter {r['iter']} (oracle {ref['iter']}) err {err:.1e} launches {r['n_launches']} | device {dt*1e6:.0f} us = {dt*1e6/max(r['iter'],1):.0f} us/it | cpu oracle 1 thread {ref['solver_time']*1e6:.0f} us = {ref['solver_time']*1e6/max(ref['iter'],1):.0f} us/it")
    g.close()
