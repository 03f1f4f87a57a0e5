"""Diagnostic: random trees of SMALL blocks (d <= 16) that are too wide for the single-workgroup kernel (a level of more than 96 nodes): the three-launch
family (round 4) against the launch-per-phase kernels it used to run on (TREEQP_AMD_SMALL_WIDE=0), both against the oracle."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
from treeqp_amd import capi, problems as P
import oracle_py as orc
for seed in (1, 2, 4, 7):
    f = P.random_shape_qp(seed, depth=5, max_kids=4, nx_range=(2, 4), nu_range=(1, 2), ubound=0.3)
    f.opts = dict(getattr(f, 'opts', None) or {})
    ref = orc.solve(f.as_dict(), orc.default_opts(**f.opts), f.lambda0)
    for label, env in (("three launches", {}), ("launch per phase", {"TREEQP_AMD_SMALL_WIDE": "0"})):
        os.environ.update(env)
        g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
        for k in env: os.environ.pop(k)
        g.event_timing(False)
        for _ in range(3): r = g.solve(**f.opts)
        t0 = time.perf_counter()
        for _ in range(10): r = g.solve(**f.opts)
        us = (time.perf_counter() - t0) / 10 * 1e6
        s = g.solution()
        err = max(float(np.max(np.abs(s[k] - ref[k]))) for k in ("x", "u", "lam"))
        print(f"seed {seed}: {len(f.nk)} nodes  {label:17s} path {g.path}  {r['iter']} / {r['ls_total']} (oracle {ref['iter']} / {ref['ls_total']})  launches {r['n_launches']:4d}  {us:8.1f} us  max|dev - oracle| {err:.1e}")
        g.close()
