"""Diagnostic: which device path each BASELINE config takes, iterations and solve time."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P

cases = {"C1": lambda: P.spring_mass(), "C2": lambda: P.linear_chain(2, 9, 9), "C3": lambda: P.linear_chain(2, 11, 11),
         "md3": lambda: P.linear_chain(3, 5, 5, nm=2), "C4": None}
for name, mk in cases.items():
    if mk is None:
        continue
    try:
        p = mk()
    except Exception as e:
        print(name, "skip", e); continue
    nk = p.nk()
    nx = np.full(p.Nn, p.nx, dtype=np.int32)
    nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
    g = capi.TqGpu(nk, nx, nu).upload(qp.flat(), p.lambda0)
    for _ in range(5):
        r = g.solve()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        r = g.solve()
    dt = float(g.device_times(n).mean())
    wall = (time.perf_counter() - t0) / n
    print(f"{name}: nodes {p.Nn} nx {p.nx} nu {p.nu} path {g.path} status {r['status']} iter {r['iter']} ls {r['ls_total']} device {dt*1e6:.1f} us wall {wall*1e6:.1f} us -> {r['iter']/wall:.0f} it/s")
    g.close()
