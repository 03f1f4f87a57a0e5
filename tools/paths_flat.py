"""Diagnostic: device path, iterations and solve time of the flat (per-node data) BASELINE configs C4 / C5 / thesis,
next to the CPU oracle on the same inputs."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle.oracle_py as orc

cases = {"C4": P.random_clipping_qp, "C5": P.pruned_chain_qp, "thesis": P.thesis_example, "irregular": P.irregular_clipping_qp}
for name, mk in cases.items():
    f = mk()
    g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
    for _ in range(3):
        r = g.solve(**f.opts)
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        r = g.solve(**f.opts)
    wall = (time.perf_counter() - t0) / n
    dt = float(g.device_times(n).mean())
    t1 = time.perf_counter()
    ref = min((orc.solve(f.as_dict(), orc.default_opts(**f.opts), lambda0=f.lambda0, traces=False) for _ in range(3)), key=lambda r: r["solver_time"])
    cpu = ref["solver_time"]
    print(f"{name}: nodes {len(f.nk)} path {g.path} status {r['status']} iter {r['iter']} ls {r['ls_total']} launches {r.get('n_launches')} "
          f"device {dt*1e6:.1f} us wall {wall*1e6:.1f} us | cpu oracle {cpu*1e6:.1f} us iter {ref['iter']}")
    g.close()
