"""Single small / irregular trees: device time of one solve against the CPU oracle on one thread (same box).
Usage: python tools/single_trees.py [substring of the case name]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle"))
from treeqp_amd import capi, problems as P
import oracle_py as orc

cases = [("thesis example", P.thesis_example()), ("pruned chain (one C5 tree)", P.pruned_chain_qp()),
         ("irregular clipping", P.irregular_clipping_qp()), ("random shape seed 5", P.random_shape_qp(5))]
only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, f in cases:
    if only and only not in name:
        continue
    g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
    g.event_timing(False)
    for _ in range(10):
        r = g.solve(**f.opts)
    n = 100
    t0 = time.perf_counter()
    for _ in range(n):
        r = g.solve(**f.opts)
    g.device_times(1)
    wall = (time.perf_counter() - t0) / n
    o = orc.default_opts(num_threads=1, **f.opts)
    best = min(orc.solve(f.as_dict(), o, f.lambda0, traces=False)["solver_time"] for _ in range(20))
    print(f"{name:28s} nodes {len(f.nk):4d} path {g.path} status {r['status']} iter {r['iter']:3d} ls {r['ls_total']:3d}  "
          f"gpu wall {wall * 1e6:8.1f} us  kernel clock {r['device_time'] * 1e6:8.1f} us  cpu oracle {best * 1e6:8.1f} us  gpu/cpu {wall / best:5.2f}", flush=True)
    g.close()
