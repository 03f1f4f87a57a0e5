"""Diagnostic: C5-class pruned trees (different seeds), one at a time and as a batch of independent trees."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle.oracle_py as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
fs = [P.pruned_chain_qp(seed=7 + i) for i in range(n)]
ms = [capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0) for f in fs]
print("path", ms[0].path, "nodes", [len(f.nk) for f in fs[:8]])
opts = fs[0].opts
for _ in range(2):
    r1 = [m.solve(**opts) for m in ms]
t0 = time.perf_counter(); r1 = [m.solve(**opts) for m in ms]; t1 = time.perf_counter() - t0
for _ in range(2):
    rb = capi.solve_batch(ms, **opts)
t0 = time.perf_counter(); rb = capi.solve_batch(ms, **opts); tb = time.perf_counter() - t0
cpu = 0.0
for f in fs:
    cpu += min(orc.solve(f.as_dict(), orc.default_opts(**f.opts), lambda0=f.lambda0, traces=False)["solver_time"] for _ in range(2))
print(f"{n} trees: one at a time {t1*1e3:.2f} ms ({t1/n*1e6:.0f} us/tree), batch {tb*1e3:.2f} ms ({tb/n*1e6:.0f} us/tree), cpu oracle 1 core {cpu*1e3:.2f} ms ({cpu/n*1e6:.0f} us/tree)")
print("iters", [r["iter"] for r in rb][:8], "status", set(r["status"] for r in rb), "same as single:", all(a["iter"] == b["iter"] for a, b in zip(r1, rb)))
