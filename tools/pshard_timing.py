"""Diagnostic: what dealing the workgroups of ONE persistent launch over n launches costs on ONE device (the protocol of the
sharded mode with real concurrency, without the xGMI hop): C2 / C3, n ranks of this process, against the single launch.
Run with GPU_MAX_HW_QUEUES >= 2 n."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P
reps = 200
for name, levels in (("C2", 9), ("C3", 11)):
    p = P.linear_chain(2, levels, levels)
    nk = p.nk(); nx = np.full(p.Nn, p.nx, dtype=np.int32); nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    flat = capi.TreeQp(nx, nu, nk).fill_lti(p).flat()
    g = capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0)
    g.event_timing(False)
    for _ in range(20):
        r = g.solve()
    t0 = time.perf_counter()
    for _ in range(reps):
        r = g.solve()
    t1 = (time.perf_counter() - t0) / reps
    g.close()
    line = f"{name}: single launch {1e6 * t1:7.1f} us per solve ({r['iter']} iterations)"
    for n in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
        ms = [capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0).pshard_init(r_, n) for r_ in range(n)]
        capi.pshard_solve_local(ms)                         # connects, first solve
        for _ in range(10):
            for m in ms: m.pshard_begin()
            for m in ms: m.pshard_end()
        t0 = time.perf_counter()
        dev = 0.0
        for _ in range(reps):
            for m in ms: m.pshard_begin()
            rs = [m.pshard_end() for m in ms]
            dev += rs[0]["device_time"]
        tn = (time.perf_counter() - t0) / reps
        line += f" | {n} launches: {1e6 * dev / reps:6.1f} us launch start -> verdict in the top workgroup ({1e6 * tn:6.1f} us wall incl. n stream syncs)"
        for m in ms: m.close()
    print(line, flush=True)
