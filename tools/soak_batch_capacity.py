"""Soak of the batch launch at full capacity (every CU but one carries two workgroups; no residency margin) and of the lazy end-of-batch
synchronisation: batches back to back across the 16-bit launch-number wrap, every verdict identical to the first; every 1000 batches a
member that is not the lead is solved on its own and its solution read (both wait for the lead's stream first), and the lead changes.
Usage: python tools/soak_batch_capacity.py [batches per case]"""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
from helpers import product_qp_from_lti

n = int(sys.argv[1]) if len(sys.argv) > 1 else 70000
for name, p in (("C1", P.spring_mass()), ("C2", P.linear_chain(2, 9, 9))):
    flat = product_qp_from_lti(capi, p).flat()
    mk = lambda: capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, p.lambda0)
    g0 = mk()
    geo = g0.geometry()
    nt = geo["capacity"] // geo["workgroups"]
    ms = [g0] + [mk() for _ in range(nt - 1)]
    single = ms[1].solve()
    sol_ref = ms[1].solution()
    rs0 = capi.solve_batch(ms)
    key = lambda r: (r["status"], r["iter"], r["ls_total"])
    assert all(key(r) == key(single) for r in rs0)
    t0 = time.perf_counter()
    order = ms
    for i in range(n):
        rs = capi.solve_batch(order)
        assert all(key(r) == key(single) for r in rs), (name, i, [key(r) for r in rs])
        if i % 1000 == 999:
            m = order[1 + (i // 1000) % (nt - 1)]
            sol = m.solution()
            assert all(np.array_equal(sol[k], sol_ref[k]) for k in ("x", "u", "lam")), (name, i)
            assert key(m.solve()) == key(single)
            order = order[::-1]                    # another lead
        if i % 10000 == 0:
            print(f"  {name} x {nt}: {i} batches, {(time.perf_counter() - t0) / max(i, 1) * 1e6:.0f} us each", flush=True)
    assert all(capi.lib().tqgpu_timeouts(m.h) == 0 for m in ms)
    print(f"{name} x {nt} trees ({nt * geo['workgroups']} workgroups of {geo['capacity']}): {n} batches identical to the single solve "
          f"({single['iter']} iterations), no bounded wait fired, {(time.perf_counter() - t0) / n * 1e6:.0f} us per batch", flush=True)
    for m in ms:
        m.close()
