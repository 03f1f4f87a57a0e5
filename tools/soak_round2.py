"""Soak of the round-2 launch protocols: many back-to-back solves, every verdict and solution identical to the first.
  * launch-per-phase path with a sweep as one launch and the reductions as sweep tails (a pruned tree, an irregular tree)
  * batches of one shape as ONE launch (C1 x 24 trees, C2 x 4 trees)
  * single persistent launches with the short and the long poll naps (C2: 73 workgroups, C3: 293)
Usage: python tools/soak_round2.py [solves per case]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000


def lti(p):
    nk = p.nk()
    nx = np.full(p.Nn, p.nx, dtype=np.int32)
    nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
    return capi.TqGpu(nk, nx, nu).upload(qp.flat(), p.lambda0)


os.environ["TREEQP_AMD_PATH"] = "generic"
for name, f in (("pruned tree, generic path", P.pruned_chain_qp()), ("irregular tree, generic path", P.irregular_clipping_qp())):
    g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
    g.event_timing(False)
    r0 = g.solve(**f.opts)
    s0 = g.solution()
    t0 = time.perf_counter()
    for i in range(n):
        r = g.solve(**f.opts)
        assert (r["status"], r["iter"], r["ls_total"]) == (r0["status"], r0["iter"], r0["ls_total"]), (i, r, r0)
        if i % 500 == 0:
            print(f"  {name}: {i} solves, {(time.perf_counter() - t0) / max(i, 1) * 1e6:.0f} us each", flush=True)
    s1 = g.solution()
    assert all(np.array_equal(s0[k], s1[k]) for k in ("x", "u", "lam")), name
    print(f"{name}: {n} solves identical (status {r0['status']}, {r0['iter']} iterations, {r0['ls_total']} trials, path {g.path})", flush=True)
    g.close()
del os.environ["TREEQP_AMD_PATH"]
for name, p, nt in (("C1 x 24 in one launch", P.spring_mass(), 24), ("C2 x 4 in one launch", P.linear_chain(2, 9, 9), 4)):
    ms = [lti(p) for _ in range(nt)]
    rs0 = capi.solve_batch(ms)
    sol0 = ms[-1].solution()
    t0 = time.perf_counter()
    for i in range(n):
        rs = capi.solve_batch(ms)
        assert all((r["status"], r["iter"], r["ls_total"]) == (q["status"], q["iter"], q["ls_total"]) for r, q in zip(rs, rs0)), i
        if i % 500 == 0:
            print(f"  {name}: {i} batches, {(time.perf_counter() - t0) / max(i, 1) * 1e6:.0f} us each", flush=True)
    sol1 = ms[-1].solution()
    assert all(np.array_equal(sol0[k], sol1[k]) for k in ("x", "u", "lam")), name
    print(f"{name}: {n} batches identical ({rs0[0]['iter']} iterations per tree)", flush=True)
    for m in ms:
        m.close()
for name, p in (("C2 single persistent launch", P.linear_chain(2, 9, 9)), ("C3 single persistent launch (long naps)", P.linear_chain(2, 11, 11))):
    g = lti(p)
    g.event_timing(False)
    r0 = g.solve()
    s0 = g.solution()
    t0 = time.perf_counter()
    for i in range(n):
        r = g.solve()
        assert (r["status"], r["iter"], r["ls_total"]) == (r0["status"], r0["iter"], r0["ls_total"]), (i, r, r0)
        if i % 500 == 0:
            print(f"  {name}: {i} solves, {(time.perf_counter() - t0) / max(i, 1) * 1e6:.0f} us each", flush=True)
    s1 = g.solution()
    assert all(np.array_equal(s0[k], s1[k]) for k in ("x", "u", "lam")), name
    assert g.path == 2, g.path
    print(f"{name}: {n} solves identical (status {r0['status']}, {r0['iter']} iterations, path {g.path})", flush=True)
    g.close()
print("soak ok")
