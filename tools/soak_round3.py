"""Soak of the round-3 protocols: many back-to-back solves, every verdict and solution identical to the first.
  * the three-launch family of the wide-block class (k_sgp / k_hf_w / k_fwd3): C4 (one iteration, 3280 nodes), a pruned tree with
    d = 24 blocks, several iterations and line-search trials, a random tree with mixed block sizes and more than four children
  * ONE tree over 2 / 4 ranks of this process inside the persistent launch (C2 and C3; needs GPU_MAX_HW_QUEUES >= 8)
Usage: GPU_MAX_HW_QUEUES=16 python tools/soak_round3.py [solves per case]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import capi, problems as P

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000

os.environ["TREEQP_AMD_PATH"] = "generic"
for name, f, reps in (("C4, three launches per iteration", P.random_clipping_qp(), n), ("pruned tree (d = 24), three launches per iteration", P.pruned_chain_qp(), n),
                      ("random shape seed 5 (mixed blocks, six children)", P.random_shape_qp(5, 2, 6, (3, 9), (2, 5)), n)):
    g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
    g.event_timing(False)
    r0 = g.solve(**f.opts)
    s0 = g.solution()
    t0 = time.perf_counter()
    for i in range(reps):
        r = g.solve(**f.opts)
        assert (r["status"], r["iter"], r["ls_total"]) == (r0["status"], r0["iter"], r0["ls_total"]), (i, r, r0)
        if i % 1000 == 0:
            print(f"  {name}: {i} solves, {(time.perf_counter() - t0) / max(i, 1) * 1e6:.0f} us each", flush=True)
    s1 = g.solution()
    assert all(np.array_equal(s0[k], s1[k]) for k in ("x", "u", "lam")), name
    print(f"{name}: {reps} solves identical (status {r0['status']}, {r0['iter']} iterations, {r0['ls_total']} trials, path {g.path}, {r['n_launches']} launches per solve)", flush=True)
    g.close()
del os.environ["TREEQP_AMD_PATH"]

for name, levels, ranks in (("C2", 9, 2), ("C2", 9, 4), ("C3", 11, 2), ("C3", 11, 4)):
    p = P.linear_chain(2, levels, levels)
    nk = p.nk(); nx = np.full(p.Nn, p.nx, dtype=np.int32); nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
    flat = capi.TreeQp(nx, nu, nk).fill_lti(p).flat()
    g = capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0)
    ref_r, ref = g.solve(), g.solution()
    g.close()
    ms = [capi.TqGpu(nk, nx, nu).upload(flat, p.lambda0).pshard_init(r, ranks) for r in range(ranks)]
    capi.pshard_solve_local(ms)
    t0 = time.perf_counter()
    for i in range(n):
        if i % 50000 == 49999:                               # the 16-bit launch number: rewind before it wraps (all ranks idle here)
            for m in ms: m.pshard_rewind()
        for m in ms: m.pshard_begin()
        rs = [m.pshard_end() for m in ms]
        assert all((r["status"], r["iter"], r["ls_total"]) == (ref_r["status"], ref_r["iter"], ref_r["ls_total"]) for r in rs), (i, rs)
        if i % 1000 == 0:
            print(f"  {name} over {ranks} ranks: {i} solves, {(time.perf_counter() - t0) / max(i, 1) * 1e6:.0f} us each", flush=True)
    capi.pshard_solve_local(ms)                            # once more through the collecting entry point: the solution in every mirror
    for m in ms:
        sol = m.solution()
        assert all(np.array_equal(sol[k], ref[k]) for k in ("x", "u", "lam", "mu_x", "mu_u")), (name, ranks)
        m.close()
    print(f"{name} over {ranks} ranks of one process: {n + 2} sharded solves, every verdict and the final solution identical to the single-device solve", flush=True)
