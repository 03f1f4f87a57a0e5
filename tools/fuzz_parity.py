"""Parity campaign on the GPU box: seeded random problems over tree shapes, node dimensions, bounds and option sets, each solved by the
device path the library picks (and by the launch-per-phase path where that differs) and by the CPU oracle; verdict, iteration and
trial counts must be EQUAL, the solution within 1e-9 (relative to the largest entry).  A case where verdict or counts differ but both runs converge to the same optimum (solution difference below 1e-5) is listed as a decision at rounding level: near the optimum two dual function values differ in the last bit and
the Armijo or termination test is a coin flip in any implementation.  Precisely: such a case must have the same trial count in EVERY
iteration before the first one at which the oracle's own error is within a factor 10 of the tolerance, and reach the same optimum.  A case outside that window is
still listed as rounding-level when the ORACLE's own trial counts change, at or before the first iteration in which device and oracle
differ, under one-ulp perturbations of the problem data (tests/helpers.py::ulp_sensitivity) -- then no implementation of the algorithm
can be expected to reproduce the decision.  Cases the oracle itself marks ill-conditioned
(more than 40 iterations or 400 trials) are counted apart: implementations legitimately part ways there.
Usage: python tools/fuzz_parity.py [cases] [first seed]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle_py as orc
from helpers import ulp_sensitivity

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.Generator(np.random.PCG64(s0))
stats = {"cases": 0, "solves": 0, "ill": 0, "fail": 0}
by_path = {}
t0 = time.perf_counter()
for c in range(n):
    seed = s0 + c
    kind = c % 4
    if kind == 0:
        f = P.random_shape_qp(seed, depth=int(rng.integers(2, 6)), max_kids=int(rng.integers(2, 5)), nx_range=(1, int(rng.integers(2, 9))), nu_range=(1, int(rng.integers(1, 5))), ubound=float(rng.choice([0.1, 0.3, 1.0])))
    elif kind == 1:
        f = P.pruned_chain_qp(Nh=int(rng.integers(4, 11)), seed=seed)
    elif kind == 2:
        md = int(rng.integers(1, 4)); Nr = int(rng.integers(1, 5)); Nh = Nr + int(rng.integers(0, 4))
        f = P.random_uniform_tree_qp(seed, nx=int(rng.choice([2, 4, 8])), nu=int(rng.integers(1, 4)), md=md, Nr=Nr, Nh=Nh, ubound=float(rng.choice([0.2, 0.4, 2.0])))
    else:
        f = P.random_shape_qp(seed, depth=int(rng.integers(2, 4)), max_kids=3, nx_range=(6, 14), nu_range=(2, 6), ubound=float(rng.choice([0.2, 0.5])))      # blocks of 16 < d <= 42 rows
    opts = dict(f.opts) if getattr(f, "opts", None) else {}
    opts.update(termCondition=int(rng.integers(0, 3)), regType=int(rng.integers(0, 3)))
    if opts["termCondition"] == 0:
        opts["stationarityTolerance"] = 1e-12
    if opts["regType"] == 1:
        opts["regValue"] = 1e-8
    try:
        ref = orc.solve(f.as_dict(), orc.default_opts(**opts), lambda0=f.lambda0)
    except Exception as e:                      # the oracle refuses the case (option set it does not take)
        continue
    ill = ref["iter"] > 40 or ref["ls_total"] > 400 or ref["status"] != 0
    stats["cases"] += 1
    for path in ("auto", "generic"):
        os.environ["TREEQP_AMD_PATH"] = path
        try:
            g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
        finally:
            os.environ.pop("TREEQP_AMD_PATH", None)
        if path == "generic" and g.path != 0:
            g.close(); continue
        r = g.solve(**opts)
        r2 = g.solve(**opts)                     # again: predicted chunks / trial counts of the first solve
        sol = g.solution()
        dev_ls = g.iteration_log(256)[0]
        gpath = g.path
        g.close()
        stats["solves"] += 2
        by_path[gpath] = by_path.get(gpath, 0) + 1
        same = all((q["status"], q["iter"], q["ls_total"]) == (ref["status"], ref["iter"], ref["ls_total"]) for q in (r, r2))
        scale = max(1.0, float(np.max(np.abs(ref["x"]))) if len(ref["x"]) else 1.0)
        err = max(float(np.max(np.abs(sol[k] - ref[k]))) if len(ref[k]) else 0.0 for k in ("x", "u", "lam")) / scale
        ok = same and err < 1e-9
        # an Armijo test decided by the last bits of two dual function values (the sums are taken in different orders): same verdict and
        # iteration count, one or two trials more or less, both solutions within the termination tolerance of the optimum
        # ... or a termination test within rounding of the tolerance: one iteration more or less at the optimum (dual function values
        # that differ in the last bit: the line search of such an iteration is a coin flip in the reference as well)
        # The criterion: both runs converge to the same optimum, and they are IDENTICAL (trial count of every iteration) up to the first
        # iteration at which the oracle's own error is within a factor 10 of the tolerance -- what differs is the endgame only.
        tolv = opts.get("stationarityTolerance", 1e-8)
        te = ref["trace_err"][:ref["iter"] + 1]
        te_cmp = te                                                   # (in the norm the tolerance is compared with)
        near = [k for k in range(len(te_cmp)) if te_cmp[k] < 10.0 * tolv]
        kstar = near[0] if near else len(te_cmp)                      # iterations 0 .. kstar - 2 took their steps far from the tolerance
        npre = max(0, min(kstar - 1, r["iter"], ref["iter"]))
        prefix_same = all(int(dev_ls[k]) == int(ref["trace_ls"][k]) for k in range(npre))
        tie = (not ok) and ref["status"] == 0 and all(q["status"] == 0 for q in (r, r2)) and err < 1e-5 and prefix_same and r["iter"] >= npre
        if tie:
            stats["tie"] = stats.get("tie", 0) + 1
            print(f"  (decision at rounding level: seed {seed} path {gpath} device iterations / trials {r['iter']} / {r['ls_total']} oracle {ref['iter']} / {ref['ls_total']}, solution difference {err:.1e}, "
                  f"tolerance {opts.get('stationarityTolerance', 1e-8):.0e}, termCondition {opts['termCondition']}, oracle's last errors {ref['trace_err'][max(ref['iter'] - 1, 0)]:.2e} -> {ref['trace_err'][ref['iter']]:.2e})", flush=True)
        elif ill and not ok:
            stats["ill"] += 1
        elif not ok and ref["status"] == 0 and all(q["status"] == 0 for q in (r, r2)) and err < 1e-5 and (moved := ulp_sensitivity(orc, f, opts, kd := next(
                (k for k in range(min(r["iter"], ref["iter"])) if int(dev_ls[k]) != int(ref["trace_ls"][k])), min(r["iter"], ref["iter"])))) > 0:
            # farther from the tolerance than the window above, and still decided by rounding: the ORACLE's own trial counts change in
            # iteration <= kd (the first one in which device and oracle differ) when its input data moves by one unit in the last place
            stats["ulp"] = stats.get("ulp", 0) + 1
            print(f"  (decision at rounding level, shown by perturbation: seed {seed} path {gpath} device iterations / trials {r['iter']} / {r['ls_total']} oracle {ref['iter']} / {ref['ls_total']}, first difference in iteration {kd} "
                  f"(oracle's error there {ref['trace_err'][kd]:.2e}); {moved} of 6 one-ulp perturbations of the data change the oracle's own trial counts in an iteration <= {kd}; solution difference {err:.1e})", flush=True)
        elif not ok:
            stats["fail"] += 1
            print(f"MISMATCH seed {seed} kind {kind} path {gpath} ({path}) opts {opts}: device {(r['status'], r['iter'], r['ls_total'])} / {(r2['status'], r2['iter'], r2['ls_total'])} oracle {(ref['status'], ref['iter'], ref['ls_total'])} err {err:.2e}  [{f.name}]", flush=True)
    if c % 25 == 24:
        print(f"  {c + 1} cases, {stats['solves']} device solves, {stats['fail']} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
print(f"{stats['cases']} cases (seeds {s0}..{s0 + n - 1}), {stats['solves']} device solves on paths {dict(sorted(by_path.items()))} "
      f"(0 launch per phase / three launches, 1 per tier, 2 persistent, 3 single workgroup): {stats['fail']} mismatches; {stats.get('tie', 0)} Armijo / termination decisions at rounding level (listed above: same optimum, a trial or an iteration more or less); "
      f"{stats.get('ulp', 0)} more where one-ulp perturbations of the data change the oracle's own decisions at the iteration in question; {stats['ill']} differences on cases the oracle marks ill-conditioned or not converged")
sys.exit(1 if stats["fail"] else 0)
