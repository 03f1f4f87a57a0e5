"""Parity campaign on trees with WIDE dual blocks (the MFMA kernels of the three-launch family, 16 < d <= 64, and what lies beyond them):
random shapes with 10 - 21 states per node and 2 - 4 children (blocks of 20 - 84 rows), random option sets, against the CPU oracle;
classes of differences as in tools/fuzz_persist.py.  Usage: python tools/fuzz_wide.py [cases] [first seed]"""
import sys, time, importlib.util
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle_py as orc
from helpers import ulp_sensitivity, ulp_solution_spread


def case(seed):
    rng = np.random.default_rng(seed)
    f = P.random_shape_qp(seed, depth=int(rng.integers(2, 4)), max_kids=int(rng.integers(2, 5)), nx_range=(10, int(rng.integers(11, 22))), nu_range=(2, int(rng.integers(3, 9))), ubound=float(rng.choice([0.2, 0.5, 2.0])))
    opts = dict(maxIter=int(rng.choice([3, 100])), termCondition=int(rng.integers(0, 3)), regType=int(rng.integers(1, 3)), lineSearchMaxIter=int(rng.choice([20, 100])), lineSearchGamma=0.1, lineSearchBeta=float(rng.choice([0.6, 0.8])))
    opts["stationarityTolerance"] = 1e-12 if opts["termCondition"] == 0 else 1e-8
    opts["regValue"] = 1e-8 if opts["regType"] == 1 else 1e-6
    return f, opts


def run(n=100, s0=500000):
    stats = {"cases": 0, "fail": 0, "tie": 0, "ulp": 0, "cond": 0, "cut": 0, "ill": 0}
    dims, paths = {}, {}
    t0 = time.perf_counter()
    for c in range(n):
        seed = s0 + c
        f, opts = case(seed)
        ref = orc.solve(f.as_dict(), orc.default_opts(**opts), lambda0=f.lambda0)
        g = capi.TqGpu(f.nk, f.nx, f.nu).upload(f.as_dict(), f.lambda0)
        r = g.solve(**opts); r2 = g.solve(**opts)
        sol = g.solution(); dev_ls = g.iteration_log(256)[0]; path = g.path
        g.close()
        stats["cases"] += 1
        nk, nx = np.asarray(f.nk), np.asarray(f.nx)
        kid0 = np.concatenate([[1], 1 + np.cumsum(nk)[:-1]])
        dmax = max(int(nx[kid0[k]:kid0[k] + nk[k]].sum()) for k in range(len(nk)) if nk[k] > 0)
        b = "d <= 16" if dmax <= 16 else "16 < d <= 64" if dmax <= 64 else "d > 64"
        dims[b] = dims.get(b, 0) + 1; paths[path] = paths.get(path, 0) + 1
        same = all((q["status"], q["iter"], q["ls_total"]) == (ref["status"], ref["iter"], ref["ls_total"]) for q in (r, r2))
        err = max(float(np.max(np.abs(sol[k] - ref[k]))) / max(1.0, float(np.max(np.abs(ref[k])))) if len(ref[k]) else 0.0 for k in ("x", "u", "lam"))
        if same and err < 1e-9:
            continue
        if same and ref["status"] == 1 and err < 1e-6:
            stats["cut"] += 1; continue
        n_ = min(r["iter"], ref["iter"])
        kd = next((k for k in range(n_) if int(dev_ls[k]) != int(ref["trace_ls"][k])), n_)
        tolv = opts["stationarityTolerance"]
        te = ref["trace_err"][:ref["iter"] + 1]
        near = [k for k in range(len(te)) if te[k] < 10.0 * tolv]
        kstar = near[0] if near else len(te)
        # (an endgame at the tolerance may also be cut by the iteration cap on one side: the oracle accepts the full step of its last iteration and is done,
        # the device's Armijo test -- decided by the last bits of two dual function values -- rejects it, backtracks, and the cap ends the run next to the optimum)
        if ref["status"] in (0, 1) and r["status"] in (0, 1) and r2["status"] in (0, 1) and err < 1e-5 and kd >= max(0, min(kstar - 1, n_)) and (ref["status"] == 0 or r["status"] == 0):
            stats["tie"] += 1; continue
        if not same and ulp_sensitivity(orc, f, opts, kd, copies=4) > 0:
            stats["ulp"] += 1; continue
        if same and ulp_solution_spread(orc, f, opts) >= 0.1 * err:
            stats["cond"] += 1; continue
        stats["fail"] += 1
        print(f"MISMATCH seed {seed} path {path} [{f.name}] largest block {dmax} opts {opts}: device {(r['status'], r['iter'], r['ls_total'])} / {(r2['status'], r2['iter'], r2['ls_total'])} oracle {(ref['status'], ref['iter'], ref['ls_total'])} err {err:.2e}", flush=True)
    print(f"{stats['cases']} random trees with wide blocks (seeds {s0}..{s0 + n - 1}; largest block of the tree: {dict(sorted(dims.items()))}; device paths {dict(sorted(paths.items()))}): {stats['fail']} mismatches; "
          f"{stats['tie']} rounding-level endgames inside the 10 x tolerance window, {stats['ulp']} more by the perturbation test, {stats['cond']} equal-count runs whose difference is the run's conditioning, {stats['cut']} runs cut short by the iteration cap with equal counts and solutions within 1e-6; {time.perf_counter() - t0:.0f} s")
    return stats


if __name__ == "__main__":
    st_ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 500000)
    sys.exit(1 if st_["fail"] else 0)
