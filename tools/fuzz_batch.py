"""Batch campaign on the GPU box: tqgpu_solve_batch on batches of 2 - 12 trees of MIXED classes (uniform and multistage trees of the
persistent kernels' shapes, several of one shape and one of another; small irregular trees; pruned chains; trees with blocks of more than
16 rows) under one option set -- the call groups its members by the launch that can carry them (one persistent batch launch per
shape, one single-workgroup batch launch, the rest one after the other), and this grouping is what the campaign is after.  Every
member against the CPU oracle: verdict, iteration and trial counts equal, solution within 1e-9; the call is made twice (the second
one finds the state the first left).  Rounding-level endgames are listed as in tools/fuzz_parity.py.
Usage: python tools/fuzz_batch.py [batches] [first seed]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle_py as orc


def member(kind, seed, rng):
    """-> (flat dict, lambda0)"""
    if kind in (0, 1):
        p = P.linear_chain(2, (nr := int(rng.integers(3, 7))), nr) if kind == 0 else P.spring_mass()
        nk = p.nk(); Nn = len(nk)
        nx = np.full(Nn, p.nx, dtype=np.int32); nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
        return capi.TreeQp(nx, nu, nk).fill_lti(p).flat(), p.lambda0
    if kind == 2:
        f = P.random_shape_qp(seed, depth=int(rng.integers(2, 5)), max_kids=3, nx_range=(1, 6), nu_range=(1, 3), ubound=0.3)
    elif kind == 3:
        f = P.pruned_chain_qp(Nh=int(rng.integers(4, 9)), seed=seed)
    elif kind == 4:
        f = P.random_shape_qp(seed, depth=int(rng.integers(2, 4)), max_kids=3, nx_range=(6, 12), nu_range=(2, 5), ubound=0.3)
    else:
        f = P.random_uniform_tree_qp(seed, nx=int(rng.choice([2, 4, 8])), nu=int(rng.integers(1, 4)), md=int(rng.integers(2, 4)), Nr=(nr := int(rng.integers(2, 5))), Nh=nr + int(rng.integers(0, 3)), ubound=0.4)
    return f.as_dict(), f.lambda0


def run(n_batches=50, s0=5000):
    stats = {"batches": 0, "solves": 0, "fail": 0, "tie": 0, "ill": 0}
    shapes = {}
    t0 = time.perf_counter()
    for q in range(n_batches):
        seed = s0 + q
        rng = np.random.default_rng(seed)
        nb = int(rng.integers(2, 13))
        lead = int(rng.integers(0, 6))
        kinds = [lead if rng.random() < 0.5 else int(rng.integers(0, 6)) for _ in range(nb)]      # half of the members share a class
        opts = dict(maxIter=200, termCondition=int(rng.integers(0, 3)), regType=int(rng.integers(0, 3)), lineSearchMaxIter=100, lineSearchGamma=0.1, lineSearchBeta=float(rng.choice([0.6, 0.8])))
        opts["stationarityTolerance"] = 1e-12 if opts["termCondition"] == 0 else 1e-8
        opts["regValue"] = 1e-8 if opts["regType"] == 1 else 1e-6
        items = [member(k, seed * 100 + i, rng) for i, k in enumerate(kinds)]
        refs = [orc.solve(fl, orc.default_opts(**opts), lambda0=l0) for fl, l0 in items]
        ms = [capi.TqGpu(fl["nk"], fl["nx"], fl["nu"]).upload(fl, l0) for fl, l0 in items]
        key = tuple(sorted(m.path for m in ms))
        shapes[key] = shapes.get(key, 0) + 1
        for rep in range(2):
            rs = capi.solve_batch(ms, **opts)
            for i, (m, r, ref) in enumerate(zip(ms, rs, refs)):
                stats["solves"] += 1
                sol = m.solution()
                # (every array relative to its own largest entry: a nearly infeasible problem has duals of 1e5 and more)
                err = max(float(np.max(np.abs(sol[k] - ref[k]))) / max(1.0, float(np.max(np.abs(ref[k])))) if len(ref[k]) else 0.0 for k in ("x", "u", "lam"))
                same = (r["status"], r["iter"], r["ls_total"]) == (ref["status"], ref["iter"], ref["ls_total"])
                if same and err < 1e-9:
                    continue
                if ref["status"] != 0 or ref["iter"] > 40 or ref["ls_total"] > 400:
                    stats["ill"] += 1
                elif r["status"] == 0 and err < 1e-5:
                    stats["tie"] += 1
                    print(f"  (rounding-level endgame: batch {seed} member {i} kind {kinds[i]} path {m.path}: device {r['iter']} / {r['ls_total']} oracle {ref['iter']} / {ref['ls_total']}, difference {err:.1e})", flush=True)
                else:
                    stats["fail"] += 1
                    print(f"MISMATCH batch {seed} call {rep} member {i} of {nb} kinds {kinds} path {m.path} opts {opts}: device {(r['status'], r['iter'], r['ls_total'])} oracle {(ref['status'], ref['iter'], ref['ls_total'])} err {err:.2e}", flush=True)
        for m in ms:
            m.close()
        stats["batches"] += 1
        if q % 25 == 24:
            print(f"  {q + 1} batches, {stats['solves']} member solves, {stats['fail']} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
    mixes = sum(1 for k in shapes if len(set(k)) > 1)
    print(f"{stats['batches']} batches (seeds {s0}..{s0 + n_batches - 1}), {stats['solves']} member solves; {len(shapes)} different mixes of member paths, {mixes} of them with more than one path in the batch: "
          f"{stats['fail']} mismatches; {stats['tie']} rounding-level endgames (same optimum); {stats['ill']} differences on members the oracle marks ill-conditioned or not converged")
    return stats


if __name__ == "__main__":
    st_ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 5000)
    sys.exit(1 if st_["fail"] else 0)
