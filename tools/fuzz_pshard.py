"""Campaign on the sharded persistent solve, rehearsed on ONE device: random uniform trees of the sharded kernels' shapes (nx, nu, md) in
{(8,3,2), (4,1,2), (8,2,2), (6,2,2)}, 6 - 10 levels, with randomly perturbed linear terms and bounds and random option sets (termination
norm, regularisation mode, backtracking factor, iteration caps that cut a solve short), solved as n = 2 or 4 launches that wait for each
other through tagged words in every rank's slab (tqgpu_pshard_solve_local) and as one launch: verdict, iteration and trial counts equal,
x, u, lambda, mu equal to the LAST BIT (a workgroup's arithmetic does not depend on the launch it runs in).  What the campaign is after
is the protocol: hand-overs across ranks with any number of backtracking batches, every kind of exit.
Usage: python tools/fuzz_pshard.py [cases] [first seed]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # n launches that wait for each other must all be in flight: streams that share a hardware queue run one after the other
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P

SHAPES = [(4, 3), (2, 1), (4, 2), (3, 2)]      # (nm, nu): nx = 2 nm


def run(n=100, s0=1000):
    stats = {"cases": 0, "fail": 0, "skipped": 0, "exits": {}}
    t0 = time.perf_counter()
    for c in range(n):
        seed = s0 + c
        rng = np.random.default_rng(seed)
        nm, nu_ = SHAPES[int(rng.integers(0, len(SHAPES)))]
        levels = int(rng.integers(6, 11))
        p = P.linear_chain(2, levels, levels, nm=nm, nu=nu_, ubound=float(rng.choice([0.05, 0.2, 0.5])))
        nk = p.nk(); Nn = len(nk)
        nx = np.full(Nn, p.nx, dtype=np.int32); nu = np.where(nk > 0, p.nu, 0).astype(np.int32)
        flat = capi.TreeQp(nx, nu, nk).fill_lti(p).flat()
        flat["q"] = flat["q"] + float(rng.choice([0.0, 0.05, 0.5])) * rng.standard_normal(len(flat["q"]))
        flat["r"] = flat["r"] + float(rng.choice([0.0, 0.05])) * rng.standard_normal(len(flat["r"]))
        if rng.random() < 0.3:
            flat["xmax"] = np.where(flat["xmax"] == flat["xmin"], flat["xmax"], flat["xmax"] * float(rng.uniform(0.2, 1.0)))
        opts = dict(maxIter=int(rng.choice([1, 2, 3, 50, 200])), termCondition=int(rng.integers(0, 3)), regType=int(rng.integers(0, 3)),
                    lineSearchMaxIter=int(rng.choice([3, 20, 100])), lineSearchGamma=0.1, lineSearchBeta=float(rng.choice([0.5, 0.8])))
        opts["stationarityTolerance"] = 1e-12 if opts["termCondition"] == 0 else 1e-8
        opts["regValue"] = 1e-8 if opts["regType"] == 1 else 1e-6
        lam0 = 0.1 * rng.standard_normal(len(p.lambda0)) if rng.random() < 0.5 else p.lambda0
        g = capi.TqGpu(nk, nx, nu).upload(flat, lam0)
        if g.path != 2:
            g.close(); stats["skipped"] += 1; continue
        ref_r = g.solve(**opts)
        ref = g.solution()
        g.close()
        nr = int(rng.choice([2, 4]))
        try:
            ms = [capi.TqGpu(nk, nx, nu).upload(flat, lam0).pshard_init(r, nr) for r in range(nr)]
        except RuntimeError:
            stats["skipped"] += 1; continue
        stats["cases"] += 1
        key = (ref_r["status"], "several trials" if ref_r["ls_total"] > ref_r["iter"] else "one trial per iteration")
        stats["exits"][key] = stats["exits"].get(key, 0) + 1
        for rep in range(2):
            try:
                rs = capi.pshard_solve_local(ms, **opts)
            except RuntimeError as e:
                stats["fail"] += 1
                print(f"FAILED seed {seed} call {rep}: {nr} ranks, {Nn} nodes nx {p.nx} nu {p.nu}, opts {opts}: {e}", flush=True)
                break
            bad = [r for r in rs if (r["status"], r["iter"], r["ls_total"]) != (ref_r["status"], ref_r["iter"], ref_r["ls_total"])]
            diff = [k for m in ms for k in ("x", "u", "lam", "mu_x", "mu_u") if not np.array_equal(m.solution()[k], ref[k], equal_nan=True)]
            if bad or diff:
                stats["fail"] += 1
                print(f"MISMATCH seed {seed} call {rep}: {nr} ranks, {Nn} nodes nx {p.nx} nu {p.nu}, opts {opts}: single {(ref_r['status'], ref_r['iter'], ref_r['ls_total'])} sharded {[(r['status'], r['iter'], r['ls_total']) for r in rs]} arrays that differ {sorted(set(diff))}", flush=True)
                break
        for m in ms:
            m.close()
        if c % 50 == 49:
            print(f"  {c + 1} cases, {stats['fail']} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"{stats['cases']} trees (seeds {s0}..{s0 + n - 1}; {stats['skipped']} not on the persistent path) sharded over 2 or 4 concurrent launches of one device, two solves each, against the single launch, bit for bit: "
          f"{stats['fail']} mismatches; exits seen (status, trials): {dict(sorted(stats['exits'].items()))}")
    return stats


if __name__ == "__main__":
    st_ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1000)
    sys.exit(1 if st_["fail"] else 0)
