"""Campaign on the per-node choice of the stage solver (opts->qp_solver[], dual_Newton_tree.c:124-162) on the GPU box: random trees
whose nodes are, at random, clipping nodes (diagonal weights, box bounds) or dense unconstrained ones (full Q, R, S).  (i) With diagonal
weights on every node the mixed solve must BE the all-clipping solve of the CPU oracle (verdict, iteration count, solution to 1e-9):
the kind of a node only selects the code that solves its stage QP.  (ii) With genuinely dense stage Hessians on the dense nodes there is
no oracle restatement of the mix: the solution is checked against the KKT conditions of the QP with the oracle's residual measure
(convex QP: a KKT point is the solution), 1e-9.
Usage: python tools/fuzz_mixed.py [cases] [first seed]"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle_py as orc


def mixed_problem(seed, dense_blocks, rng):
    f = P.random_shape_qp(seed, depth=int(rng.integers(2, 5)), max_kids=int(rng.integers(2, 4)), nx_range=(1, int(rng.integers(2, 7))), nu_range=(1, int(rng.integers(1, 4))), ubound=float(rng.choice([0.15, 0.3])))
    d = {k: np.array(v, copy=True) for k, v in f.as_dict().items()}
    nk, nx, nu = d["nk"], d["nx"], d["nu"]
    Nn = len(nk)
    kind = (rng.random(Nn) < 0.5).astype(np.int32)
    xo, uo = np.concatenate([[0], np.cumsum(nx)]), np.concatenate([[0], np.cumsum(nu)])
    Q, R, S = [], [], []
    for k in range(Nn):
        Qk, Rk, Sk = np.diag(d["Qd"][xo[k]:xo[k + 1]]), np.diag(d["Rd"][uo[k]:uo[k + 1]]), np.zeros((nu[k], nx[k]))
        if kind[k]:
            d["xmin"][xo[k]:xo[k + 1]] = -1e12; d["xmax"][xo[k]:xo[k + 1]] = 1e12
            d["umin"][uo[k]:uo[k + 1]] = -1e12; d["umax"][uo[k]:uo[k + 1]] = 1e12
            if dense_blocks:
                nz = nx[k] + nu[k]
                M = 0.3 * rng.standard_normal((nz, nz))
                H = np.block([[Qk, Sk.T], [Sk, Rk]]) + M @ M.T
                Qk, Rk, Sk = H[:nx[k], :nx[k]], H[nx[k]:, nx[k]:], H[nx[k]:, :nx[k]]
        Q.append(Qk.ravel(order="F")); R.append(Rk.ravel(order="F")); S.append(Sk.ravel(order="F"))
    d["Q"], d["R"], d["S"] = np.concatenate(Q), np.concatenate(R), np.concatenate(S)
    return d, kind, f.lambda0


def run(n=100, s0=3000):
    stats = {"cases": 0, "fail": 0, "tie": 0, "kkt_max": 0.0}
    t0 = time.perf_counter()
    for c in range(n):
        seed = s0 + c
        rng = np.random.default_rng(seed)
        dense = c % 2 == 1
        d, kind, lam0 = mixed_problem(seed, dense, rng)
        if not (0 < kind.sum() < len(kind)):
            continue
        stats["cases"] += 1
        g = capi.TqGpu(d["nk"], d["nx"], d["nu"]).upload_mixed(d, kind, lam0)
        if not dense:
            ref = orc.solve(d, lambda0=lam0)
            r = g.solve()
            sol = g.solution()
            err = max(float(np.max(np.abs(sol[k] - ref[k]))) if len(ref[k]) else 0.0 for k in ("x", "u", "lam")) / max(1.0, float(np.max(np.abs(ref["x"]))))
            if not ((r["status"], r["iter"]) == (ref["status"], ref["iter"]) and err < 1e-9):
                if r["status"] == 0 and ref["status"] == 0 and err < 1e-5:
                    stats["tie"] += 1
                else:
                    stats["fail"] += 1
                    print(f"MISMATCH seed {seed} (diagonal weights): device {(r['status'], r['iter'])} oracle {(ref['status'], ref['iter'])} err {err:.2e}, {int(kind.sum())} of {len(kind)} nodes dense", flush=True)
        else:
            r = g.solve(stationarityTolerance=1e-10, maxIter=200)
            sol = g.solution()
            kkt = float(orc.max_kkt(d, sol, dense=True))
            stats["kkt_max"] = max(stats["kkt_max"], kkt if r["status"] == 0 else 0.0)
            if r["status"] != 0 or not (kkt < 1e-9):
                stats["fail"] += 1
                print(f"MISMATCH seed {seed} (dense blocks): status {r['status']} iter {r['iter']} KKT residual {kkt:.2e}, {int(kind.sum())} of {len(kind)} nodes dense", flush=True)
        g.close()
        if c % 100 == 99:
            print(f"  {c + 1} cases, {stats['fail']} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"{stats['cases']} trees with a random mix of clipping and dense unconstrained nodes (seeds {s0}..{s0 + n - 1}; every second one with dense stage Hessians, checked by its KKT residual: largest {stats['kkt_max']:.1e}): "
          f"{stats['fail']} mismatches; {stats['tie']} rounding-level endgames")
    return stats


if __name__ == "__main__":
    st_ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 3000)
    sys.exit(1 if st_["fail"] else 0)
