"""Sequence campaign through the DROP-IN API (treeqp_tdunes_solve) on the GPU box: a solver object lives through a random sequence of
solves with the caller changing the problem in between -- b of an edge, q / r of a node, the weights of a node, the bounds of a node,
A / B of an edge, nothing -- and warm-starting from the previous duals or not.  treeqp_tdunes_solve re-reads qp_in at every call
(dual_Newton_tree.c:1142-1160); the device mirror uploads only what changed and repacks only what depends on it, and this is what the
campaign is after: every solve is compared with the CPU oracle solving the problem as it stands (from the same starting duals):
same verdict and iteration count, solution within 1e-9.
Usage: python tools/fuzz_sequence.py [sequences] [first seed] [steps per sequence]"""
import sys, time
import ctypes as C
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle")); sys.path.insert(0, str(ROOT / "tests"))
from treeqp_amd import capi, problems as P
import oracle_py as orc



def make(kind, seed, rng):
    if kind == 0:
        return P.random_uniform_tree_qp(seed, nx=int(rng.choice([2, 4, 8])), nu=int(rng.integers(1, 4)), md=int(rng.integers(2, 4)), Nr=(nr := int(rng.integers(2, 5))), Nh=nr + int(rng.integers(0, 3)), ubound=0.4)
    if kind == 1:
        return P.random_shape_qp(seed, depth=int(rng.integers(2, 5)), max_kids=3, nx_range=(1, 6), nu_range=(1, 3), ubound=0.3)
    if kind == 2:
        return P.pruned_chain_qp(Nh=int(rng.integers(4, 9)), seed=seed)
    return P.random_shape_qp(seed, depth=int(rng.integers(2, 4)), max_kids=3, nx_range=(6, 12), nu_range=(2, 5), ubound=0.3)      # blocks of 16 < d <= 36 rows


def run(n_seq=40, s0=7000, n_steps=12):
    stats = {"solves": 0, "fail": 0, "tie": 0}
    by_path = {}
    t0 = time.perf_counter()
    for q in range(n_seq):
        seed = s0 + q
        rng = np.random.default_rng(seed)
        kind = seed % 8
        x0 = None
        if kind >= 6:
            # the same two classes with x0 eliminated (tree_qp_in_eliminate_x0: the root has no state; fault_tolerance.c:625-632 changes x0 between
            # solves with tree_qp_in_set_x0_colmaj): change 6 below is a new x0
            f = P.linear_chain(2, (nr := int(rng.integers(3, 7))), nr) if kind == 6 else P.spring_mass(Nh=int(rng.integers(3, 8)), Nr=int(rng.integers(1, 3)))
            nk = f.nk(); Nn = len(nk)
            qp = capi.TreeQp(np.full(Nn, f.nx, dtype=np.int32), np.where(nk > 0, f.nu, 0).astype(np.int32), nk).fill_lti(f)
            qp.eliminate_x0()
            nx, nu = qp.nx, qp.nu
            x0 = np.asarray(f.x0, dtype=float)
            opts = {}
            f.name = f"lti kind {kind} Nn={Nn}, x0 eliminated"
            f.lambda0 = None
        elif kind >= 4:
            # uniform / multistage trees of the shapes the persistent single-launch kernels are instantiated for, through the reference's own
            # fill routine (tree_qp_in_fill_lti_data_diag_weights)
            f = P.linear_chain(2, (nr := int(rng.integers(3, 8))), nr) if kind == 4 else P.spring_mass()
            nk = f.nk(); Nn = len(nk)
            nx = np.full(Nn, f.nx, dtype=np.int32); nu = np.where(nk > 0, f.nu, 0).astype(np.int32)
            qp = capi.TreeQp(nx, nu, nk).fill_lti(f)
            opts = {}
            f.name = f"lti kind {kind} Nn={Nn}"
        else:
            f = make(kind, seed, rng)
            nk, nx, nu = np.asarray(f.nk), np.asarray(f.nx), np.asarray(f.nu)
            Nn = len(nk)
            opts = dict(f.opts) if getattr(f, "opts", None) else {}
            qp = capi.TreeQp(nx, nu, nk).set_flat(f)
        s = capi.TdunesSolver(qp, **{k: v for k, v in opts.items()})
        xoff = np.concatenate([[0], np.cumsum(nx)]); uoff = np.concatenate([[0], np.cumsum(nu)])
        lam_prev = None
        trail = []
        for step in range(n_steps):
            change = int(rng.integers(0, 7)) if step > 0 else 0
            flat = qp.flat()
            if change == 6 and x0 is not None:
                qp.set_x0(x0 * float(rng.uniform(0.3, 1.3)) + 0.02 * rng.standard_normal(len(x0)))
            trail.append(change)
            last_touch = None
            # --- apply the change through the reference's setters
            if change in (1, 5):
                e = int(rng.integers(0, Nn - 1)); k = e + 1
                dad = P.parents_of(nk)
                ao = int(sum(int(nx[j]) * int(nx[dad[j]]) for j in range(1, k))); bo = int(sum(int(nx[j]) * int(nu[dad[j]]) for j in range(1, k)))
                na, nb = int(nx[k]) * int(nx[dad[k]]), int(nx[k]) * int(nu[dad[k]])
                A = flat["A"][ao:ao + na].copy(); B = flat["B"][bo:bo + nb].copy(); b = flat["b"][int(xoff[k] - nx[0]):int(xoff[k] - nx[0] + nx[k])].copy()
                if change == 1:
                    b = b + 0.05 * rng.standard_normal(len(b))
                else:
                    A = A * (1.0 + 0.03 * rng.standard_normal(len(A))); B = B * (1.0 + 0.03 * rng.standard_normal(len(B)))
                qp.set_edge_dynamics(e, A, B, b)
            elif change in (2, 3):
                k = int(rng.integers(0, Nn))
                Qd = flat["Qd"][xoff[k]:xoff[k + 1]].copy(); Rd = flat["Rd"][uoff[k]:uoff[k + 1]].copy()
                qv = flat["q"][xoff[k]:xoff[k + 1]].copy(); rv = flat["r"][uoff[k]:uoff[k + 1]].copy()
                if change == 2:
                    qv = qv + 0.1 * rng.standard_normal(len(qv)); rv = rv + 0.1 * rng.standard_normal(len(rv))
                else:
                    Qd = Qd * float(rng.uniform(0.5, 2.0)); Rd = Rd * float(rng.uniform(0.5, 2.0))
                qp.set_node_objective_diag(k, Qd, Rd, qv, rv)
            elif change == 4:
                k = int(rng.integers(1, Nn))
                xl = flat["xmin"][xoff[k]:xoff[k + 1]].copy(); xu = flat["xmax"][xoff[k]:xoff[k + 1]].copy()
                ul = flat["umin"][uoff[k]:uoff[k + 1]].copy(); uu = flat["umax"][uoff[k]:uoff[k + 1]].copy()
                sc = float(rng.uniform(0.6, 1.5))
                fin = lambda v: np.where(np.abs(v) < 1e10, v * sc, v)
                qp.set_node_bounds(k, fin(xl), fin(xu), fin(ul), fin(uu))
                last_touch = f"bounds of node {k} x {sc:.3f}: x in [{fin(xl)}, {fin(xu)}], u in [{fin(ul)}, {fin(uu)}]"
            flat = qp.flat()
            warm = lam_prev is not None and rng.random() < 0.5
            lam0 = lam_prev if warm else (f.lambda0 if f.lambda0 is not None else np.zeros(int(np.sum(nx[1:]))))
            s.set_dual_initialization(lam0)
            st = s.solve()
            ref = orc.solve(flat, orc.default_opts(**opts), lambda0=lam0)
            sol = qp.solution()
            gp = int(capi.lib().tqgpu_uses_fused_path(C.c_void_p(s.work.device)))
            by_path[gp] = by_path.get(gp, 0) + 1
            stats["solves"] += 1
            # (every array relative to its own largest entry: a nearly infeasible problem has duals of 1e5 and more)
            err = max(float(np.max(np.abs(sol[k] - ref[k]))) / max(1.0, float(np.max(np.abs(ref[k])))) if len(ref[k]) else 0.0 for k in ("x", "u", "lam"))
            st_ref = ref["status"]
            same = (st == st_ref or (st == 1 and ref["iter"] == opts.get("maxIter", 100))) and qp.info["iter"] == ref["iter"]
            if st_ref != 0:
                # the change made the problem infeasible (a tightened state bound the dynamics cannot meet): the dual iteration diverges, in the
                # oracle as on the device, and two diverging runs share their verdict and nothing else
                stats["ill"] = stats.get("ill", 0) + 1
                if not (st == st_ref and qp.info["iter"] == ref["iter"]):
                    stats["fail"] += 1
                    print(f"MISMATCH (verdict of a diverging run) seed {seed} step {step} path {gp}: device status {st} iter {qp.info['iter']} oracle {st_ref} / {ref['iter']} [{f.name}]", flush=True)
            elif not (same and err < 1e-9):
                # the same rounding-level class as tools/fuzz_parity.py: same optimum, endgame at the tolerance
                if st == 0 and st_ref == 0 and err < 1e-5:
                    stats["tie"] += 1
                    print(f"  (rounding-level endgame: seed {seed} step {step} change {change} warm {warm}: device iterations {qp.info['iter']} oracle {ref['iter']}, difference {err:.1e})", flush=True)
                else:
                    stats["fail"] += 1
                    for key_, off_ in (("x", xoff), ("u", uoff)):
                        dv = np.abs(sol[key_] - ref[key_])
                        if len(dv) and dv.max() > 1e-9:
                            j_ = int(np.argmax(dv)); node_ = int(np.searchsorted(off_, j_, side="right") - 1)
                            print(f"    largest difference in {key_}: entry {j_ - int(off_[node_])} of node {node_} (children {int(nk[node_])}): device {sol[key_][j_]:.9g} oracle {ref[key_][j_]:.9g}; last change touched {last_touch}", flush=True)
                    dl = np.abs(sol["lam"] - ref["lam"])
                    if len(dl) and dl.max() > 1e-9:
                        j_ = int(np.argmax(dl)); loff = xoff[1:] - nx[0]; node_ = int(np.searchsorted(loff, j_, side="right"))
                        kx = slice(int(xoff[node_]), int(xoff[node_ + 1]))
                        print(f"    largest difference in lam: entry {j_ - int(loff[node_ - 1])} of the duals of node {node_} ({int((dl > 1e-9).sum())} entries differ): device {sol['lam'][j_]:.9g} oracle {ref['lam'][j_]:.9g}; "
                              f"x of that node {np.array2string(ref['x'][kx], precision=6)} in [{flat['xmin'][kx][0]:.6g}, {flat['xmax'][kx][0]:.6g}]; KKT residual of the device's solution {qp.max_kkt_res():.2e}, "
                              f"of the oracle's {float(orc.max_kkt(flat, ref)):.2e}", flush=True)
                    print(f"MISMATCH seed {seed} step {step} changes so far {trail} warm {warm} path {gp}: device status {st} iter {qp.info['iter']} oracle {st_ref} / {ref['iter']} err {err:.2e} [{f.name}]", flush=True)
            lam_prev = sol["lam"].copy()
        s.destroy()
        if q % 10 == 9:
            print(f"  {q + 1} sequences, {stats['solves']} solves, {stats['fail']} mismatches, {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"{n_seq} sequences (seeds {s0}..{s0 + n_seq - 1}) of {n_steps} solves through treeqp_tdunes_solve, {stats['solves']} solves on device paths {dict(sorted(by_path.items()))}: "
          f"{stats['fail']} mismatches; {stats['tie']} rounding-level endgames (same optimum); {stats.get('ill', 0)} solves of problems a change had made infeasible (the oracle does not converge either: same verdict and iteration count required, nothing else)")
    return stats


if __name__ == "__main__":
    st_ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 7000, int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    sys.exit(1 if st_["fail"] else 0)
