"""Oracle-independent golden solutions for the CLIPPING branch of tdunes (tests/golden/kkt_pin_*.npz).

The reference holds no golden OUTPUTS for a clipping run, so the oracle's phases S and L were pinned
only by the drivers' KKT asserts.  This script produces the missing vectors without running one line of the oracle,
of the product or of the reference's algorithm: a tree QP with diagonal weights and box bounds is a strictly convex QP,
its solution is unique, and a point is THE solution iff it satisfies the KKT conditions.  So:

  1. the QP is built here in numpy (multistage tree numbering, the LTI fill of tree_qp_common.c:1837-1949 including its
     integer-division stage scaling -- restated from the reference's text, not from oracle/);
  2. an active set is found by a primal active-set iteration on the sparse KKT system (scipy.sparse.linalg.spsolve +
     two steps of iterative refinement): fix the active bounds, solve the equality-constrained QP, add violated bounds /
     drop bounds whose multiplier has the wrong sign, repeat until neither happens;
  3. the final point is checked against EVERY KKT condition (stationarity, dynamics, primal and dual feasibility,
     complementarity, strict complementarity margin) to 1e-11; only then it is stored.

Multiplier conventions (tree_qp_common.c:540-788, as every KKT check of the reference uses them):
    Q x_k + q_k + mu_x,k - lam_k + sum_kids A_kid' lam_kid = 0,     R u_k + r_k + mu_u,k + sum_kids B_kid' lam_kid = 0,
    x_k = A_k x_dad + B_k u_dad + b_k,     mu > 0 only at an upper bound, mu < 0 only at a lower bound.

Run in the build container: python tools/make_kkt_pins.py   (no GPU, no oracle, no reference needed: pure data + numpy)
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from treeqp_amd import problems as P          # noqa: E402  (problem DATA generators only: numpy, no native code)

OUT = ROOT / "tests" / "golden"


def tree_from_nk(nk):
    """BFS numbering (tree.c:171-243): kids of node i are the next unassigned contiguous indices."""
    Nn = len(nk)
    dad = np.full(Nn, -1)
    stage = np.zeros(Nn, dtype=int)
    real = np.full(Nn, -1)
    cur = 1
    for i in range(Nn):
        for c in range(nk[i]):
            dad[cur + c] = i
            stage[cur + c] = stage[i] + 1
            # `real`: child ordinal if the dad branches, else inherited (0 under a non-branching root) (tree.c:224-238)
            real[cur + c] = c if nk[i] > 1 else (real[i] if i > 0 else 0)
        cur += nk[i]
    assert cur == Nn
    return dad, stage, real


def flat_from_lti(p):
    """tree_qp_in_fill_lti_data_diag_weights (tree_qp_common.c:1837-1949), numpy restatement."""
    nk = np.asarray(p.nk(), dtype=int)
    Nn, nx, nu = len(nk), p.nx, p.nu
    dad, stage, real = tree_from_nk(nk)
    nxv = np.full(Nn, nx)
    nuv = np.where(nk > 0, nu, 0)
    leaves = 1
    for ii in range(Nn - 1, 0, -1):
        if stage[ii] == stage[ii - 1]:
            leaves += 1
        else:
            break
    A = np.concatenate([np.asarray(p.A)[real[k] * nx * nx:(real[k] + 1) * nx * nx] for k in range(1, Nn)])
    B = np.concatenate([np.asarray(p.B)[real[k] * nx * nu:(real[k] + 1) * nx * nu] for k in range(1, Nn)])
    b = np.concatenate([np.asarray(p.b)[real[k] * nx:(real[k] + 1) * nx] for k in range(1, Nn)])
    Qd = np.concatenate([np.asarray(p.Qd if nk[k] > 0 else p.Pd, dtype=float) for k in range(Nn)])
    q = np.concatenate([np.asarray(p.q if nk[k] > 0 else p.p, dtype=float) for k in range(Nn)])
    Rd = np.concatenate([np.asarray(p.Rd, dtype=float) for k in range(Nn) if nk[k] > 0])
    r = np.concatenate([np.asarray(p.r, dtype=float) for k in range(Nn) if nk[k] > 0])
    xo = np.concatenate([[0], np.cumsum(nxv)])
    uo = np.concatenate([[0], np.cumsum(nuv)])
    # stage scaling: every COMPLETED stage is multiplied by numberOfLeaves / nodesInStage, an INTEGER division (:1911);
    # the last stage is never scaled (the loop only scales when it meets the next stage)
    cur_stage, in_stage = 0, 0
    for ii in range(Nn):
        if stage[ii] > cur_stage:
            f = float(leaves // in_stage)
            for jj in range(1, in_stage + 1):
                n = ii - jj
                Qd[xo[n]:xo[n + 1]] *= f
                q[xo[n]:xo[n + 1]] *= f
                Rd[uo[n]:uo[n + 1]] *= f
                r[uo[n]:uo[n + 1]] *= f
            cur_stage, in_stage = stage[ii], 1
        else:
            in_stage += 1
    xmin = np.concatenate([np.asarray(p.x0 if k == 0 else p.xmin, dtype=float) for k in range(Nn)])
    xmax = np.concatenate([np.asarray(p.x0 if k == 0 else p.xmax, dtype=float) for k in range(Nn)])
    umin = np.concatenate([np.asarray(p.umin, dtype=float) for k in range(Nn) if nk[k] > 0])
    umax = np.concatenate([np.asarray(p.umax, dtype=float) for k in range(Nn) if nk[k] > 0])
    return dict(nk=nk, nx=nxv, nu=nuv, A=A, B=B, b=b, Qd=Qd, Rd=Rd, q=q, r=r, xmin=xmin, xmax=xmax, umin=umin, umax=umax)


def kkt_pieces(f):
    """Sparse pieces of the KKT system of a flat tree QP: H z + g, dynamics G z = -b (z = [x; u])."""
    nk, nx, nu = f["nk"], f["nx"], f["nu"]
    Nn = len(nk)
    dad, _, _ = tree_from_nk(nk)
    xo = np.concatenate([[0], np.cumsum(nx)])
    uo = np.concatenate([[0], np.cumsum(nu)])
    SX, SU = int(xo[-1]), int(uo[-1])
    n0 = int(nx[0])
    nl = SX - n0
    rows, cols, vals = [], [], []
    ao = bo = 0
    for k in range(1, Nn):
        p = dad[k]
        r0 = xo[k] - n0
        Ak = f["A"][ao:ao + nx[k] * nx[p]].reshape((nx[k], nx[p]), order="F")
        Bk = f["B"][bo:bo + nx[k] * nu[p]].reshape((nx[k], nu[p]), order="F")
        ao += nx[k] * nx[p]
        bo += nx[k] * nu[p]
        for i in range(nx[k]):
            for j in range(nx[p]):
                if Ak[i, j] != 0.0:
                    rows.append(r0 + i); cols.append(xo[p] + j); vals.append(Ak[i, j])
            for j in range(nu[p]):
                if Bk[i, j] != 0.0:
                    rows.append(r0 + i); cols.append(SX + uo[p] + j); vals.append(Bk[i, j])
            rows.append(r0 + i); cols.append(xo[k] + i); vals.append(-1.0)
    G = sp.csr_matrix((vals, (rows, cols)), shape=(nl, SX + SU))
    h = np.concatenate([f["Qd"], f["Rd"]])
    g = np.concatenate([f["q"], f["r"]])
    lo = np.concatenate([f["xmin"], f["umin"]])
    hi = np.concatenate([f["xmax"], f["umax"]])
    return G, h, g, lo, hi, SX, SU, nl


def solve_equality_qp(G, h, g, b, fixed, value):
    """min 1/2 z'Hz + g'z  s.t.  G z = -b, z[fixed] = value[fixed].  Returns z, lam (dynamics), mu (bounds, full length)."""
    n = len(h)
    free = np.flatnonzero(~fixed)
    fx = np.flatnonzero(fixed)
    # eliminate the fixed entries: G_f z_f = -b - G_x v
    Gf = G[:, free].tocsc()
    rhs_dyn = -b - G[:, fx] @ value[fx]
    nf, nl = len(free), G.shape[0]
    # the reference's sign: stationarity is H z + g + mu + G' lam = 0 (with the -lam_k term inside G' since G holds -I)
    K = sp.bmat([[sp.diags(h[free]), Gf.T], [Gf, None]], format="csc")
    rhs = np.concatenate([-g[free], rhs_dyn])
    lu = spla.splu(K)
    sol = lu.solve(rhs)
    for _ in range(3):                                    # iterative refinement (residual in extended precision)
        res = rhs.astype(np.longdouble) - (K @ sol).astype(np.longdouble)
        sol = sol + lu.solve(np.asarray(res, dtype=np.float64))
    z = value.copy()
    z[free] = sol[:nf]
    lam = sol[nf:]
    mu = np.zeros(n)
    stat = h * z + g + G.T @ lam
    mu[fx] = -stat[fx]
    return z, lam, mu


def active_set_solve(f, name, max_rounds=200):
    G, h, g, lo, hi, SX, SU, nl = kkt_pieces(f)
    b = f["b"]
    n = SX + SU
    at_lo = np.zeros(n, dtype=bool)
    at_hi = np.zeros(n, dtype=bool)
    eq = lo == hi                                          # root state pinned to x0
    at_hi[eq] = True
    for rnd in range(max_rounds):
        fixed = at_lo | at_hi
        value = np.where(at_hi, hi, np.where(at_lo, lo, 0.0))
        z, lam, mu = solve_equality_qp(G, h, g, b, fixed, value)
        viol_hi = (~fixed) & (z > hi + 1e-13)
        viol_lo = (~fixed) & (z < lo - 1e-13)
        wrong_hi = at_hi & ~eq & (mu < -1e-13)
        wrong_lo = at_lo & ~eq & (mu > 1e-13)
        if not (viol_hi.any() or viol_lo.any() or wrong_hi.any() or wrong_lo.any()):
            break
        at_hi |= viol_hi
        at_lo |= viol_lo
        at_hi &= ~wrong_hi
        at_lo &= ~wrong_lo
    else:
        raise RuntimeError(f"{name}: active-set iteration did not settle")
    # ---- full KKT verification of the point that will be stored ----
    stat = h * z + g + mu + G.T @ lam
    dyn = G @ z + b
    comp = np.where(mu > 0, mu * (z - hi), mu * (lo - z))
    inactive = ~(at_lo | at_hi)
    margin_primal = np.min(np.minimum(z[inactive] - lo[inactive], hi[inactive] - z[inactive])) if inactive.any() else np.inf
    act = (at_lo | at_hi) & ~eq
    margin_dual = np.min(np.abs(mu[act])) if act.any() else np.inf
    checks = dict(stationarity=np.max(np.abs(stat)), dynamics=np.max(np.abs(dyn)),
                  bound_violation=max(np.max(z - hi), np.max(lo - z), 0.0),
                  wrong_sign=max(np.max(-mu[at_hi & ~eq], initial=0.0), np.max(mu[at_lo & ~eq], initial=0.0)),
                  complementarity=np.max(np.abs(comp)))
    scale = max(1.0, np.max(np.abs(z)), np.max(np.abs(lam)))
    print(f"{name}: rounds {rnd + 1}, active bounds {int(act.sum())} (+{int(eq.sum())} pinned), "
          + ", ".join(f"{k} {v:.2e}" for k, v in checks.items())
          + f", strict-complementarity margins: primal {margin_primal:.2e}, dual {margin_dual:.2e}")
    assert all(v <= 1e-11 * scale for v in checks.values()), checks
    assert margin_primal > 1e-9 and margin_dual > 1e-9, "degenerate active set: the pin would not be unique to 1e-10"
    return dict(x=z[:SX], u=z[SX:], lam=lam, mu_x=mu[:SX], mu_u=mu[SX:], n_active=int(act.sum()))


def main():
    OUT.mkdir(parents=True, exist_ok=True)
    cases = {
        "c1_spring_mass": flat_from_lti(P.spring_mass()),                  # BASELINE C1: the reference's example (85 nodes)
        "c1_depth4": flat_from_lti(P.spring_mass(Nh=4)),                   # "depth 4" plumbing config (31 nodes)
        "c2_linear_chain": flat_from_lti(P.linear_chain(2, 9, 9)),         # BASELINE C2 (1023 nodes)
        "thesis_example": P.thesis_example().as_dict(),                    # examples/thesis_example.c (irregular 6-node tree)
        "irregular": P.irregular_clipping_qp().as_dict(),                  # SURVEY §8c probe (ii): per-node nx, nu
    }
    for name, f in cases.items():
        f = {k: np.asarray(v) for k, v in f.items()}
        sol = active_set_solve(f, name)
        np.savez_compressed(OUT / f"kkt_pin_{name}.npz", **{k: np.asarray(v) for k, v in f.items()},
                            **{"sol_" + k: np.asarray(v) for k, v in sol.items()})
    print("wrote", sorted(p.name for p in OUT.glob("kkt_pin_*.npz")))


if __name__ == "__main__":
    main()
