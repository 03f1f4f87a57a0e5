"""Shared helpers for the parity tests (test infrastructure; may use the oracle)."""
from __future__ import annotations

import numpy as np

from treeqp_amd import problems as P


def lti_dims(p):
    nk = p.nk()
    return np.full(p.Nn, p.nx, dtype=np.int32), np.where(nk > 0, p.nu, 0).astype(np.int32), nk


def oracle_flat_from_lti(orc, p):
    return orc.fill_lti_diag(p.nk(), p.nx, p.nu, p.A, p.B, p.b, p.Qd, p.q, p.Pd, p.p, p.Rd, p.r,
                             p.xmin, p.xmax, p.umin, p.umax, p.x0)


def product_qp_from_lti(capi, p, eliminate_x0=False):
    nx, nu, nk = lti_dims(p)
    qp = capi.TreeQp(nx, nu, nk).fill_lti(p)
    if eliminate_x0:
        qp.eliminate_x0()
    return qp


def product_qp_from_flat(capi, f):
    return capi.TreeQp(f.nx, f.nu, f.nk).set_flat(f)


def oracle_opts(orc, opts: dict):
    return orc.default_opts(**opts)


def rel_err(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(1.0, float(np.max(np.abs(b)))))


def assert_solution_close(got: dict, ref: dict, tol=1e-10, keys=("x", "u", "lam", "mu_x", "mu_u")):
    for k in keys:
        e = rel_err(got[k], ref[k])
        assert e <= tol, f"{k}: relative error {e:.3e} > {tol:.1e}"


def flat_to_json(flat: dict, options: dict | None = None) -> dict:
    """Flat ("ltv" order) clipping QP -> the qp_in.json wire format of the reference's JSON front end
    (examples/solve_qp_json.cpp): nodes {Q,R,S,q,r,lx,lu,ux,uu}, edges {from,to,A,B,b}, matrices as arrays of rows."""
    nk, nx, nu = [np.asarray(flat[k], dtype=int) for k in ("nk", "nx", "nu")]
    Nn = len(nk)
    dad = np.full(Nn, -1)
    c = 1
    for k in range(Nn):
        for _ in range(nk[k]):
            dad[c] = k
            c += 1
    xo, uo = np.concatenate([[0], np.cumsum(nx)]), np.concatenate([[0], np.cumsum(nu)])
    rows = lambda v, m, n: np.asarray(v, dtype=float).reshape((m, n), order="F").tolist()
    nodes, edges = [], []
    for k in range(Nn):
        sx, su = slice(xo[k], xo[k + 1]), slice(uo[k], uo[k + 1])
        nodes.append(dict(Q=np.diag(flat["Qd"][sx]).tolist(), R=np.diag(flat["Rd"][su]).tolist(),
                          S=np.zeros((nu[k], nx[k])).tolist(), q=list(map(float, flat["q"][sx])), r=list(map(float, flat["r"][su])),
                          lx=list(map(float, flat["xmin"][sx])), ux=list(map(float, flat["xmax"][sx])),
                          lu=list(map(float, flat["umin"][su])), uu=list(map(float, flat["umax"][su]))))
    ao = bo = lo = 0
    for k in range(1, Nn):
        p = dad[k]
        na, nb = nx[k] * nx[p], nx[k] * nu[p]
        edges.append({"from": int(p), "to": int(k), "A": rows(flat["A"][ao:ao + na], nx[k], nx[p]),
                      "B": rows(flat["B"][bo:bo + nb], nx[k], nu[p]), "b": list(map(float, flat["b"][lo:lo + nx[k]]))})
        ao += na; bo += nb; lo += nx[k]
    out = dict(nodes=nodes, edges=edges)
    if options is not None:
        out["options"] = options
    return out


def fuzz_case(seed, s0=20000):
    """Problem and options of case `seed` of the parity campaign `python tools/fuzz_parity.py N s0` (the campaign draws its parameters
    from ONE generator seeded with s0, case after case: replayed here without building the cases before `seed`)."""
    import numpy as np
    from treeqp_amd import problems as P
    rng = np.random.Generator(np.random.PCG64(s0))
    for c in range(seed - s0 + 1):
        sd = s0 + c
        kind = c % 4
        last = sd == seed
        if kind == 0:
            a = dict(depth=int(rng.integers(2, 6)), max_kids=int(rng.integers(2, 5)), nx_range=(1, int(rng.integers(2, 9))), nu_range=(1, int(rng.integers(1, 5))), ubound=float(rng.choice([0.1, 0.3, 1.0])))
            f = P.random_shape_qp(sd, **a) if last else None
        elif kind == 1:
            nh = int(rng.integers(4, 11))
            f = P.pruned_chain_qp(Nh=nh, seed=sd) if last else None
        elif kind == 2:
            md = int(rng.integers(1, 4)); Nr = int(rng.integers(1, 5)); Nh = Nr + int(rng.integers(0, 4))
            a = dict(nx=int(rng.choice([2, 4, 8])), nu=int(rng.integers(1, 4)), md=md, Nr=Nr, Nh=Nh, ubound=float(rng.choice([0.2, 0.4, 2.0])))
            f = P.random_uniform_tree_qp(sd, **a) if last else None
        else:
            a = dict(depth=int(rng.integers(2, 4)), max_kids=3, nx_range=(6, 14), nu_range=(2, 6), ubound=float(rng.choice([0.2, 0.5])))
            f = P.random_shape_qp(sd, **a) if last else None
        tc, rt = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    opts = dict(f.opts) if getattr(f, "opts", None) else {}
    opts.update(termCondition=tc, regType=rt)
    if tc == 0:
        opts["stationarityTolerance"] = 1e-12
    if rt == 1:
        opts["regValue"] = 1e-8
    return f, opts


def ulp_sensitivity(orc, f, opts, upto, copies=6):
    """Does the ORACLE keep its own line-search decisions when every non-zero of the problem data moves by one unit in the last place?
    Returns the number of perturbed copies (of `copies`) whose trial counts differ from the unperturbed oracle run in an iteration
    <= `upto` (or whose verdict differs).  A case with a positive count is decided by rounding at or before that iteration in ANY
    implementation of the algorithm: the parity campaign lists a device / oracle difference there apart from a mismatch."""
    import numpy as np
    d = f.as_dict()
    base = orc.solve(d, orc.default_opts(**opts), lambda0=f.lambda0)
    rng = np.random.default_rng(12345)
    moved = 0
    for _ in range(copies):
        d2 = dict(d)
        for k in ("A", "B", "b", "Qd", "Rd", "q", "r"):
            a = np.array(d2[k], dtype=np.float64, copy=True)
            a *= 1.0 + (rng.integers(0, 2, a.shape) * 2 - 1) * 2.0 ** -52
            d2[k] = a
        p = orc.solve(d2, orc.default_opts(**opts), lambda0=f.lambda0)
        n = min(upto + 1, base["iter"], p["iter"])
        same = p["status"] == base["status"] and all(int(p["trace_ls"][k]) == int(base["trace_ls"][k]) for k in range(n))
        if n < upto + 1 and p["iter"] != base["iter"]:
            same = False
        moved += 0 if same else 1
    return moved


def ulp_solution_spread(orc, f, opts, copies=4):
    """How far the ORACLE's solution moves (largest relative change over x, u, lambda, each relative to its own largest entry) when every
    non-zero of the problem data moves by one unit in the last place: the conditioning of the run as the oracle takes it -- a device /
    oracle difference of that size or below, with equal verdict and counts, is rounding in an ill-conditioned Newton system, not an error."""
    import numpy as np
    d = f.as_dict()
    base = orc.solve(d, orc.default_opts(**opts), lambda0=f.lambda0)
    rng = np.random.default_rng(54321)
    spread = 0.0
    for _ in range(copies):
        d2 = dict(d)
        for k in ("A", "B", "b", "Qd", "Rd", "q", "r"):
            a = np.array(d2[k], dtype=np.float64, copy=True)
            a *= 1.0 + (rng.integers(0, 2, a.shape) * 2 - 1) * 2.0 ** -52
            d2[k] = a
        p = orc.solve(d2, orc.default_opts(**opts), lambda0=f.lambda0)
        for k in ("x", "u", "lam"):
            if len(base[k]):
                with np.errstate(invalid="ignore"):
                    v = float(np.nanmax(np.abs(p[k] - base[k]))) / max(1.0, float(np.nanmax(np.abs(base[k]))))
                spread = max(spread, v)
    return spread
