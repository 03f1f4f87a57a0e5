"""Synthetic inputs for the tdunes hot path (BASELINE.json configs).

Pure numpy/scipy, no solver code.  Two kinds of descriptions are produced:

* ``LtiProblem``  -- LTI data replicated over a multistage tree by realization id; consumed by
  ``TreeQp.fill_lti`` (the product's ``tree_qp_in_fill_lti_data_diag_weights``) and, in tests, by
  the oracle's restatement of the same filler.
* ``FlatProblem`` -- per-edge / per-node arrays in the "ltv" order of the C API
  (tree_qp_in_set_ltv_dynamics_colmajor etc.).

Every generator is deterministic (the C2/C3 matrices involve no RNG at all; C4/C5 use a seeded
PCG64 stream -- MATLAB's ``rng`` stream of the reference's generate_random_tree.m cannot be
reproduced, only its distribution).
"""
from __future__ import annotations

import json
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent.parent / "tests" / "golden"
INF = 1e12


@dataclass
class LtiProblem:
    name: str
    md: int
    Nr: int
    Nh: int
    nx: int
    nu: int
    A: np.ndarray      # n_real * nx*nx, column major per realization
    B: np.ndarray
    b: np.ndarray
    Qd: np.ndarray
    q: np.ndarray
    Pd: np.ndarray
    p: np.ndarray
    Rd: np.ndarray
    r: np.ndarray
    xmin: np.ndarray
    xmax: np.ndarray
    umin: np.ndarray
    umax: np.ndarray
    x0: np.ndarray
    lambda0: np.ndarray | None = None
    opts: dict = field(default_factory=dict)
    expect: dict = field(default_factory=dict)

    @property
    def Nn(self) -> int:
        return number_of_nodes(self.md, self.Nr, self.Nh)

    def nk(self) -> np.ndarray:
        return multistage_nk(self.md, self.Nr, self.Nh)


@dataclass
class FlatProblem:
    name: str
    nk: np.ndarray
    nx: np.ndarray
    nu: np.ndarray
    A: np.ndarray
    B: np.ndarray
    b: np.ndarray
    Qd: np.ndarray
    Rd: np.ndarray
    q: np.ndarray
    r: np.ndarray
    xmin: np.ndarray
    xmax: np.ndarray
    umin: np.ndarray
    umax: np.ndarray
    lambda0: np.ndarray | None = None
    opts: dict = field(default_factory=dict)
    expect: dict = field(default_factory=dict)

    def as_dict(self) -> dict:
        return {k: getattr(self, k) for k in ("nk", "nx", "nu", "A", "B", "b", "Qd", "Rd", "q", "r", "xmin", "xmax", "umin", "umax")}


# ------------------------------------------------------------------------------------------
# tree shapes (plain integer arithmetic; the C implementations are tested against these)
# ------------------------------------------------------------------------------------------

def number_of_nodes(md: int, Nr: int, Nh: int) -> int:
    if md == 1:
        return Nh + 1
    return (Nh - Nr) * md ** Nr + (md ** (Nr + 1) - 1) // (md - 1)


def multistage_nk(md: int, Nr: int, Nh: int) -> np.ndarray:
    nk = []
    width = 1
    for stage in range(Nh):
        fan = md if stage < Nr else 1
        nk += [fan] * width
        width *= fan
    nk += [0] * width
    return np.asarray(nk, dtype=np.int32)


def parents_of(nk) -> np.ndarray:
    nk = np.asarray(nk)
    dad = np.full(len(nk), -1, dtype=np.int64)
    cursor = 1
    for i, c in enumerate(nk):
        dad[cursor:cursor + c] = i
        cursor += c
    return dad


# ------------------------------------------------------------------------------------------
# C1: the reference's spring-mass example data (examples/spring_mass_utils, via tests/golden)
# ------------------------------------------------------------------------------------------

def spring_mass(Nh: int | None = None, Nr: int | None = None, md: int | None = None,
                xmax1: float | None = None) -> LtiProblem:
    """examples/spring_mass_dual_newton_tree.c:52-129 (defaults md=3, Nr=2, Nh=10 -> 85 nodes).
    `xmax1=0.2` gives the variant of examples/spring_mass.c:124."""
    d = json.loads((GOLDEN / "spring_mass_data.json").read_text())
    NX, NU = d["NX"], d["NU"]
    md = d["md"] if md is None else md
    Nr = d["Nr"] if Nr is None else Nr
    Nh = d["Nh"] if Nh is None else Nh
    A = np.asarray(d["A"])[NX * NX:]       # the driver skips the nominal realization (:101)
    B = np.asarray(d["B"])[NX * NU:]
    b = np.asarray(d["b"])[NX:]
    assert md <= len(A) // (NX * NX)
    xmax = np.asarray(d["xmax"], dtype=float).copy()
    if xmax1 is not None:
        xmax[1] = xmax1
    Nn = number_of_nodes(md, Nr, Nh)
    lam0 = np.asarray(d["lambda0_tree"], dtype=float)
    lam0 = np.resize(lam0, (Nn - 1) * NX) if len(lam0) < (Nn - 1) * NX else lam0[:(Nn - 1) * NX]
    return LtiProblem(
        name=f"spring_mass(md={md},Nr={Nr},Nh={Nh})", md=md, Nr=Nr, Nh=Nh, nx=NX, nu=NU, A=A, B=B, b=b,
        Qd=np.asarray(d["dQ"], float), q=np.asarray(d["q"], float), Pd=np.asarray(d["dP"], float),
        p=np.asarray(d["p"], float), Rd=np.asarray(d["dR"], float), r=np.asarray(d["r"], float),
        xmin=np.asarray(d["xmin"], float), xmax=xmax, umin=np.asarray(d["umin"], float),
        umax=np.asarray(d["umax"], float), x0=np.asarray(d["x0"], float), lambda0=lam0)


# ------------------------------------------------------------------------------------------
# C2 / C3: the reference's linear_chain benchmark model (nm masses, nx = 2 nm, nu = nm - 1)
# benchmark/linear_chain/initialize_linear_chain.m:39-75, default_params_linear_chain.m:19-23,
# utils/discretize_model.m:8-10
# ------------------------------------------------------------------------------------------

def _zoh(Ac: np.ndarray, Bc: np.ndarray, Ts: float):
    from scipy.linalg import expm
    n, m = Bc.shape
    M = np.zeros((n + m, n + m))
    M[:n, :n] = Ts * Ac
    M[:n, n:] = Ts * Bc
    E = expm(M)
    return E[:n, :n], E[:n, n:]


def linear_chain(md: int = 2, Nr: int = 9, Nh: int = 9, nm: int = 4, nu: int | None = None,
                 Ts: float = 0.05, ubound: float = 0.5) -> LtiProblem:
    """C2 = linear_chain(2, 9, 9) -> 1023 nodes; C3 = linear_chain(2, 11, 11) -> 4095 nodes.
    |u| <= 0.5 (instead of the model's 2.0) so that the active set is non-trivial."""
    nu = nm - 1 if nu is None else nu
    nx = 2 * nm
    Tm = -2.0 * np.eye(nm) + np.eye(nm, k=1) + np.eye(nm, k=-1)
    Bc = np.zeros((nx, nu))
    Bc[nm:nm + nu, :] = np.eye(nu)
    As, Bs = [], []
    for k in np.linspace(4.0, 8.0, md):
        Ac = np.zeros((nx, nx))
        Ac[:nm, nm:] = np.eye(nm)
        Ac[nm:, :nm] = k * Tm
        Ad, Bd = _zoh(Ac, Bc, Ts)
        As.append(Ad.flatten(order="F"))
        Bs.append(Bd.flatten(order="F"))
    x0 = np.zeros(nx)
    x0[nx - 1] = 2.0
    Nn = number_of_nodes(md, Nr, Nh)
    return LtiProblem(
        name=f"linear_chain(nx={nx},nu={nu},md={md},Nr={Nr},Nh={Nh})", md=md, Nr=Nr, Nh=Nh, nx=nx, nu=nu,
        A=np.concatenate(As), B=np.concatenate(Bs), b=np.zeros(md * nx),
        Qd=10.0 * np.ones(nx), q=np.zeros(nx), Pd=10.0 * np.ones(nx), p=np.zeros(nx),
        Rd=np.ones(nu), r=np.zeros(nu), xmin=-2.0 * np.ones(nx), xmax=2.0 * np.ones(nx),
        umin=-ubound * np.ones(nu), umax=ubound * np.ones(nu), x0=x0, lambda0=np.zeros((Nn - 1) * nx))


# ------------------------------------------------------------------------------------------
# C4: random clipping QP (examples/random_qp_utils/generate_random_tree.m:52-80, CLIPPING=true)
# ------------------------------------------------------------------------------------------

def random_clipping_qp(nx: int = 20, nu: int = 10, md: int = 3, levels: int = 8, seed: int = 20260101,
                       scale_A: bool = True) -> FlatProblem:
    """Unconstrained random tree QP, diagonal weights.  Draw order: edge-major A,B,b then node-major
    Q,R,q,r.  `scale_A` multiplies A by 2/nx (documented deviation, SURVEY §7: unscaled U(0,1)
    dynamics amplify by ~1e7 over 7 stages at nx=20 and exhaust double precision)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nk = multistage_nk(md, levels - 1, levels - 1)
    Nn = len(nk)
    nxv = np.full(Nn, nx, dtype=np.int32)
    nuv = np.where(nk > 0, nu, 0).astype(np.int32)
    A = rng.random((Nn - 1, nx * nx)) * ((2.0 / nx) if scale_A else 1.0)
    B = rng.random((Nn - 1, nx * nu))
    b = rng.random((Nn - 1, nx))
    Qd = 10.0 * rng.random(Nn * nx) + 1e-3
    Rd = rng.random(int(nuv.sum())) + 1e-3
    q = rng.random(Nn * nx)
    r = rng.random(int(nuv.sum()))
    return FlatProblem(
        name=f"random_clipping_qp(nx={nx},nu={nu},md={md},levels={levels},seed={seed})", nk=nk, nx=nxv, nu=nuv,
        A=A.ravel(), B=B.ravel(), b=b.ravel(), Qd=Qd, Rd=Rd, q=q, r=r,
        xmin=-INF * np.ones(Nn * nx), xmax=INF * np.ones(Nn * nx),
        umin=-INF * np.ones(int(nuv.sum())), umax=INF * np.ones(int(nuv.sum())),
        opts=dict(maxIter=10, stationarityTolerance=1e-10, regType=0))


# ------------------------------------------------------------------------------------------
# C5: pruned scenario tree in the shape class of examples/fault_tolerance.c (variable nk,
# uniform leaf depth, nx=8, nu=2), dynamics = chain model with per-realization stiffness
# ------------------------------------------------------------------------------------------

def pruned_tree_nk(Nh: int = 10, n_real: int = 3, branch_stages: int = 4, max_leaves: int = 40,
                   seed: int = 7) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    nk = []
    width = 1
    for stage in range(Nh):
        counts = []
        for _ in range(width):
            if stage < branch_stages:
                c = int(rng.integers(1, n_real + 1))
            else:
                c = 1
            counts.append(c)
        # prune so that the number of scenarios stays bounded
        while sum(counts) > max_leaves:
            i = int(np.argmax(counts))
            counts[i] -= 1
        nk += counts
        width = sum(counts)
    nk += [0] * width
    return np.asarray(nk, dtype=np.int32)


def pruned_chain_qp(Nh: int = 10, seed: int = 7, nm: int = 4, nu: int = 2, n_real: int = 3,
                    max_leaves: int = 40) -> FlatProblem:
    nx = 2 * nm
    nk = pruned_tree_nk(Nh, n_real, 4, max_leaves, seed)
    Nn = len(nk)
    dad = parents_of(nk)
    Tm = -2.0 * np.eye(nm) + np.eye(nm, k=1) + np.eye(nm, k=-1)
    Bc = np.zeros((nx, nu))
    Bc[nm:nm + nu, :] = np.eye(nu)
    reals = []
    for rr, k in enumerate(np.linspace(2.0, 6.0, n_real)):
        Ac = np.zeros((nx, nx))
        Ac[:nm, nm:] = np.eye(nm)
        Kt = k * Tm.copy()
        if rr > 0:                                  # a "faulted" (weakened) spring per realization
            Kt[rr % nm, :] *= 0.25
        Ac[nm:, :nm] = Kt
        reals.append(_zoh(Ac, Bc, 0.1))
    nuv = np.where(nk > 0, nu, 0).astype(np.int32)
    A, B = [], []
    ordinal = np.zeros(Nn, dtype=int)
    cursor = 1
    for i, c in enumerate(nk):
        ordinal[cursor:cursor + c] = np.arange(c)
        cursor += c
    for k in range(1, Nn):
        Ad, Bd = reals[ordinal[k] % n_real]
        A.append(Ad.flatten(order="F"))
        B.append(Bd.flatten(order="F"))
    su = int(nuv.sum())
    xlo = np.concatenate([-3.0 * np.ones(nm), -8.0 * np.ones(nm)])
    xhi = np.concatenate([2.5 * np.ones(nm), 8.0 * np.ones(nm)])
    x0 = np.zeros(nx)
    x0[0] = 1.5
    x0[nm - 1] = -1.0
    xmin = np.tile(xlo, Nn)
    xmax = np.tile(xhi, Nn)
    xmin[:nx] = x0
    xmax[:nx] = x0
    return FlatProblem(
        name=f"pruned_chain_qp(Nh={Nh},seed={seed},Nn={Nn})", nk=nk, nx=np.full(Nn, nx, dtype=np.int32), nu=nuv,
        A=np.concatenate(A), B=np.concatenate(B), b=np.zeros((Nn - 1) * nx),
        Qd=np.tile(np.concatenate([10.0 * np.ones(nm), np.ones(nm)]), Nn), Rd=0.1 * np.ones(su),
        q=np.zeros(Nn * nx), r=np.zeros(su), xmin=xmin, xmax=xmax,
        umin=-0.4 * np.ones(su), umax=0.4 * np.ones(su),
        opts=dict(maxIter=200, stationarityTolerance=1e-8, lineSearchMaxIter=100, lineSearchGamma=0.1,
                  lineSearchBeta=0.8, regType=1, regValue=1e-10))


# ------------------------------------------------------------------------------------------
# small irregular tree with per-node dimensions (shape class of the reference's random_qp fixtures)
# ------------------------------------------------------------------------------------------

def irregular_clipping_qp(seed: int = 3) -> FlatProblem:
    rng = np.random.Generator(np.random.PCG64(seed))
    nk = np.asarray([3, 2, 1, 2, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0], dtype=np.int32)
    Nn = len(nk)
    dad = parents_of(nk)
    nx = rng.integers(1, 4, size=Nn).astype(np.int32)
    nu = np.where(nk > 0, rng.integers(1, 3, size=Nn), 0).astype(np.int32)
    A = np.concatenate([0.9 * rng.random(nx[k] * nx[dad[k]]) for k in range(1, Nn)])
    B = np.concatenate([rng.random(nx[k] * nu[dad[k]]) for k in range(1, Nn)])
    b = 0.1 * rng.random(int(nx[1:].sum()))
    sx, su = int(nx.sum()), int(nu.sum())
    xmin = -INF * np.ones(sx)
    xmax = INF * np.ones(sx)
    x0 = rng.random(nx[0])
    xmin[:nx[0]] = x0
    xmax[:nx[0]] = x0
    return FlatProblem(
        name=f"irregular_clipping_qp(seed={seed})", nk=nk, nx=nx, nu=nu, A=A, B=B, b=b,
        Qd=1.0 + 9.0 * rng.random(sx), Rd=0.5 + rng.random(su), q=rng.random(sx) - 0.5, r=rng.random(su) - 0.5,
        xmin=xmin, xmax=xmax, umin=-0.3 * np.ones(su), umax=0.3 * np.ones(su))


def random_shape_qp(seed: int, depth: int = 3, max_kids: int = 3, nx_range=(1, 4), nu_range=(1, 3), ubound: float = 0.3) -> FlatProblem:
    """Random tree SHAPE (every non-leaf has 1..max_kids children, all leaves at `depth`) with per-node nx / nu drawn from the
    given ranges (inclusive) and seeded random data: the shape class of the irregular probe of SURVEY section 8c, scaled up so
    that blocks of any dimension (and parents with more than four children) occur."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nk, width = [], 1
    for _ in range(depth):
        counts = [int(rng.integers(1, max_kids + 1)) for _ in range(width)]
        nk += counts
        width = sum(counts)
    nk += [0] * width
    nk = np.asarray(nk, dtype=np.int32)
    Nn = len(nk)
    dad = parents_of(nk)
    nx = rng.integers(nx_range[0], nx_range[1] + 1, size=Nn).astype(np.int32)
    nu = np.where(nk > 0, rng.integers(nu_range[0], nu_range[1] + 1, size=Nn), 0).astype(np.int32)
    A = np.concatenate([(1.2 / max(1, nx[dad[k]])) * rng.random(nx[k] * nx[dad[k]]) for k in range(1, Nn)])
    B = np.concatenate([rng.random(nx[k] * nu[dad[k]]) for k in range(1, Nn)])
    b = 0.1 * rng.random(int(nx[1:].sum()))
    sx, su = int(nx.sum()), int(nu.sum())
    xmin = -INF * np.ones(sx)
    xmax = INF * np.ones(sx)
    x0 = rng.random(nx[0])
    xmin[:nx[0]] = x0
    xmax[:nx[0]] = x0
    return FlatProblem(
        name=f"random_shape_qp(seed={seed},depth={depth},max_kids={max_kids},nx={nx_range},nu={nu_range})", nk=nk, nx=nx, nu=nu, A=A, B=B, b=b,
        Qd=1.0 + 9.0 * rng.random(sx), Rd=0.5 + rng.random(su), q=rng.random(sx) - 0.5, r=rng.random(su) - 0.5,
        xmin=xmin, xmax=xmax, umin=-ubound * np.ones(su), umax=ubound * np.ones(su))


def random_uniform_tree_qp(seed: int, nx: int, nu: int, md: int, Nr: int, Nh: int, ubound: float = 0.4, xbound: float = 3.0) -> FlatProblem:
    """Uniform / multistage tree shape (setup_multistage_tree(md, Nr, Nh)) with every edge and node carrying its OWN random
    data (time- and scenario-varying A, B, b, diagonal weights, linear terms, bounds): the shapes the persistent path takes,
    without the structure of the LTI models (same matrices per realization) the other fixtures have."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nk = multistage_nk(md, Nr, Nh)
    Nn = len(nk)
    Np = int((nk > 0).sum())
    nxv = np.full(Nn, nx, dtype=np.int32)
    nuv = np.where(nk > 0, nu, 0).astype(np.int32)
    A = (1.1 / nx) * rng.random((Nn - 1, nx * nx)) + np.tile((0.4 * np.eye(nx)).reshape(-1, order="F"), (Nn - 1, 1))
    B = rng.random((Nn - 1, nx * nu)) - 0.3
    b = 0.05 * (rng.random((Nn - 1, nx)) - 0.5)
    x0 = rng.random(nx) - 0.5
    xmin = -xbound * (0.5 + rng.random(Nn * nx))
    xmax = xbound * (0.5 + rng.random(Nn * nx))
    xmin[:nx] = x0
    xmax[:nx] = x0
    return FlatProblem(
        name=f"random_uniform_tree_qp(seed={seed},nx={nx},nu={nu},md={md},Nr={Nr},Nh={Nh})", nk=nk, nx=nxv, nu=nuv,
        A=A.ravel(), B=B.ravel(), b=b.ravel(), Qd=0.5 + 9.5 * rng.random(Nn * nx), Rd=0.2 + 2.0 * rng.random(Np * nu),
        q=rng.random(Nn * nx) - 0.5, r=rng.random(Np * nu) - 0.5, xmin=xmin, xmax=xmax,
        umin=-ubound * (0.5 + rng.random(Np * nu)), umax=ubound * (0.5 + rng.random(Np * nu)))


def thesis_example() -> FlatProblem:
    """The 6-node tree of examples/thesis_example.c:52-92 (values typed from the example's setters)."""
    nk = np.asarray([2, 2, 1, 0, 0, 0], dtype=np.int32)
    nx = np.full(6, 2, dtype=np.int32)
    nu = np.asarray([1, 1, 1, 0, 0, 0], dtype=np.int32)
    A1, A2 = [1.1, 3.3, 2.2, 4.4], [5.5, 7.7, 6.6, 8.8]
    B1, B2 = [1.0, 2.0], [3.0, 4.0]
    b1, b2 = [0.0, 0.0], [1.0, 1.0]
    edges = [(A1, B1, b1), (A2, B2, b2), (A1, B1, b1), (A2, B2, b2), (A2, B2, b2)]
    xmin = -INF * np.ones(12)
    xmax = INF * np.ones(12)
    xmin[:2] = 2.1
    xmax[:2] = 2.1
    return FlatProblem(
        name="thesis_example", nk=nk, nx=nx, nu=nu,
        A=np.concatenate([e[0] for e in edges]), B=np.concatenate([e[1] for e in edges]),
        b=np.concatenate([e[2] for e in edges]), Qd=2.0 * np.ones(12), Rd=np.ones(3),
        q=np.zeros(12), r=np.zeros(3), xmin=xmin, xmax=xmax, umin=-np.ones(3), umax=np.ones(3))


def random_qp_fixture(i: int) -> dict:
    """Reference unit-test fixture data0<i>.json -> flat dense arrays + golden xopt/uopt."""
    d = json.loads((GOLDEN / f"random_qp_data0{i}.json").read_text())
    nodes, edges = d["nodes"], d["edges"]
    Nn = len(nodes)
    col = lambda M: np.asarray(M, dtype=float).reshape(-1, order="F") if np.ndim(M) == 2 else np.atleast_1d(np.asarray(M, dtype=float)).ravel()
    nx = np.asarray([len(np.atleast_1d(n["q"])) for n in nodes], dtype=np.int32)
    nu = np.asarray([len(np.atleast_1d(n["r"])) if n["r"] is not None else 0 for n in nodes], dtype=np.int32)
    edges = sorted(edges, key=lambda e: e["to"])
    dad = np.full(Nn, -1)
    for e in edges:
        dad[e["to"]] = e["from"]
    nk = np.asarray([(dad == k).sum() for k in range(Nn)], dtype=np.int32)
    out = dict(nk=nk, nx=nx, nu=nu,
               A=np.concatenate([col(e["A"]) for e in edges]), B=np.concatenate([col(e["B"]) for e in edges]),
               b=np.concatenate([col(e["b"]) for e in edges]),
               Q=np.concatenate([col(n["Q"]) for n in nodes]),
               R=np.concatenate([col(n["R"]) for n in nodes if n["R"] is not None] or [np.zeros(0)]),
               S=np.concatenate([col(n["S"]) for n in nodes if n["S"] is not None] or [np.zeros(0)]),
               q=np.concatenate([col(n["q"]) for n in nodes]),
               r=np.concatenate([col(n["r"]) for n in nodes if n["r"] is not None] or [np.zeros(0)]),
               xopt=np.concatenate([col(n["xopt"]) for n in nodes]),
               uopt=np.concatenate([col(n["uopt"]) for n in nodes if n["uopt"] is not None] or [np.zeros(0)]))
    return out
