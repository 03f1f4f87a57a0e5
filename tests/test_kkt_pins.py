"""Oracle-independent pin of the CLIPPING branch: tests/golden/kkt_pin_*.npz hold the unique KKT point of BASELINE C1, C2,
the thesis example and an irregular tree, computed by tools/make_kkt_pins.py with numpy / scipy only (sparse active-set
iteration + full KKT verification; no oracle, no product, no reference code involved).  The oracle (here, CPU) and the
device path (-m gpu) must both reproduce them -- to 1e-10 relative, the tolerance north_star states.  Without these the
oracle's phases S and L were held only by the drivers' own KKT asserts and the oracle and the device could drift together."""
from pathlib import Path

import numpy as np
import pytest

from helpers import assert_solution_close, oracle_flat_from_lti, product_qp_from_flat
from treeqp_amd import problems as P

GOLD = Path(__file__).resolve().parent / "golden"
CASES = {
    "c1_spring_mass": lambda: P.spring_mass(),
    "c1_depth4": lambda: P.spring_mass(Nh=4),
    "c2_linear_chain": lambda: P.linear_chain(2, 9, 9),
    "thesis_example": lambda: P.thesis_example(),
    "irregular": lambda: P.irregular_clipping_qp(),
}
FLAT_KEYS = ("nk", "nx", "nu", "A", "B", "b", "Qd", "Rd", "q", "r", "xmin", "xmax", "umin", "umax")
TOL = 1e-10


def load_pin(name):
    d = np.load(GOLD / f"kkt_pin_{name}.npz")
    flat = {k: d[k] for k in FLAT_KEYS}
    sol = {k[4:]: d[k] for k in d.files if k.startswith("sol_") and k != "sol_n_active"}
    return flat, sol, int(d["sol_n_active"])


def problem_and_flat(orc, name):
    p = CASES[name]()
    if isinstance(p, P.LtiProblem):
        return p, oracle_flat_from_lti(orc, p), p.lambda0
    return p, p.as_dict(), p.lambda0


@pytest.mark.parametrize("name", list(CASES))
def test_pin_data_is_the_problem_the_tests_solve(orc, name):
    """the numpy restatement of the LTI fill (incl. the integer-division stage scaling) and the oracle's agree bit for bit"""
    flat, _, _ = load_pin(name)
    _, mine, _ = problem_and_flat(orc, name)
    for k in FLAT_KEYS:
        assert np.array_equal(np.asarray(mine[k], dtype=flat[k].dtype), flat[k]), k


@pytest.mark.parametrize("name", list(CASES))
def test_oracle_reproduces_the_kkt_pin(orc, name):
    flat, pin, n_active = load_pin(name)
    p, mine, lam0 = problem_and_flat(orc, name)
    sol = orc.solve(mine, lambda0=lam0)
    assert sol["status"] == 0
    assert_solution_close(sol, pin, TOL)
    assert orc.max_kkt(mine, pin) < 1e-10            # and the pin satisfies the reference's own KKT measure
    hit = int(np.sum((sol["x"] >= flat["xmax"]) | (sol["x"] <= flat["xmin"])) + np.sum((sol["u"] >= flat["umax"]) | (sol["u"] <= flat["umin"])))
    pinned = int(np.sum(flat["xmin"] == flat["xmax"]))
    assert hit - pinned == n_active                   # same active set


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_device_reproduces_the_kkt_pin(orc, capi, name):
    flat, pin, _ = load_pin(name)
    p, mine, lam0 = problem_and_flat(orc, name)
    g = capi.TqGpu(flat["nk"], flat["nx"], flat["nu"]).upload(flat, lam0)
    r = g.solve()
    assert r["status"] == 0
    assert_solution_close(g.solution(), pin, TOL)
    g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c1_spring_mass", "thesis_example"])
def test_dropin_api_reproduces_the_kkt_pin(orc, capi, name):
    """through the reference-compatible host API (tree_qp_in / treeqp_tdunes_solve / tree_qp_out)"""
    flat, pin, _ = load_pin(name)
    p, mine, lam0 = problem_and_flat(orc, name)
    qp = product_qp_from_flat(capi, P.FlatProblem(name=name, **{k: flat[k] for k in FLAT_KEYS}))
    s = capi.TdunesSolver(qp)
    if lam0 is not None:
        s.set_dual_initialization(lam0)
    assert s.solve() == 0
    assert_solution_close(qp.solution(), pin, TOL)
    s.destroy()
