"""Pin the oracle on the reference's own fixtures / assertions for this path.

These are the anchors that make oracle == reference behaviour credible:
  * examples/spring_mass_dual_newton_tree.c:154-157   KKT < 1e-8 on the default example (C1)
  * examples/spring_mass.c:304-331 (tdunes branch)    x0 eliminated, xmax[1]=0.2, KKT < 1e-10
  * examples/thesis_example.c                         6-node clipping example
  * examples/random_qp.c:249-254 + data0[0-5].json    YALMIP/quadprog goldens, 1e-12, <= 1 iteration
Values marked "survey probe" were measured in SURVEY.md §8(c) by running the unmodified reference
treeqp sources; they are quoted from that document, not re-generated here.
"""
import numpy as np
import pytest

from helpers import oracle_flat_from_lti, product_qp_from_lti
from treeqp_amd import problems as P


def test_c1_default_example(orc):
    p = P.spring_mass()
    qp = oracle_flat_from_lti(orc, p)
    sol = orc.solve(qp, lambda0=p.lambda0)
    assert sol["status"] == 0
    assert orc.max_kkt(qp, sol) < 1e-8                       # the reference driver's own assert
    assert sol["iter"] == 3                                  # survey probe: 3 Newton iterations
    assert orc.max_kkt(qp, sol) < 1e-12                      # survey probe: 8.5e-14
    x1 = sol["x"][4:8]
    probe = np.array([0.018047417833111958, 0.023952582166885367, 0.1308800866412696, 0.039119913358725])
    assert np.max(np.abs(x1 - probe)) < 1e-12                # survey probe x[1]
    assert sol["u"][0] == 1.0                                # saturated input


def test_c1_depth4_plumbing(orc):
    p = P.spring_mass(Nh=4)
    assert p.Nn == 31
    qp = oracle_flat_from_lti(orc, p)
    sol = orc.solve(qp, lambda0=p.lambda0)
    assert sol["status"] == 0 and orc.max_kkt(qp, sol) < 1e-8


def test_spring_mass_x0_eliminated(orc, capi):
    p = P.spring_mass(xmax1=0.2)
    qp = product_qp_from_lti(capi, p, eliminate_x0=True)     # host container does the elimination
    fl = qp.flat()
    sol = orc.solve(fl, lambda0=p.lambda0)
    assert sol["status"] == 0
    assert orc.max_kkt(fl, sol) < 1e-10                      # examples/spring_mass.c:331
    assert sol["iter"] == 58                                 # survey probe: 58 Newton iterations


def test_thesis_example(orc):
    t = P.thesis_example()
    sol = orc.solve(t.as_dict())
    assert sol["status"] == 0 and sol["iter"] == 3           # survey probe: status 0, 3 iterations
    assert orc.max_kkt(t.as_dict(), sol) < 1e-10
    assert np.allclose(sol["x"][10:12], [335.645, 455.777], rtol=0, atol=1e-9)   # survey probe x[5]


@pytest.mark.parametrize("i", range(6))
def test_random_qp_goldens_dense(orc, i):
    f = P.random_qp_fixture(i)
    o = orc.default_opts(maxIter=10, stationarityTolerance=1e-10, regType=0)     # random_qp.c:131-133
    sol = orc.solve_dense(f, o)
    assert sol["status"] == 0
    assert sol["iter"] in (0, 1)                                                  # :251-253
    assert np.max(np.abs(sol["x"] - f["xopt"])) < 1e-12                           # :250
    assert np.max(np.abs(sol["u"] - f["uopt"])) < 1e-12
    assert orc.max_kkt(f, sol, dense=True) < 1e-12                                # :249


def test_c2_linear_chain(orc):
    p = P.linear_chain()
    assert p.Nn == 1023
    qp = oracle_flat_from_lti(orc, p)
    sol = orc.solve(qp, lambda0=p.lambda0)
    assert sol["status"] == 0 and sol["iter"] == 3 and sol["ls_total"] == 3      # survey probe
    assert orc.max_kkt(qp, sol) < 1e-10
    assert sol["n_active"] - 8 == 243                                             # survey probe (x0 pins 8 more)
    assert np.allclose(sol["u"][:3], [-0.018065, -0.271016, -0.5], rtol=0, atol=1e-6)


def test_active_set_bookkeeping_is_result_neutral(orc):
    """checkLastActiveSet only skips recomputation (dual_Newton_tree.c:556-614,703-709): the
    device path always rebuilds, which must be bit-identical."""
    for p in (P.spring_mass(), P.spring_mass(xmax1=0.2), P.linear_chain(2, 5, 5)):
        qp = oracle_flat_from_lti(orc, p)
        a = orc.solve(qp, orc.default_opts(checkLastActiveSet=1), p.lambda0)
        b = orc.solve(qp, orc.default_opts(checkLastActiveSet=0), p.lambda0)
        assert a["iter"] == b["iter"] and a["ls_total"] == b["ls_total"]
        for k in ("x", "u", "lam", "mu_x", "mu_u"):
            assert np.array_equal(a[k], b[k]), k


def test_options_and_failure_modes(orc):
    p = P.spring_mass(Nh=4)
    qp = oracle_flat_from_lti(orc, p)
    assert orc.solve(qp, orc.default_opts(termCondition=7))["status"] == 9       # INVALID_OPTION
    assert orc.solve(qp, orc.default_opts(regValue=-1.0))["status"] == 9
    s = orc.solve(qp, orc.default_opts(maxIter=1), p.lambda0)
    assert s["status"] == 1 and s["iter"] == 1                                    # MAXIMUM_ITERATIONS
    for term in (0, 1, 2):
        s = orc.solve(qp, orc.default_opts(termCondition=term), p.lambda0)
        assert s["status"] == 0
    for reg in (0, 1, 2):
        s = orc.solve(qp, orc.default_opts(regType=reg), p.lambda0)
        assert s["status"] == 0 and orc.max_kkt(qp, s) < 1e-8


def test_openmp_threads_do_not_change_results(orc):
    p = P.linear_chain(2, 6, 6)
    qp = oracle_flat_from_lti(orc, p)
    a = orc.solve(qp, orc.default_opts(num_threads=1), p.lambda0)
    b = orc.solve(qp, orc.default_opts(num_threads=4), p.lambda0)
    for k in ("x", "u", "lam"):
        assert np.array_equal(a[k], b[k])


def test_random_fixture_families_are_well_posed(orc):
    """The seeded random families the GPU parity tests draw from (random tree shapes with per-node dimensions, uniform /
    multistage shapes with time-varying data, pruned scenario trees): the oracle converges on every member to a KKT point,
    most of them with Armijo backtracking -- so a GPU mismatch on one of them is a device bug, not a fixture problem."""
    from treeqp_amd import problems as P
    import test_gpu_parity as G
    backtracking = 0
    for c in G.RANDOM_SHAPES:
        f = P.random_shape_qp(*c)
        ref = orc.solve(f.as_dict())
        assert ref["status"] == 0 and orc.max_kkt(f.as_dict(), ref) < 1e-9, f.name
        backtracking += ref["ls_total"] > ref["iter"]
    for c in G.RANDOM_UNIFORM:
        f = P.random_uniform_tree_qp(*c)
        ref = orc.solve(f.as_dict())
        assert ref["status"] == 0 and orc.max_kkt(f.as_dict(), ref) < 1e-9, f.name
        backtracking += ref["ls_total"] > ref["iter"]
    for seed in range(7, 13):
        f = P.pruned_chain_qp(seed=seed)
        ref = orc.solve(f.as_dict(), orc.default_opts(**f.opts), lambda0=f.lambda0)
        assert ref["status"] == 0 and orc.max_kkt(f.as_dict(), ref) < 1e-8, f.name
    assert backtracking >= 15


def test_one_ulp_perturbations_tell_rounding_decisions_from_sound_ones(orc):
    """The classifier of the parity campaigns (helpers.ulp_sensitivity / ulp_solution_spread) on the oracle alone: the two campaign
    cases whose line-search decisions flip in iterations 9 and 12 (tools/fuzz_replay.py 200000 272081 279701) do so under one-ulp
    perturbations of the data in that iteration and in none before; a well-conditioned problem keeps every decision and moves its
    solution by rounding only."""
    from helpers import fuzz_case, ulp_sensitivity, ulp_solution_spread
    for seed, first_diff in ((272081, 9), (279701, 12)):
        f, opts = fuzz_case(seed, 200000)
        assert ulp_sensitivity(orc, f, opts, first_diff - 1) == 0
        assert ulp_sensitivity(orc, f, opts, first_diff) >= 3
    f = P.random_shape_qp(5)
    assert ulp_sensitivity(orc, f, {}, 100) == 0
    assert ulp_solution_spread(orc, f, {}) < 1e-12
