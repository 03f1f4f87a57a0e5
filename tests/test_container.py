"""Host QP container: setters/getters, LTI filler, x0 elimination, KKT check (no GPU needed).
Reference behaviour: treeqp/src/tree_qp_common.c (lines cited in qp_container.c)."""
import ctypes as C

import numpy as np
import pytest

from helpers import lti_dims, oracle_flat_from_lti, product_qp_from_flat, product_qp_from_lti
from treeqp_amd import problems as P


def test_struct_sizes_match(capi):
    capi.lib()      # raises if any ctypes struct disagrees with the compiled library


@pytest.mark.parametrize("prob", [P.spring_mass(), P.spring_mass(Nh=4), P.linear_chain(2, 3, 5)], ids=lambda p: p.name)
def test_lti_fill_matches_oracle_bitwise(capi, orc, prob):
    qp = product_qp_from_lti(capi, prob)
    got, ref = qp.flat(), oracle_flat_from_lti(orc, prob)
    for k in ("A", "B", "b", "Qd", "Rd", "q", "r", "xmin", "xmax", "umin", "umax", "nx", "nu", "nk"):
        assert np.array_equal(got[k], ref[k]), k


def test_lti_stage_scaling_uses_integer_division(capi):
    # md=3,Nr=2: 9 leaves; stage 1 has 3 nodes -> 9/3 = 3, stage 0 -> 9/1 = 9 (tree_qp_common.c:1911)
    p = P.spring_mass()
    f = product_qp_from_lti(capi, p).flat()
    assert np.allclose(f["Qd"][:4], 9 * p.Qd) and np.allclose(f["Qd"][4:8], 3 * p.Qd)
    assert np.allclose(f["Qd"][-4:], p.Pd)
    # 2 leaves per... md=2,Nr=1,Nh=3: stages have 1,2,2,2 nodes and 2 leaves: factor 2/1=2 then 1
    p2 = P.spring_mass(Nh=3, Nr=1, md=2)
    f2 = product_qp_from_lti(capi, p2).flat()
    assert np.allclose(f2["Qd"][:4], 2 * p2.Qd) and np.allclose(f2["Qd"][4:8], p2.Qd)


def test_flat_roundtrip_irregular(capi):
    f = P.irregular_clipping_qp()
    got = product_qp_from_flat(capi, f).flat()
    for k in ("A", "B", "b", "Qd", "Rd", "q", "r", "xmin", "xmax", "umin", "umax"):
        assert np.array_equal(got[k], getattr(f, k)), k


def test_edge_and_node_accessors(capi):
    L = capi.lib()
    f = P.thesis_example()
    qp = capi.TreeQp(f.nx, f.nu, f.nk)
    A = np.arange(4.0) + 1
    B = np.array([5.0, 6.0])
    b = np.array([7.0, 8.0])
    qp.set_edge_dynamics(3, A, B, b)
    oA, oB, ob = np.zeros(4), np.zeros(2), np.zeros(2)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    L.tree_qp_in_get_edge_dynamics_colmajor(dp(oA), dp(oB), dp(ob), C.byref(qp.qp_in), 3)
    assert np.array_equal(oA, A) and np.array_equal(oB, B) and np.array_equal(ob, b)
    # padded lda
    Apad = np.zeros(6)
    L.tree_qp_in_get_edge_A_colmajor(dp(Apad), 3, C.byref(qp.qp_in), 3)
    assert Apad.tolist() == [1, 2, 0, 3, 4, 0]
    # default bounds are +-TREEQP_INF (tree_qp_in_create calls set_inf_bounds)
    fl = qp.flat()
    assert np.all(fl["xmin"] == -1e12) and np.all(fl["umax"] == 1e12)
    # dense objective setter + getter
    Q = np.array([[2.0, 0.5], [0.5, 3.0]])
    qp.set_node_objective(1, Q, np.array([[1.5]]), np.array([[0.1, 0.2]]), np.array([1.0, 2.0]), np.array([3.0]))
    oQ = np.zeros(4)
    L.tree_qp_in_get_node_Q_colmajor(dp(oQ), -1, C.byref(qp.qp_in), 1)
    assert np.array_equal(oQ, Q.reshape(-1, order="F"))


def test_sizes(capi):
    L = capi.lib()
    f = P.irregular_clipping_qp()
    qp = product_qp_from_flat(capi, f)
    q = C.byref(qp.qp_in)
    assert L.total_number_of_states(q) == int(f.nx.sum())
    assert L.total_number_of_controls(q) == int(f.nu.sum())
    assert L.max_number_of_states(q) == int(f.nx.max())
    assert L.total_number_of_dynamic_constraints(q) == int(f.nx[1:].sum())
    assert L.total_number_of_primal_variables(q) == int(f.nx.sum() + f.nu.sum())


def test_host_kkt_matches_oracle(capi, orc):
    for prob in (P.thesis_example(), P.irregular_clipping_qp()):
        qp = product_qp_from_flat(capi, prob)
        sol = orc.solve(prob.as_dict())
        assert sol["status"] == 0
        qp.set_solution(sol)
        k_host, k_orc = qp.max_kkt_res(), orc.max_kkt(prob.as_dict(), sol)
        assert k_host < 1e-9 and abs(k_host - k_orc) <= 1e-11   # same residual, different summation order
        # a perturbed point must be flagged by both
        bad = {k: v.copy() for k, v in sol.items() if isinstance(v, np.ndarray)}
        bad["x"][-1] += 1e-3
        qp.set_solution(bad)
        assert qp.max_kkt_res() > 1e-4 and orc.max_kkt(prob.as_dict(), bad) > 1e-4


def test_eliminate_x0_and_update(capi, orc):
    """examples/spring_mass.c:237-239 + tree_qp_in_set_x0_colmaj (tree_qp_common.c:2154-2235)."""
    p = P.spring_mass(xmax1=0.2)
    qp = product_qp_from_lti(capi, p)
    before = qp.flat()
    qp.eliminate_x0()
    after = qp.flat()
    nx = p.nx
    assert after["nx"][0] == 0 and np.all(after["nx"][1:] == nx)
    nk0 = int(p.nk()[0])
    # b of the root's children absorbed A*x0; the remaining data is untouched
    for e in range(nk0):
        A0 = before["A"][e * nx * nx:(e + 1) * nx * nx].reshape(nx, nx, order="F")
        expect = before["b"][e * nx:(e + 1) * nx] + A0 @ p.x0
        assert np.allclose(after["b"][e * nx:(e + 1) * nx], expect, rtol=0, atol=1e-15)
    assert np.array_equal(after["A"], before["A"][nk0 * nx * nx:])
    assert np.array_equal(after["B"], before["B"])
    assert np.array_equal(after["Qd"], before["Qd"][nx:]) and np.array_equal(after["xmin"], before["xmin"][nx:])
    # new x0 after elimination
    x0b = 2.0 * p.x0
    qp.set_x0(x0b)
    again = qp.flat()
    for e in range(nk0):
        A0 = before["A"][e * nx * nx:(e + 1) * nx * nx].reshape(nx, nx, order="F")
        assert np.allclose(again["b"][e * nx:(e + 1) * nx], before["b"][e * nx:(e + 1) * nx] + A0 @ x0b, rtol=0, atol=1e-15)
    # the oracle solves the eliminated problem and the host KKT check accepts its solution
    qp.set_x0(p.x0)
    fl = qp.flat()
    sol = orc.solve(fl, lambda0=p.lambda0)
    assert sol["status"] == 0
    qp.set_solution(sol)
    assert qp.max_kkt_res() < 1e-10


# ---- JSON front end (treeqp_solve_json): parsing needs no device --------------------------------

def _json_tool():
    from pathlib import Path
    exe = Path(__file__).resolve().parent.parent / "treeqp_amd" / "lib" / "treeqp_solve_json"
    if not exe.exists():
        pytest.skip("treeqp_solve_json was not built")
    return exe


@pytest.mark.parametrize("i", range(6))
def test_json_front_end_reads_reference_fixture_dims(i):
    """examples/random_qp_utils/data0<i>.json: dimensions and tree shape as the Python loader sees them."""
    import json, subprocess
    from pathlib import Path
    from treeqp_amd import problems as P
    f = P.random_qp_fixture(i)
    path = Path(__file__).resolve().parent / "golden" / f"random_qp_data0{i}.json"
    out = subprocess.run([str(_json_tool()), "--dims", str(path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout)
    assert d["Nn"] == len(f["nk"]) and d["nx"] == list(map(int, f["nx"])) and d["nu"] == list(map(int, f["nu"]))
    assert d["nk"] == list(map(int, f["nk"])) and d["has_options"] is False


def test_json_front_end_rejects_malformed_input(tmp_path):
    import subprocess
    bad = tmp_path / "bad.json"
    bad.write_text('{"nodes": [{"q": [1, 2], "r": []}], "edges": [')
    out = subprocess.run([str(_json_tool()), "--dims", str(bad)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "solve_qp_json" in out.stderr


def test_regularised_cholesky_utilities(capi):
    """treeqp_dpotrf_l(_mn)_with_reg_opts (dual_Newton_common.c:36-123): NO / ALWAYS / ON_THE_FLY, zero column for a
    non-positive pivot, M modified when regularised."""
    import ctypes as C
    L = capi.lib()
    L.treeqp_dpotrf_l_with_reg_opts.argtypes = [C.POINTER(capi.Dmat), C.POINTER(capi.Dmat), C.c_int, C.c_double, C.c_double]
    L.treeqp_dpotrf_l_mn_with_reg_opts.argtypes = L.treeqp_dpotrf_l_with_reg_opts.argtypes
    rng = np.random.Generator(np.random.PCG64(5))

    def dmat(a):
        a = np.asfortranarray(a, dtype=np.float64)
        m = capi.Dmat()
        L.blasfeo_allocate_dmat(a.shape[0], a.shape[1], C.byref(m))
        L.blasfeo_pack_dmat(a.shape[0], a.shape[1], a.ctypes.data_as(C.POINTER(C.c_double)), a.shape[0], C.byref(m), 0, 0)
        return m

    def arr(m, r, c):
        return np.ctypeslib.as_array(m.pA, shape=(c, r)).T.copy()

    B = rng.standard_normal((6, 6))
    W = B @ B.T + 0.5 * np.eye(6)
    for reg, val in ((0, 0.0), (1, 1e-3), (2, 1e-3)):
        M, Lc = dmat(W), dmat(np.zeros((6, 6)))
        res = L.treeqp_dpotrf_l_with_reg_opts(C.byref(M), C.byref(Lc), reg, 1e-6, val)
        Wreg = W + (val if reg == 1 else 0.0) * np.eye(6)
        assert res == (1 if reg == 1 else 0)
        assert np.allclose(np.tril(arr(Lc, 6, 6)), np.linalg.cholesky(Wreg), rtol=0, atol=1e-12)
        assert np.allclose(arr(M, 6, 6), Wreg, rtol=0, atol=0)
    # singular block: ON_THE_FLY notices the zero diagonal of the first factor, shifts and refactorises
    v = rng.standard_normal((6, 2))
    Ws = v @ v.T
    M, Lc = dmat(Ws), dmat(np.zeros((6, 6)))
    assert L.treeqp_dpotrf_l_with_reg_opts(C.byref(M), C.byref(Lc), 2, 1e-6, 1e-4) == 1
    assert np.allclose(np.tril(arr(Lc, 6, 6)), np.linalg.cholesky(Ws + 1e-4 * np.eye(6)), rtol=0, atol=1e-9)
    # without regularisation the non-positive pivot gives a ZERO column, not NaN
    M, Lc = dmat(np.diag([4.0, -1.0, 9.0])), dmat(np.zeros((3, 3)))
    assert L.treeqp_dpotrf_l_with_reg_opts(C.byref(M), C.byref(Lc), 0, 1e-6, 0.0) == 0
    assert np.array_equal(np.diag(arr(Lc, 3, 3)), [2.0, 0.0, 3.0])
    # tall variant: the rows below the square part are divided through (trsm fused into the factorisation)
    T = np.vstack([W, rng.standard_normal((3, 6))])
    M, Lc = dmat(T), dmat(np.zeros((9, 6)))
    assert L.treeqp_dpotrf_l_mn_with_reg_opts(C.byref(M), C.byref(Lc), 0, 1e-6, 0.0) == 0
    Lw = np.linalg.cholesky(W)
    got = arr(Lc, 9, 6)
    assert np.allclose(np.tril(got[:6]), Lw, rtol=0, atol=1e-12) and np.allclose(got[6:], T[6:] @ np.linalg.inv(Lw).T, rtol=0, atol=1e-11)
