"""Python mirror of the treeQP C API served by libtreeqp_amd.so (ctypes, no torch types).

Two levels, both calling straight into the C-ABI:

* ``TreeQp`` / ``TdunesSolver`` -- mirror the reference's own object wrapper
  (interfaces/treeqp_cpp/treeqp_cpp_interface.hpp:36-175: TreeQp, TdunesSolver with string-keyed
  SetOption) on top of the reference-compatible C functions (tree_qp_in_*, treeqp_tdunes_*).
* ``TqGpu`` -- the thin device C-ABI of include/treeqp_amd.h (tqgpu_*), flat arrays in/out.

There is no CPU solve path behind either: constructing a solver without a usable HIP device
raises ``RuntimeError`` (the C function would print the HIP error and ``exit(1)``).
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_ROOT = Path(__file__).resolve().parent
_LIB = None

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)

RETURN_T = {0: "TREEQP_OPTIMAL_SOLUTION_FOUND", 1: "TREEQP_MAXIMUM_ITERATIONS_REACHED",
            2: "TREEQP_DN_NOT_DESCENT_DIRECTION", 7: "TREEQP_OK", 9: "TREEQP_INVALID_OPTION"}
TERMINATION = {"TREEQP_SUMSQUAREDERRORS": 0, "TREEQP_TWONORM": 1, "TREEQP_INFNORM": 2}
REGTYPE = {"TREEQP_NO_REGULARIZATION": 0, "TREEQP_ALWAYS_LEVENBERG_MARQUARDT": 1,
           "TREEQP_ON_THE_FLY_LEVENBERG_MARQUARDT": 2}


class Dmat(C.Structure):
    _fields_ = [("pA", c_dbl_p), ("m", C.c_int), ("n", C.c_int), ("memsize", C.c_int)]


class Dvec(C.Structure):
    _fields_ = [("pa", c_dbl_p), ("m", C.c_int), ("memsize", C.c_int)]


class Node(C.Structure):
    _fields_ = [("kids", c_int_p), ("idx", C.c_int), ("dad", C.c_int), ("nkids", C.c_int),
                ("stage", C.c_int), ("real", C.c_int), ("idxkid", C.c_int)]


class Info(C.Structure):
    _fields_ = [("Nn", C.c_int), ("iter", C.c_int), ("total_time", C.c_double),
                ("solver_time", C.c_double), ("interface_time", C.c_double)]


class QpInternal(C.Structure):
    _fields_ = [("is_A_initialized", c_int_p), ("is_b_initialized", c_int_p), ("is_C_initialized", C.c_int),
                ("is_dmin_initialized", C.c_int), ("is_dmax_initialized", C.c_int), ("is_S_initialized", C.c_int),
                ("is_r_initialized", C.c_int), ("x0", Dvec), ("A0", C.POINTER(Dmat)), ("b0", C.POINTER(Dvec)),
                ("C0", Dmat), ("dmax0", Dvec), ("dmin0", Dvec), ("S0", Dmat), ("r0", Dvec)]


class QpIn(C.Structure):
    _fields_ = [("N", C.c_int), ("nx", c_int_p), ("nu", c_int_p), ("nc", c_int_p),
                ("A", C.POINTER(Dmat)), ("B", C.POINTER(Dmat)), ("b", C.POINTER(Dvec)),
                ("Q", C.POINTER(Dmat)), ("R", C.POINTER(Dmat)), ("S", C.POINTER(Dmat)),
                ("q", C.POINTER(Dvec)), ("r", C.POINTER(Dvec)),
                ("xmin", C.POINTER(Dvec)), ("xmax", C.POINTER(Dvec)), ("umin", C.POINTER(Dvec)), ("umax", C.POINTER(Dvec)),
                ("C", C.POINTER(Dmat)), ("D", C.POINTER(Dmat)), ("dmin", C.POINTER(Dvec)), ("dmax", C.POINTER(Dvec)),
                ("tree", C.POINTER(Node)), ("internal_memory", QpInternal)]


class QpOut(C.Structure):
    _fields_ = [("info", Info), ("x", C.POINTER(Dvec)), ("u", C.POINTER(Dvec)), ("lam", C.POINTER(Dvec)),
                ("mu_x", C.POINTER(Dvec)), ("mu_u", C.POINTER(Dvec)), ("mu_d", C.POINTER(Dvec))]


class TdunesOpts(C.Structure):
    _fields_ = [("maxIter", C.c_int), ("qp_solver", c_int_p), ("checkLastActiveSet", C.c_int),
                ("stationarityTolerance", C.c_double), ("termCondition", C.c_int), ("regType", C.c_int),
                ("regTol", C.c_double), ("regValue", C.c_double), ("lineSearchMaxIter", C.c_int),
                ("lineSearchGamma", C.c_double), ("lineSearchBeta", C.c_double), ("lineSearchRestartTrigger", C.c_int)]


class Profiling(C.Structure):
    _fields_ = [("num_iter", C.c_int), ("run_indx", C.c_int), ("total_time", C.c_double), ("min_total_time", C.c_double),
                ("total_ls_iter", C.c_int), ("iter_times", c_dbl_p), ("min_iter_times", c_dbl_p), ("ls_iters", c_int_p),
                ("stage_qps_times", c_dbl_p), ("min_stage_qps_times", c_dbl_p), ("build_dual_times", c_dbl_p),
                ("min_build_dual_times", c_dbl_p), ("newton_direction_times", c_dbl_p),
                ("min_newton_direction_times", c_dbl_p), ("line_search_times", c_dbl_p), ("min_line_search_times", c_dbl_p)]


class TdunesWork(C.Structure):
    _fields_ = [("Nn", C.c_int), ("Np", C.c_int), ("lsIter", C.c_int), ("lineSearchRestartCounter", C.c_int),
                ("npar", c_int_p), ("idxpos", c_int_p), ("sx", C.POINTER(Dvec)), ("su", C.POINTER(Dvec)),
                ("slambda", C.POINTER(Dvec)), ("sDeltalambda", C.POINTER(Dvec)), ("timings", Profiling),
                ("device", C.c_void_p), ("stage", c_dbl_p), ("stage_doubles", C.c_int), ("lsTotal", C.c_int),
                ("maxIterAtCreate", C.c_int), ("denseStageSolver", C.c_int)]


class GpuOpts(C.Structure):
    _fields_ = [("maxIter", C.c_int), ("termCondition", C.c_int), ("stationarityTolerance", C.c_double),
                ("regType", C.c_int), ("regTol", C.c_double), ("regValue", C.c_double),
                ("lineSearchMaxIter", C.c_int), ("lineSearchGamma", C.c_double), ("lineSearchBeta", C.c_double),
                ("lineSearchRestartTrigger", C.c_int), ("profile", C.c_int), ("checkLastActiveSet", C.c_int)]


class GpuResult(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("ls_total", C.c_int), ("ls_last", C.c_int),
                ("n_launches", C.c_int), ("device_time", C.c_double), ("last_error_norm", C.c_double),
                ("last_fval", C.c_double)]


_SIZEOF_CHECK = [(0, Dmat), (1, Dvec), (2, Node), (3, QpIn), (4, QpOut), (5, TdunesOpts), (6, TdunesWork),
                 (7, Profiling), (8, QpInternal)]


def library_path() -> Path:
    # TREEQP_AMD_LIB: an experiment build of the same library (treeqp_amd/build.py --variant), for A/B timing runs
    override = os.environ.get("TREEQP_AMD_LIB")
    return Path(override) if override else _ROOT / "lib" / "libtreeqp_amd.so"


def lib():
    """Load libtreeqp_amd.so (built in-tree by treeqp_amd/build.py); fail loudly if it is missing."""
    global _LIB
    if _LIB is None:
        so = library_path()
        if not so.exists():
            raise RuntimeError(f"{so} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                               "(the treeqp_amd hot path has no pure-Python or CPU fallback)")
        L = C.CDLL(str(so))
        L.tqgpu_last_error.restype = C.c_char_p
        L.tqgpu_version.restype = C.c_char_p
        L.tree_qp_out_max_KKT_res.restype = C.c_double
        for which, typ in _SIZEOF_CHECK:
            got = L.treeqp_amd_sizeof(which)
            if got != C.sizeof(typ):
                raise RuntimeError(f"ctypes declaration of {typ.__name__} ({C.sizeof(typ)} B) does not match the library ({got} B)")
        _LIB = L
    return _LIB


def device_count() -> int:
    return int(lib().tqgpu_device_count())


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dbl_p)


def _aligned_buffer(nbytes: int):
    raw = np.zeros(nbytes + 64, dtype=np.uint8)
    off = (-raw.ctypes.data) % 64
    return raw, C.c_void_p(raw.ctypes.data + off)


# ------------------------------------------------------------------------------------------------
# TreeQp: tree_qp_in + tree_qp_out (reference: TreeQp in treeqp_cpp_interface.hpp:36-99)
# ------------------------------------------------------------------------------------------------

class TreeQp:
    def __init__(self, nx, nu, nk, nc=None):
        L = lib()
        self.nk = _i32(nk)
        self.nx0 = _i32(nx).copy()      # dimensions at creation (before any x0 elimination)
        self.nu0 = _i32(nu).copy()
        self.N = len(self.nk)
        if not (len(self.nx0) == len(self.nu0) == self.N):
            raise ValueError("nx, nu, nk must have one entry per node")
        if L.number_of_nodes_from_nkids(_ip(self.nk)) != self.N:
            raise ValueError("nk does not describe a tree with len(nk) nodes and uniform leaf depth")
        ncp = None if nc is None else _ip(_i32(nc))
        self.qp_in = QpIn()
        self.qp_out = QpOut()
        size_in = L.tree_qp_in_calculate_size(self.N, _ip(self.nx0), _ip(self.nu0), ncp, _ip(self.nk))
        self._mem_in, p_in = _aligned_buffer(size_in)
        L.tree_qp_in_create(self.N, _ip(self.nx0), _ip(self.nu0), ncp, _ip(self.nk), C.byref(self.qp_in), p_in)
        size_out = L.tree_qp_out_calculate_size(self.N, _ip(self.nx0), _ip(self.nu0), ncp)
        self._mem_out, p_out = _aligned_buffer(size_out)
        L.tree_qp_out_create(self.N, _ip(self.nx0), _ip(self.nu0), ncp, C.byref(self.qp_out), p_out)

    # ---- dimensions as the library currently sees them ----
    @property
    def nx(self):
        return np.ctypeslib.as_array(self.qp_in.nx, shape=(self.N,)).copy()

    @property
    def nu(self):
        return np.ctypeslib.as_array(self.qp_in.nu, shape=(self.N,)).copy()

    def tree(self):
        t = self.qp_in.tree
        out = {k: np.asarray([getattr(t[i], k) for i in range(self.N)], dtype=np.int32)
               for k in ("idx", "dad", "nkids", "stage", "real", "idxkid")}
        out["kids"] = [[t[i].kids[c] for c in range(t[i].nkids)] for i in range(self.N)]
        return out

    # ---- setters (names follow the C API / the C++ wrapper) ----
    def set_edge_dynamics(self, e, A, B, b):
        lib().tree_qp_in_set_edge_dynamics_colmajor(_dp(_f64(np.asarray(A).reshape(-1, order="F") if np.ndim(A) == 2 else A)),
                                                    _dp(_f64(np.asarray(B).reshape(-1, order="F") if np.ndim(B) == 2 else B)),
                                                    _dp(_f64(b)), C.byref(self.qp_in), int(e))

    def set_node_objective_diag(self, k, Qd, Rd, q, r):
        lib().tree_qp_in_set_node_objective_diag(_dp(_f64(Qd)), _dp(_f64(Rd)), _dp(_f64(q)), _dp(_f64(r)), C.byref(self.qp_in), int(k))

    def set_node_objective(self, k, Q, R, S, q, r):
        f = lambda M: _dp(_f64(np.asarray(M).reshape(-1, order="F") if np.ndim(M) == 2 else M))
        lib().tree_qp_in_set_node_objective_colmajor(f(Q), f(R), f(S), _dp(_f64(q)), _dp(_f64(r)), C.byref(self.qp_in), int(k))

    def set_node_bounds(self, k, xmin, xmax, umin, umax):
        lib().tree_qp_in_set_node_bounds(_dp(_f64(xmin)), _dp(_f64(xmax)), _dp(_f64(umin)), _dp(_f64(umax)), C.byref(self.qp_in), int(k))

    def set_ltv_dynamics(self, A, B, b):
        lib().tree_qp_in_set_ltv_dynamics_colmajor(_dp(_f64(A)), _dp(_f64(B)), _dp(_f64(b)), C.byref(self.qp_in))

    def set_ltv_objective_diag(self, Qd, Rd, q, r):
        lib().tree_qp_in_set_ltv_objective_diag(_dp(_f64(Qd)), _dp(_f64(Rd)), _dp(_f64(q)), _dp(_f64(r)), C.byref(self.qp_in))

    def set_ltv_bounds(self, xmin, xmax, umin, umax):
        lib().tree_qp_in_set_ltv_bounds(_dp(_f64(xmin)), _dp(_f64(xmax)), _dp(_f64(umin)), _dp(_f64(umax)), C.byref(self.qp_in))

    def set_inf_bounds(self):
        lib().tree_qp_in_set_inf_bounds(C.byref(self.qp_in))

    def set_flat(self, p):
        """Load a problems.FlatProblem (or mapping with the same keys)."""
        g = (lambda k: getattr(p, k)) if not isinstance(p, dict) else (lambda k: p[k])
        self.set_ltv_dynamics(g("A"), g("B"), g("b"))
        self.set_ltv_objective_diag(g("Qd"), g("Rd"), g("q"), g("r"))
        self.set_ltv_bounds(g("xmin"), g("xmax"), g("umin"), g("umax"))
        return self

    def fill_lti(self, p):
        """tree_qp_in_fill_lti_data_diag_weights with a problems.LtiProblem."""
        a = [_f64(v) for v in (p.A, p.B, p.b, p.Qd, p.q, p.Pd, p.p, p.Rd, p.r, p.xmin, p.xmax, p.umin, p.umax, p.x0)]
        lib().tree_qp_in_fill_lti_data_diag_weights(*[_dp(v) for v in a], None, None, None, None, None, C.byref(self.qp_in))
        return self

    def eliminate_x0(self):
        lib().tree_qp_in_eliminate_x0(C.byref(self.qp_in))
        lib().tree_qp_out_eliminate_x0(C.byref(self.qp_out))

    def set_x0(self, x0):
        lib().tree_qp_in_set_x0_colmaj(C.byref(self.qp_in), _dp(_f64(x0)))

    # ---- getters ----
    def _cat_vec(self, arr, n):
        return np.concatenate([np.ctypeslib.as_array(arr[i].pa, shape=(arr[i].m,)) if arr[i].m > 0 else np.zeros(0) for i in range(n)] or [np.zeros(0)]).copy()

    def _cat_mat(self, arr, n):
        return np.concatenate([np.ctypeslib.as_array(arr[i].pA, shape=(arr[i].m * arr[i].n,)) if arr[i].m * arr[i].n > 0 and arr[i].pA else np.zeros(0) for i in range(n)] or [np.zeros(0)]).copy()

    def flat(self) -> dict:
        """Flat ("ltv" order) copy of the data currently stored in the container."""
        q = self.qp_in
        N = self.N
        Qd = np.concatenate([np.asarray([q.Q[k].pA[j * q.Q[k].m + j] for j in range(q.Q[k].m)], dtype=float) for k in range(N)] or [np.zeros(0)])
        Rd = np.concatenate([np.asarray([q.R[k].pA[j * q.R[k].m + j] for j in range(q.R[k].m)], dtype=float) for k in range(N)] or [np.zeros(0)])
        return dict(nk=self.nk.copy(), nx=self.nx, nu=self.nu, A=self._cat_mat(q.A, N - 1), B=self._cat_mat(q.B, N - 1),
                    b=self._cat_vec(q.b, N - 1), Qd=Qd, Rd=Rd, q=self._cat_vec(q.q, N), r=self._cat_vec(q.r, N),
                    xmin=self._cat_vec(q.xmin, N), xmax=self._cat_vec(q.xmax, N),
                    umin=self._cat_vec(q.umin, N), umax=self._cat_vec(q.umax, N))

    def solution(self) -> dict:
        o = self.qp_out
        N = self.N
        return dict(x=self._cat_vec(o.x, N), u=self._cat_vec(o.u, N), lam=self._cat_vec(o.lam, N - 1),
                    mu_x=self._cat_vec(o.mu_x, N), mu_u=self._cat_vec(o.mu_u, N))

    def set_solution(self, sol: dict):
        """Write a flat solution into qp_out (used to exercise the host KKT check without a GPU)."""
        L = lib()
        off = {"x": 0, "u": 0, "lam": 0, "mu_x": 0, "mu_u": 0}
        o = self.qp_out
        for k in range(self.N):
            for name, arr, setter in (("x", o.x, L.tree_qp_out_set_node_x), ("u", o.u, L.tree_qp_out_set_node_u),
                                      ("mu_x", o.mu_x, L.tree_qp_out_set_node_mu_x), ("mu_u", o.mu_u, L.tree_qp_out_set_node_mu_u)):
                m = arr[k].m
                setter(_dp(_f64(sol[name][off[name]:off[name] + m])), C.byref(o), k)
                off[name] += m
            if k > 0:
                m = o.lam[k - 1].m
                L.tree_qp_out_set_edge_lam(_dp(_f64(sol["lam"][off["lam"]:off["lam"] + m])), C.byref(o), k - 1)
                off["lam"] += m

    def max_kkt_res(self) -> float:
        return float(lib().tree_qp_out_max_KKT_res(C.byref(self.qp_in), C.byref(self.qp_out)))

    @property
    def info(self):
        i = self.qp_out.info
        return dict(iter=i.iter, total_time=i.total_time, solver_time=i.solver_time, interface_time=i.interface_time)


# ------------------------------------------------------------------------------------------------
# TdunesSolver (reference: TdunesSolver in treeqp_cpp_interface.hpp:127-152, SetOption strings
# of treeqp_cpp_interface.cpp:185-262)
# ------------------------------------------------------------------------------------------------

class TdunesSolver:
    _INT_OPTS = {"maxIter", "checkLastActiveSet", "lineSearchMaxIter", "lineSearchRestartTrigger"}
    _DBL_OPTS = {"stationarityTolerance", "regTol", "regValue", "lineSearchGamma", "lineSearchBeta"}

    def __init__(self, qp: TreeQp, **options):
        L = lib()
        self.qp = qp
        N = qp.N
        self.opts = TdunesOpts()
        self._opts_mem = np.zeros(max(L.treeqp_tdunes_opts_calculate_size(N), 4), dtype=np.uint8)
        L.treeqp_tdunes_opts_create(N, C.byref(self.opts), C.c_void_p(self._opts_mem.ctypes.data))
        L.treeqp_tdunes_opts_set_default(N, C.byref(self.opts))
        for k, v in options.items():
            self.set_option(k, v)
        self.work = TdunesWork()
        self._created = False

    def set_option(self, name, value):
        # validate BEFORE assigning: a rejected value must not stay in the options struct
        if getattr(self, "_created", False) and name == "maxIter" and int(value) > self.work.maxIterAtCreate:
            raise ValueError("maxIter cannot be increased after the solver was created")
        if name in self._INT_OPTS:
            setattr(self.opts, name, int(value))
        elif name in self._DBL_OPTS:
            setattr(self.opts, name, float(value))
        elif name == "termCondition":
            self.opts.termCondition = TERMINATION[value] if isinstance(value, str) else int(value)
        elif name == "regType":
            self.opts.regType = REGTYPE[value] if isinstance(value, str) else int(value)
        elif name == "clipping":
            for k in range(self.qp.N):
                self.opts.qp_solver[k] = 0 if value else 1
        else:
            raise KeyError(f"unknown tdunes option {name!r}")

    def create(self):
        """treeqp_tdunes_calculate_size + caller-owned buffer + treeqp_tdunes_create."""
        if self._created:
            return self
        L = lib()
        if device_count() < 1:
            raise RuntimeError("treeqp_tdunes_create needs a HIP device (MI355X): none is visible and the "
                               "tdunes hot path has no CPU fallback")
        size = L.treeqp_tdunes_calculate_size(C.byref(self.qp.qp_in), C.byref(self.opts))
        self._mem, p = _aligned_buffer(size)
        L.treeqp_tdunes_create(C.byref(self.qp.qp_in), C.byref(self.opts), C.byref(self.work), p)
        self._created = True
        return self

    def set_dual_initialization(self, lam):
        self.create()
        lam = _f64(lam)
        lib().treeqp_tdunes_set_dual_initialization(_dp(lam), C.byref(self.work))

    def solve(self) -> int:
        self.create()
        return int(lib().treeqp_tdunes_solve(C.byref(self.qp.qp_in), C.byref(self.qp.qp_out), C.byref(self.opts), C.byref(self.work)))

    def idxpos(self):
        self.create()
        return np.ctypeslib.as_array(self.work.idxpos, shape=(self.qp.N,)).copy()

    def npar(self):
        self.create()
        Nh = int(self.qp.tree()["stage"][-1])
        return np.ctypeslib.as_array(self.work.npar, shape=(Nh + 1,)).copy()

    @property
    def ls_total(self):
        return int(self.work.lsTotal)

    def ls_iters(self):
        n = self.qp.qp_out.info.iter
        return np.ctypeslib.as_array(self.work.timings.ls_iters, shape=(self.work.timings.num_iter,))[:n].copy()

    def destroy(self):
        if self._created:
            lib().treeqp_tdunes_destroy(C.byref(self.work))
            self._created = False

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# TqGpu: the device C-ABI proper (include/treeqp_amd.h)
# ------------------------------------------------------------------------------------------------

def shard_unique_id() -> bytes:
    """128-byte RCCL unique id (call on rank 0, broadcast to the other ranks)."""
    buf = C.create_string_buffer(128)
    rc = lib().tqgpu_shard_unique_id(buf)
    if rc != 0:
        raise RuntimeError(f"tqgpu_shard_unique_id failed ({rc}): {lib().tqgpu_last_error().decode()}")
    return buf.raw


def solve_virtual_ranks(mirrors, **kw) -> dict:
    """Lock-step sharded solve of n mirrors of one problem in this process (diagnostic / tests)."""
    o = GpuOpts(maxIter=100, termCondition=2, stationarityTolerance=1e-8, regType=2, regTol=1e-6, regValue=1e-6,
                lineSearchMaxIter=50, lineSearchGamma=0.1, lineSearchBeta=0.6, lineSearchRestartTrigger=-1, profile=0, checkLastActiveSet=1)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    arr = (C.c_void_p * len(mirrors))(*[m.h for m in mirrors])
    r = GpuResult()
    rc = lib().tqgpu_solve_virtual_ranks(arr, len(mirrors), C.byref(o), C.byref(r))
    if rc != 0:
        raise RuntimeError(f"tqgpu_solve_virtual_ranks failed ({rc}): {lib().tqgpu_last_error().decode()}")
    return {f: getattr(r, f) for f, _ in GpuResult._fields_}


def _default_opts(**kw):
    o = GpuOpts(maxIter=100, termCondition=2, stationarityTolerance=1e-8, regType=2, regTol=1e-6, regValue=1e-6,
                lineSearchMaxIter=50, lineSearchGamma=0.1, lineSearchBeta=0.6, lineSearchRestartTrigger=-1, profile=0, checkLastActiveSet=1)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def pshard_solve_local(mirrors, **kw) -> list:
    """ONE tree over the n mirrors of this process, inside the persistent launch: n launches that wait for each other (tagged
    words in every mirror's slab).  Every mirror must be pshard_init(r, n); the solution is collected into every mirror."""
    o = _default_opts(**kw)
    n = len(mirrors)
    arr = (C.c_void_p * n)(*[m.h for m in mirrors])
    res = (GpuResult * n)()
    rc = lib().tqgpu_pshard_solve_local(arr, n, C.byref(o), res)
    if rc != 0:
        raise RuntimeError(f"tqgpu_pshard_solve_local failed ({rc}): {lib().tqgpu_last_error().decode()}")
    return [{f: getattr(res[i], f) for f, _ in GpuResult._fields_} for i in range(n)]


def solve_batch(mirrors, profile=0, **kw) -> list:
    """Batched multi-tree solve: independent mirrors, same options; persistent launches run concurrently."""
    o = GpuOpts(maxIter=100, termCondition=2, stationarityTolerance=1e-8, regType=2, regTol=1e-6, regValue=1e-6,
                lineSearchMaxIter=50, lineSearchGamma=0.1, lineSearchBeta=0.6, lineSearchRestartTrigger=-1, profile=profile, checkLastActiveSet=1)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    n = len(mirrors)
    arr = (C.c_void_p * n)(*[m.h for m in mirrors])
    res = (GpuResult * n)()
    rc = lib().tqgpu_solve_batch(arr, n, C.byref(o), res)
    if rc != 0:
        raise RuntimeError(f"tqgpu_solve_batch failed ({rc}): {lib().tqgpu_last_error().decode()}")
    return [{f: getattr(res[i], f) for f, _ in GpuResult._fields_} for i in range(n)]


def solve_batch_n(mirrors, steps: int, profile=0, **kw):
    """`steps` batched solves back to back in C (tqgpu_solve_batch_n) -> (results of the last call, sum of iterations, of trials, of launches)."""
    o = GpuOpts(maxIter=100, termCondition=2, stationarityTolerance=1e-8, regType=2, regTol=1e-6, regValue=1e-6,
                lineSearchMaxIter=50, lineSearchGamma=0.1, lineSearchBeta=0.6, lineSearchRestartTrigger=-1, profile=profile, checkLastActiveSet=1)
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    n = len(mirrors)
    arr = (C.c_void_p * n)(*[m.h for m in mirrors])
    res = (GpuResult * n)()
    it, ls, la = C.c_long(0), C.c_long(0), C.c_long(0)
    rc = lib().tqgpu_solve_batch_n(arr, n, C.byref(o), int(steps), res, C.byref(it), C.byref(ls), C.byref(la))
    if rc != 0:
        raise RuntimeError(f"tqgpu_solve_batch_n failed ({rc}): {lib().tqgpu_last_error().decode()}")
    return [{f: getattr(res[i], f) for f, _ in GpuResult._fields_} for i in range(n)], it.value, ls.value, la.value


class TqGpu:
    def __init__(self, nk, nx, nu, device: int = -1):
        L = lib()
        self.nk, self.nx, self.nu = _i32(nk), _i32(nx), _i32(nu)
        self.h = C.c_void_p()
        rc = L.tqgpu_create(C.byref(self.h), int(device), len(self.nk), _ip(self.nk), _ip(self.nx), _ip(self.nu))
        if rc != 0:
            raise RuntimeError(f"tqgpu_create failed ({rc}): {L.tqgpu_last_error().decode()}")
        d = [C.c_int() for _ in range(5)]
        L.tqgpu_dims(self.h, *[C.byref(v) for v in d])
        self.sum_nx, self.sum_nu, self.sum_lam, self.sum_A, self.sum_B = [v.value for v in d]

    @property
    def fused(self) -> bool:
        return bool(lib().tqgpu_uses_fused_path(self.h))

    def device_times(self, n: int) -> np.ndarray:
        """HIP-event device times [s] of the last n solves (oldest first); synchronises the stream."""
        out = np.zeros(int(n), dtype=np.float64)
        got = lib().tqgpu_get_device_times(self.h, _dp(out), int(n))
        if got < 0:
            raise RuntimeError("tqgpu_get_device_times failed")
        return out[:got]

    @property
    def path(self) -> int:
        """0 generic per-level kernels, 1 tiered fused kernels, 2 persistent single launch."""
        return int(lib().tqgpu_uses_fused_path(self.h))

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(f"tqgpu call failed ({rc}): {lib().tqgpu_last_error().decode()}")

    def upload(self, p, lambda0=None):
        g = (lambda k: getattr(p, k)) if not isinstance(p, dict) else (lambda k: p[k])
        L = lib()
        keep = {k: _f64(g(k)) for k in ("A", "B", "b", "Qd", "Rd", "q", "r", "xmin", "xmax", "umin", "umax")}
        assert len(keep["A"]) == self.sum_A and len(keep["B"]) == self.sum_B and len(keep["b"]) == self.sum_lam
        assert len(keep["Qd"]) == self.sum_nx and len(keep["Rd"]) == self.sum_nu
        self._chk(L.tqgpu_set_dynamics(self.h, _dp(keep["A"]), _dp(keep["B"]), _dp(keep["b"])))
        self._chk(L.tqgpu_set_objective_diag(self.h, _dp(keep["Qd"]), _dp(keep["Rd"]), _dp(keep["q"]), _dp(keep["r"])))
        self._chk(L.tqgpu_set_bounds(self.h, _dp(keep["xmin"]), _dp(keep["xmax"]), _dp(keep["umin"]), _dp(keep["umax"])))
        self.set_lambda(lambda0)
        return self

    def upload_dense(self, p, lambda0=None):
        """Dense objective (Q, R, S per node, column major) + the dense unconstrained stage solver."""
        g = (lambda k: getattr(p, k)) if not isinstance(p, dict) else (lambda k: p[k])
        L = lib()
        keep = {k: _f64(g(k)) for k in ("A", "B", "b", "Q", "R", "S", "q", "r")}
        assert len(keep["A"]) == self.sum_A and len(keep["B"]) == self.sum_B and len(keep["b"]) == self.sum_lam
        assert len(keep["Q"]) == int((self.nx.astype(np.int64) ** 2).sum()) and len(keep["R"]) == int((self.nu.astype(np.int64) ** 2).sum())
        assert len(keep["S"]) == int((self.nx.astype(np.int64) * self.nu).sum())
        self._chk(L.tqgpu_set_dynamics(self.h, _dp(keep["A"]), _dp(keep["B"]), _dp(keep["b"])))
        self._chk(L.tqgpu_set_objective_dense(self.h, _dp(keep["Q"]), _dp(keep["R"]), _dp(keep["S"]), _dp(keep["q"]), _dp(keep["r"])))
        self.set_lambda(lambda0)
        return self

    def upload_mixed(self, p, kind, lambda0=None):
        """Per-node choice of the stage solver: kind[k] = 0 clipping (diagonal Q_k, R_k, S_k = 0, box bounds), 1 dense unconstrained
        (full Q_k, R_k, S_k, bounds at infinity).  p holds A, B, b, Q, R, S (dense layout, column major), q, r and the bounds."""
        g = (lambda k: getattr(p, k)) if not isinstance(p, dict) else (lambda k: p[k])
        L = lib()
        keep = {k: _f64(g(k)) for k in ("A", "B", "b", "Q", "R", "S", "q", "r", "xmin", "xmax", "umin", "umax")}
        kd = _i32(kind)
        assert len(kd) == len(self.nk)
        self._chk(L.tqgpu_set_dynamics(self.h, _dp(keep["A"]), _dp(keep["B"]), _dp(keep["b"])))
        self._chk(L.tqgpu_set_objective_mixed(self.h, _ip(kd), _dp(keep["Q"]), _dp(keep["R"]), _dp(keep["S"]), _dp(keep["q"]), _dp(keep["r"])))
        self._chk(L.tqgpu_set_bounds(self.h, _dp(keep["xmin"]), _dp(keep["xmax"]), _dp(keep["umin"]), _dp(keep["umax"])))
        self.set_lambda(lambda0)
        return self

    def set_lambda(self, lam):
        a = None if lam is None else _f64(lam)
        if a is not None:
            assert len(a) == self.sum_lam
        self._chk(lib().tqgpu_set_lambda(self.h, _dp(a)))

    def export_ahead(self, on: bool) -> None:
        """the solution's packing kernel and download go out behind the solve's launch (callers that fetch the solution after every solve)"""
        self._chk(lib().tqgpu_set_export_ahead(self.h, int(bool(on))))

    def event_timing(self, on: bool) -> None:
        """Per-solve HIP event pairs on / off (off: device_times() gives NaN for single-launch solves)."""
        self._chk(lib().tqgpu_set_event_timing(self.h, int(bool(on))))

    def solve(self, profile=0, **kw) -> dict:
        key = (profile, tuple(sorted(kw.items())))
        cache = self.__dict__.setdefault("_opts_cache", {})
        o = cache.get(key)
        if o is None:                       # the options struct of a (profile, kwargs) combination is built once
            o = GpuOpts(maxIter=100, termCondition=2, stationarityTolerance=1e-8, regType=2, regTol=1e-6, regValue=1e-6,
                        lineSearchMaxIter=50, lineSearchGamma=0.1, lineSearchBeta=0.6, lineSearchRestartTrigger=-1, profile=profile, checkLastActiveSet=1)
            for k, v in kw.items():
                if not hasattr(o, k):
                    raise KeyError(k)
                setattr(o, k, v)
            cache[key] = o
        r = GpuResult()
        self._chk(lib().tqgpu_solve(self.h, C.byref(o), C.byref(r)))
        return {f: getattr(r, f) for f, _ in GpuResult._fields_}

    def solve_n(self, n: int, profile=0, **kw):
        """n solves back to back (the reference drivers' NREP loop, in C) -> (last result, sum of iterations, of trials, of launches)."""
        key = (profile, tuple(sorted(kw.items())))
        cache = self.__dict__.setdefault("_opts_cache", {})
        o = cache.get(key)
        if o is None:
            o = GpuOpts(maxIter=100, termCondition=2, stationarityTolerance=1e-8, regType=2, regTol=1e-6, regValue=1e-6,
                        lineSearchMaxIter=50, lineSearchGamma=0.1, lineSearchBeta=0.6, lineSearchRestartTrigger=-1, profile=profile, checkLastActiveSet=1)
            for k, v in kw.items():
                if not hasattr(o, k):
                    raise KeyError(k)
                setattr(o, k, v)
            cache[key] = o
        r = GpuResult()
        it, ls, la = C.c_long(0), C.c_long(0), C.c_long(0)
        self._chk(lib().tqgpu_solve_n(self.h, C.byref(o), int(n), C.byref(r), C.byref(it), C.byref(ls), C.byref(la)))
        return {f: getattr(r, f) for f, _ in GpuResult._fields_}, it.value, ls.value, la.value

    def solution(self) -> dict:
        out = dict(x=np.zeros(self.sum_nx), u=np.zeros(self.sum_nu), lam=np.zeros(self.sum_lam),
                   mu_x=np.zeros(self.sum_nx), mu_u=np.zeros(self.sum_nu), dlam=np.zeros(self.sum_lam))
        self._chk(lib().tqgpu_get_solution(self.h, _dp(out["x"]), _dp(out["u"]), _dp(out["lam"]), _dp(out["mu_x"]), _dp(out["mu_u"]), _dp(out["dlam"])))
        return out

    def iteration_log(self, cap=4096):
        ls = np.zeros(cap, dtype=np.int32)
        tt = np.full(cap, np.nan)
        self._chk(lib().tqgpu_get_iteration_log(self.h, _ip(ls), _dp(tt), cap))
        return ls, tt

    def geometry(self) -> dict:
        """block levels, tiers and workgroups of the persistent launch, and how many such workgroups the device holds at once"""
        v = [C.c_int() for _ in range(5)]
        self._chk(lib().tqgpu_geometry(self.h, *[C.byref(x) for x in v]))
        return dict(zip(("levels", "tiers", "workgroups", "capacity", "compute_units"), [x.value for x in v]))

    def iteration_cost(self, n_ls=1):
        b, f = C.c_double(), C.c_double()
        self._chk(lib().tqgpu_iteration_cost(self.h, int(n_ls), C.byref(b), C.byref(f)))
        return b.value, f.value

    # ---- one tree sharded over several devices ----
    def shard_init(self, rank: int, nranks: int, unique_id: bytes | None = None):
        """Restrict this mirror to rank `rank` of `nranks`.  unique_id: 128-byte RCCL id from
        ``shard_unique_id()`` of rank 0 (broadcast by the caller); None = virtual rank (no communicator)."""
        buf = None if unique_id is None else C.create_string_buffer(bytes(unique_id), 128)
        self._chk(lib().tqgpu_shard_init(self.h, int(rank), int(nranks), buf))
        return self

    def shard_gather_solution(self):
        self._chk(lib().tqgpu_shard_gather_solution(self.h))

    # ---- one tree sharded over several devices inside the persistent launch ----
    def pshard_init(self, rank: int, nranks: int):
        self._chk(lib().tqgpu_pshard_init(self.h, int(rank), int(nranks)))
        return self

    def pshard_ipc_export(self) -> bytes:
        buf = C.create_string_buffer(64)
        self._chk(lib().tqgpu_pshard_ipc_export(self.h, buf))
        return bytes(buf.raw)

    def pshard_ipc_connect(self, r: int, handle: bytes):
        self._chk(lib().tqgpu_pshard_ipc_connect(self.h, int(r), C.create_string_buffer(bytes(handle), 64)))

    def pshard_begin(self, **kw):
        o = _default_opts(**kw)
        self._chk(lib().tqgpu_pshard_begin(self.h, C.byref(o)))

    def pshard_rewind(self):
        self._chk(lib().tqgpu_pshard_rewind(self.h))

    def pshard_end(self) -> dict:
        r = GpuResult()
        self._chk(lib().tqgpu_pshard_end(self.h, C.byref(r)))
        return {f: getattr(r, f) for f, _ in GpuResult._fields_}

    def pshard_pack(self) -> np.ndarray:
        L = lib()
        L.tqgpu_pshard_pack_size.restype = C.c_long
        n = int(L.tqgpu_pshard_pack_size(self.h))
        out = np.zeros(max(n, 1), dtype=np.float64)
        self._chk(L.tqgpu_pshard_pack(self.h, _dp(out), C.c_long(n)))
        return out

    def pshard_unpack(self, src_rank: int, buf: np.ndarray):
        buf = _f64(buf)
        self._chk(lib().tqgpu_pshard_unpack(self.h, int(src_rank), _dp(buf), C.c_long(len(buf))))

    def close(self):
        if self.h:
            lib().tqgpu_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
