"""ctypes access to the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (treeqp_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)


class OracleOpts(C.Structure):
    _fields_ = [("maxIter", C.c_int), ("termCondition", C.c_int), ("stationarityTolerance", C.c_double),
                ("checkLastActiveSet", C.c_int), ("lineSearchMaxIter", C.c_int), ("lineSearchGamma", C.c_double),
                ("lineSearchBeta", C.c_double), ("lineSearchRestartTrigger", C.c_int), ("regType", C.c_int),
                ("regTol", C.c_double), ("regValue", C.c_double), ("num_threads", C.c_int)]


class OracleInfo(C.Structure):
    _fields_ = [("status", C.c_int), ("iter", C.c_int), ("ls_total", C.c_int), ("n_active", C.c_int),
                ("n_regularized", C.c_int), ("solver_time", C.c_double), ("trace_err", c_dbl_p),
                ("trace_fval", c_dbl_p), ("trace_ls", c_int_p)]


def lib():
    global _LIB
    if _LIB is None:
        so = _HERE / "liboracle.so"
        if not so.exists():
            subprocess.run(["make", "-C", str(_HERE)], check=True)
        _LIB = C.CDLL(str(so))
        _LIB.oracle_max_kkt.restype = C.c_double
    return _LIB


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dbl_p)


def default_opts(**kw) -> OracleOpts:
    o = OracleOpts()
    lib().oracle_opts_set_default(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise KeyError(k)
        setattr(o, k, v)
    return o


def calculate_number_of_nodes(md, Nr, Nh):
    return lib().oracle_calculate_number_of_nodes(md, Nr, Nh)


def setup_multistage_tree(md, Nr, Nh):
    n = calculate_number_of_nodes(md, Nr, Nh)
    nk = np.zeros(n, dtype=np.int32)
    lib().oracle_setup_multistage_tree(md, Nr, Nh, _ip(nk))
    return nk


def tree_arrays(nk, nx=None):
    nk = _i(nk)
    n = len(nk)
    out = {k: np.zeros(n, dtype=np.int32) for k in ("dad", "stage", "real", "idxkid", "kid0")}
    Np = lib().oracle_tree_create(n, _ip(nk), *[_ip(out[k]) for k in ("dad", "stage", "real", "idxkid", "kid0")])
    out["Np"] = Np
    out["Nn_from_nk"] = lib().oracle_number_of_nodes_from_nkids(_ip(nk))
    if nx is not None:
        nx = _i(nx)
        pos = np.zeros(n, dtype=np.int32)
        lib().oracle_setup_idxpos(n, _ip(out["dad"]), _ip(out["idxkid"]), _ip(out["kid0"]), _ip(nx), _ip(pos))
        out["idxpos"] = pos
        Nh = int(out["stage"][-1])
        npar = np.zeros(Nh + 1, dtype=np.int32)
        lib().oracle_setup_npar(n, _ip(out["stage"]), Nh, _ip(npar))
        out["npar"] = npar
    return out


def fill_lti_diag(nk, nx, nu, A, B, b, Qd, q, Pd, p, Rd, r, xmin, xmax, umin, umax, x0):
    """Flat QP arrays from LTI data (tree_qp_common.c:1837-1949).  Returns a dict."""
    nk = _i(nk)
    Nn = len(nk)
    Np = int((nk > 0).sum())
    o = dict(A=np.zeros((Nn - 1) * nx * nx), B=np.zeros((Nn - 1) * nx * nu), b=np.zeros((Nn - 1) * nx),
             Qd=np.zeros(Nn * nx), Rd=np.zeros(Np * nu), q=np.zeros(Nn * nx), r=np.zeros(Np * nu),
             xmin=np.zeros(Nn * nx), xmax=np.zeros(Nn * nx), umin=np.zeros(Np * nu), umax=np.zeros(Np * nu))
    args = [_d(v) for v in (A, B, b, Qd, q, Pd, p, Rd, r, xmin, xmax, umin, umax, x0)]
    lib().oracle_fill_lti_diag(Nn, _ip(nk), int(nx), int(nu), *[_dp(a) for a in args],
                               *[_dp(o[k]) for k in ("A", "B", "b", "Qd", "Rd", "q", "r", "xmin", "xmax", "umin", "umax")])
    o["nk"] = nk
    o["nx"] = np.full(Nn, nx, dtype=np.int32)
    o["nu"] = np.where(nk > 0, nu, 0).astype(np.int32)
    return o


def solve(qp, opts: OracleOpts | None = None, lambda0=None, traces=True):
    """qp: mapping with nk,nx,nu,A,B,b,Qd,Rd,q,r,xmin,xmax,umin,umax (flat).  Returns a dict."""
    opts = opts or default_opts()
    nk, nx, nu = _i(qp["nk"]), _i(qp["nx"]), _i(qp["nu"])
    Nn = len(nk)
    sx, su = int(nx.sum()), int(nu.sum())
    sl = sx - int(nx[0])
    arrs = {k: _d(qp[k]) for k in ("A", "B", "b", "Qd", "Rd", "q", "r", "xmin", "xmax", "umin", "umax")}
    out = dict(x=np.zeros(sx), u=np.zeros(su), lam=np.zeros(sl), mu_x=np.zeros(sx), mu_u=np.zeros(su))
    info = OracleInfo()
    n = max(opts.maxIter, 0) + 1
    te, tf, tl = np.full(n, np.nan), np.full(n, np.nan), np.zeros(n, dtype=np.int32)
    if traces:
        info.trace_err, info.trace_fval, info.trace_ls = _dp(te), _dp(tf), _ip(tl)
    l0 = None if lambda0 is None else _d(lambda0)
    status = lib().oracle_tdunes_solve(
        Nn, _ip(nk), _ip(nx), _ip(nu), *[_dp(arrs[k]) for k in ("A", "B", "b", "Qd", "Rd", "q", "r", "xmin", "xmax", "umin", "umax")],
        C.byref(opts), _dp(l0), _dp(out["x"]), _dp(out["u"]), _dp(out["lam"]), _dp(out["mu_x"]), _dp(out["mu_u"]), C.byref(info))
    out.update(status=status, iter=info.iter, ls_total=info.ls_total, n_active=info.n_active,
               n_regularized=info.n_regularized, solver_time=info.solver_time,
               trace_err=te, trace_fval=tf, trace_ls=tl)
    return out


def solve_dense(qp, opts: OracleOpts | None = None, lambda0=None):
    """Dense unconstrained stage solver variant; qp has Q,R,S (flat, column major) instead of Qd,Rd."""
    opts = opts or default_opts()
    nk, nx, nu = _i(qp["nk"]), _i(qp["nx"]), _i(qp["nu"])
    Nn = len(nk)
    sx, su = int(nx.sum()), int(nu.sum())
    sl = sx - int(nx[0])
    arrs = {k: _d(qp[k]) for k in ("A", "B", "b", "Q", "R", "S", "q", "r")}
    out = dict(x=np.zeros(sx), u=np.zeros(su), lam=np.zeros(sl))
    info = OracleInfo()
    l0 = None if lambda0 is None else _d(lambda0)
    status = lib().oracle_tdunes_solve_dense(
        Nn, _ip(nk), _ip(nx), _ip(nu), *[_dp(arrs[k]) for k in ("A", "B", "b", "Q", "R", "S", "q", "r")],
        C.byref(opts), _dp(l0), _dp(out["x"]), _dp(out["u"]), _dp(out["lam"]), C.byref(info))
    out.update(status=status, iter=info.iter, ls_total=info.ls_total)
    return out


def max_kkt(qp, sol, dense=False):
    nk, nx, nu = _i(qp["nk"]), _i(qp["nx"]), _i(qp["nu"])
    g = lambda k: _dp(_d(qp[k])) if qp.get(k) is not None else None
    s = lambda k: _dp(_d(sol[k])) if sol.get(k) is not None else None
    keep = []

    def arr(src, k):
        v = src.get(k)
        if v is None:
            return None
        a = _d(v)
        keep.append(a)
        return _dp(a)

    return lib().oracle_max_kkt(
        len(nk), _ip(nk), _ip(nx), _ip(nu), arr(qp, "A"), arr(qp, "B"), arr(qp, "b"),
        None if dense else arr(qp, "Qd"), None if dense else arr(qp, "Rd"),
        arr(qp, "Q") if dense else None, arr(qp, "R") if dense else None, arr(qp, "S") if dense else None,
        arr(qp, "q"), arr(qp, "r"), arr(qp, "xmin"), arr(qp, "xmax"), arr(qp, "umin"), arr(qp, "umax"),
        arr(sol, "x"), arr(sol, "u"), arr(sol, "lam"), arr(sol, "mu_x"), arr(sol, "mu_u"))
