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
