"""BFGS as the reference's drivers ask scipy for it (``scipy.optimize.minimize(method="BFGS", tol=...)``,
ref:openvqe/ucc_family/get_energy_qucc.py:158-175, get_energy_ucc.py:158-175) — same line search (scipy's own
Wolfe search), same convergence test (max |gradient| <= tol), same update formula — with the inverse-Hessian
update written in its rank-two form.

scipy (1.15) evaluates  H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T  with two dense n x n matrix
products per iteration: 2 x 2 n^3 flops.  At the 1715 parameters of N2 / cc-pVDZ (10e,12o) that is 20 GFLOP of
host BLAS per iteration — 12 of the 14 seconds of the QUCCSD ``get_energies`` run were spent there, four times
what the device needs for the energies and gradients.  Expanded, the same expression is
    H - rho (s (H y)^T + (H y) s^T) + (rho^2 y^T H y + rho) s s^T          (H symmetric),
one matrix-vector product and a symmetric rank-two update: O(n^2) — and written so that the matrix is passed over twice per
iteration (BLAS dsymv + dsyr2 on one triangle, in place; H y follows from H g_new and the search direction, the next direction
from the same product): 1.3 ms per iteration at n = 1715 on one core.  The iterates agree with scipy's to rounding.

Used by the mirrors for n >= ``RANK_TWO_FROM`` parameters when a Jacobian is supplied; below that scipy itself
runs (the stored traces of the small molecules are reproduced call for call).
"""
from __future__ import annotations

import numpy as np
from scipy.linalg.blas import dsymv as _dsymv, dsyr2 as _dsyr2
from scipy.optimize import OptimizeResult

from .host_threads import one_blas_thread

RANK_TWO_FROM = 256


def _line_search():
    """scipy's own Wolfe search (MINPACK2's dcsrch, falling back on its Python search) — what scipy's BFGS calls, and what the
    "same iterates" claim above rests on.  It is a private name (validated on scipy 1.9 ... 1.15): where a scipy release no longer
    has it, -> None and `minimize_bfgs` hands the whole minimisation to scipy's BFGS (dense update, same iterates) rather than
    running on a DIFFERENT line search without saying so."""
    try:
        from scipy.optimize._optimize import _LineSearchError, _line_search_wolfe12
        return _line_search_wolfe12, _LineSearchError
    except ImportError:
        return None


_warned = []


def minimize_bfgs(fun, x0, jac, tol=None, maxiter=None, disp=False, c1=1e-4, c2=0.9):
    """-> OptimizeResult with the fields of scipy's BFGS (x, fun, jac, hess_inv, nit, nfev, njev, status, success, message).
    The O(n^2) host algebra of an iteration runs on ONE BLAS thread: a multi-threaded BLAS leaves its workers spinning after every
    call, and the device calls that follow — a hundred kernel launches each — then take three times as long (measured on the
    MI355X box, N2 QUCCSD gradient: 31 ms -> 89 ms with 10 ms of threaded numpy between the calls, tools/exp_mirror_eval_n2.py)."""
    with one_blas_thread():
        if _line_search() is None:
            if not _warned:
                import warnings
                warnings.warn("scipy.optimize._optimize._line_search_wolfe12 is not importable in this scipy release: the rank-two BFGS "
                              "is bypassed and scipy's own BFGS (dense inverse-Hessian update) runs instead", RuntimeWarning, stacklevel=2)
                _warned.append(True)
            from scipy.optimize import minimize
            opts = {"disp": disp, "c1": c1, "c2": c2}
            if maxiter is not None:
                opts["maxiter"] = maxiter
            return minimize(fun, x0, jac=jac, method="BFGS", tol=tol, options=opts)
        return _minimize_bfgs(fun, x0, jac, tol, maxiter, disp, c1, c2)


def _minimize_bfgs(fun, x0, jac, tol, maxiter, disp, c1, c2):
    search, LineSearchError = _line_search()
    x0 = np.asarray(x0, dtype=float).flatten()
    n = x0.size
    gtol = 1e-5 if tol is None else tol
    maxiter = n * 200 if maxiter is None else maxiter
    count = {"f": 0, "g": 0}

    def f(x):
        count["f"] += 1
        return float(fun(np.copy(x)))

    def g(x):
        count["g"] += 1
        return np.asarray(jac(np.copy(x)), dtype=float)

    old_fval = f(x0)
    gfk = g(x0)
    k = 0
    # The inverse Hessian lives in ONE triangle of a Fortran-ordered array and is touched twice per iteration: dsymv for H g and dsyr2
    # for the update, in place (numpy's outer products and sums made about ten passes over the 1715 x 1715 matrix: 12 of the 40 ms of
    # an N2 QUCCSD iteration).  H y needs no product of its own: y = g_new - g_old and H g_old = -p are known.
    Hk = np.asfortranarray(np.eye(n))
    old_old_fval = old_fval + np.linalg.norm(gfk) / 2
    xk = x0
    warnflag = 0
    gnorm = np.abs(gfk).max() if n else 0.0
    pk = -gfk.copy()   # H = 1
    while gnorm > gtol and k < maxiter:
        try:
            alpha_k, _, _, old_fval, old_old_fval, gfkp1 = search(f, g, xk, pk, gfk, old_fval, old_old_fval, amin=1e-100, amax=1e100,
                                                                  c1=c1, c2=c2)
        except LineSearchError:
            warnflag = 2
            break
        sk = alpha_k * pk
        xk = xk + sk
        if gfkp1 is None:
            gfkp1 = g(xk)
        yk = gfkp1 - gfk
        gfk = gfkp1
        k += 1
        gnorm = np.abs(gfk).max()
        if gnorm <= gtol:
            break
        if not np.isfinite(old_fval):
            warnflag = 2
            break
        rhok_inv = float(yk @ sk)
        rhok = 1000.0 if rhok_inv == 0.0 else 1.0 / rhok_inv
        Hg = _dsymv(1.0, Hk, gfk, lower=1)          # H_old g_new
        Hy = Hg + pk                                # H_old (g_new - g_old),  H_old g_old = -p
        yHy = float(yk @ Hy)
        cs = rhok * rhok * yHy + rhok
        # H_new = H - rho (s Hy^T + Hy s^T) + cs s s^T = H + s w^T + w s^T,  w = -rho Hy + cs/2 s
        wk = -rhok * Hy + (0.5 * cs) * sk
        Hk = _dsyr2(1.0, sk, wk, a=Hk, lower=1, overwrite_a=1)
        # next direction: -H_new g_new from H_old g_new (no second pass over the matrix)
        sg, wg = float(sk @ gfk), float(wk @ gfk)
        pk = -(Hg + sk * wg + wk * sg)
    fval = old_fval
    Hk = np.tril(Hk) + np.tril(Hk, -1).T   # the full symmetric matrix of scipy's result
    if warnflag == 2:
        msg = "Desired error not necessarily achieved due to precision loss."
    elif k >= maxiter:
        warnflag = 1
        msg = "Maximum number of iterations has been exceeded."
    elif np.isnan(gnorm) or np.isnan(fval) or np.isnan(xk).any():
        warnflag = 3
        msg = "NaN result encountered."
    else:
        msg = "Optimization terminated successfully."
    if disp:
        print(("Warning: " if warnflag else "") + msg)
        print("         Current function value: %f" % fval)
        print("         Iterations: %d" % k)
        print("         Function evaluations: %d" % count["f"])
        print("         Gradient evaluations: %d" % count["g"])
    return OptimizeResult(fun=fval, jac=gfk, hess_inv=Hk, nfev=count["f"], njev=count["g"], status=warnflag, success=(warnflag == 0),
                          message=msg, x=xk, nit=k)


def minimize(fun, x0, jac=None, method="BFGS", tol=None, options=None):
    """``scipy.optimize.minimize`` for the mirrors' BFGS runs: the rank-two implementation above for >= RANK_TWO_FROM parameters
    with a Jacobian, scipy otherwise"""
    import scipy.optimize
    options = dict(options or {})
    if method == "BFGS" and callable(jac) and np.size(x0) >= RANK_TWO_FROM:
        return minimize_bfgs(fun, x0, jac, tol=tol, maxiter=options.get("maxiter"), disp=options.get("disp", False))
    with one_blas_thread():   # scipy's own run, its host algebra on one BLAS thread too (host_threads.py)
        return scipy.optimize.minimize(fun, x0=x0, jac=jac, method=method, tol=tol, options=options)
