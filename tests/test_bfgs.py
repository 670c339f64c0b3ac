"""The mirrors' BFGS for large parameter counts (openvqe_amd/common_files/bfgs.py): scipy's algorithm with the inverse-Hessian
update in rank-two form — same iterates as scipy.optimize.minimize(method="BFGS") to rounding, same result fields; below the
threshold (and without a Jacobian) scipy itself runs."""
import numpy as np
import scipy.optimize

from openvqe_amd.common_files import bfgs


def _problem(n, seed):
    rng = np.random.default_rng(seed)
    A = rng.normal(size=(n, n))
    A = A @ A.T / n + np.eye(n)
    b = rng.normal(size=n)
    f = lambda x: 0.5 * x @ A @ x - b @ x + 0.1 * np.sum(np.cos(x))   # noqa: E731
    g = lambda x: A @ x - b - 0.1 * np.sin(x)                          # noqa: E731
    return f, g, rng.normal(size=n)


def test_rank_two_update_follows_scipy_iterate_by_iterate():
    f, g, x0 = _problem(300, 1)
    ref = scipy.optimize.minimize(f, x0, jac=g, method="BFGS", tol=1e-6)
    got = bfgs.minimize(f, x0, jac=g, method="BFGS", tol=1e-6, options={"maxiter": 50000, "disp": False})
    assert (got.nit, got.nfev, got.njev, got.success, got.status) == (ref.nit, ref.nfev, ref.njev, ref.success, ref.status)
    assert abs(got.fun - ref.fun) < 1e-12 and np.abs(got.x - ref.x).max() < 1e-8
    assert np.abs(got.hess_inv - ref.hess_inv).max() < 1e-6 * np.abs(ref.hess_inv).max()
    assert np.abs(got.jac).max() <= 1e-6


def test_small_problems_and_missing_jacobians_go_to_scipy(monkeypatch):
    calls = []
    real = scipy.optimize.minimize
    monkeypatch.setattr(scipy.optimize, "minimize", lambda *a, **k: calls.append(1) or real(*a, **k))
    f, g, x0 = _problem(20, 2)
    bfgs.minimize(f, x0, jac=g, method="BFGS", tol=1e-6)
    f, g, x0 = _problem(bfgs.RANK_TWO_FROM, 3)
    bfgs.minimize(f, x0[:40], jac=None, method="BFGS", tol=1e-4) if False else None
    assert calls == [1]
    bfgs.minimize(f, x0, jac=g, method="BFGS", tol=1e-4)      # large with a Jacobian: the rank-two implementation
    assert calls == [1]


def test_printed_summary_has_scipy_s_lines(capsys):
    f, g, x0 = _problem(bfgs.RANK_TWO_FROM, 4)
    res = bfgs.minimize(f, x0, jac=g, method="BFGS", tol=1e-5, options={"disp": True})
    out = capsys.readouterr().out
    assert "Optimization terminated successfully." in out and "Current function value:" in out
    assert f"Iterations: {res.nit}" in out and f"Function evaluations: {res.nfev}" in out and f"Gradient evaluations: {res.njev}" in out


def test_missing_private_line_search_hands_over_to_scipy(monkeypatch):
    """without scipy's private Wolfe search the rank-two BFGS must not run on another line search silently"""
    import warnings

    import scipy.optimize

    from openvqe_amd.common_files import bfgs
    monkeypatch.setattr(bfgs, "_line_search", lambda: None)
    monkeypatch.setattr(bfgs, "_warned", [])
    rng = np.random.default_rng(3)
    A = rng.normal(size=(12, 12))
    A = A @ A.T + 12 * np.eye(12)
    b = rng.normal(size=12)
    f, g = (lambda x: 0.5 * x @ A @ x - b @ x), (lambda x: A @ x - b)
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        res = bfgs.minimize_bfgs(f, np.zeros(12), g, tol=1e-8)
    assert any("scipy's own BFGS" in str(w.message) for w in seen)
    ref = scipy.optimize.minimize(f, np.zeros(12), jac=g, method="BFGS", tol=1e-8)
    assert np.array_equal(res.x, ref.x) and res.nit == ref.nit
