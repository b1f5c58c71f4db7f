"""CPU tests of the batched variable-metric minimiser on analytic objectives (no engine involved)."""
import numpy as np
import pytest
from scipy import optimize

from vega_amd.minimizer import BatchedMinimizer, ParameterTransform


def test_transform_round_trip_and_bounds():
    tr = ParameterTransform([(0., 2.), (None, None), (1., None), (None, 3.)])
    ext = np.array([[0.3, -5., 4., -2.], [1.9, 0.1, 1.0, 3.0]])
    x = tr.to_internal(ext)
    np.testing.assert_allclose(tr.to_external(x), ext, rtol=1e-9, atol=1e-9)
    wild = tr.to_external(np.array([[50., 7., -9., 12.]]))
    assert 0. <= wild[0, 0] <= 2. and wild[0, 2] >= 1. and wild[0, 3] <= 3.
    eps = 1e-6
    num = (tr.to_external(x + eps) - tr.to_external(x - eps)) / (2 * eps)
    np.testing.assert_allclose(tr.jacobian(x), num, rtol=1e-5, atol=1e-8)


def _quadratic_problem(n_fits, seed=0):
    rng = np.random.default_rng(seed)
    P = 5
    A = rng.standard_normal((P, P))
    H = A @ A.T + P * np.eye(P)                 # chi2 = (t - c)^T H (t - c) + offset
    centres = rng.standard_normal((n_fits, P)) * 0.3 + np.array([1.0, 0.5, -0.2, 2.0, 0.0])

    def evaluate(theta, fit):
        d = theta - centres[fit]
        return np.einsum('ni,ij,nj->n', d, H, d) + 3.0
    return evaluate, H, centres


def test_many_quadratic_fits_in_lockstep():
    evaluate, H, centres = _quadratic_problem(40)
    names = [f'p{i}' for i in range(5)]
    names[1] = 'bias_x'
    fitter = BatchedMinimizer(evaluate, names, start=[0.8, 0.3, 0., 1.5, 0.2], errors=[0.1] * 5,
                              limits=[(None, None), (-3., 3.), (None, None), (0., 5.), (None, None)])
    res = fitter.minimize(n_fits=40)
    assert res.is_valid.all()
    # Minuit's criterion: estimated distance to the minimum below 0.002 * tol * errordef = 2e-4 in chi2,
    # i.e. parameters within a few per cent of their errors
    assert np.all(np.abs(res.values - centres) < 0.05 * res.errors)
    assert np.all(res.fval - 3.0 < 1e-3) and np.all(res.fval >= 3.0 - 1e-12)
    tight = BatchedMinimizer(evaluate, names, start=[0.8, 0.3, 0., 1.5, 0.2], errors=[0.1] * 5, tol=1e-6,
                             limits=[(None, None), (-3., 3.), (None, None), (0., 5.), (None, None)]).minimize(n_fits=40)
    np.testing.assert_allclose(tight.values, centres, atol=2e-5)
    # covariance = 2 errordef H^-1 for chi2 = d^T H d  ->  (d^2 chi2 / dt^2 = 2 H)  =>  cov = H^-1
    np.testing.assert_allclose(res.covariance[0], np.linalg.inv(H), rtol=2e-3, atol=1e-6)
    assert res.nfcn.max() < 600


def test_bounded_minimum_and_fixed_parameters():
    def evaluate(theta, fit):
        return (theta[:, 0] - 2.0)**2 / 0.04 + (theta[:, 1] + 1.0)**2 / 0.09 + 10 * (theta[:, 2] - 0.5)**2
    fitter = BatchedMinimizer(evaluate, ['a', 'b', 'c'], start=[0.5, 0.0, 0.1], errors=[0.1, 0.1, 0.1],
                              limits=[(0., 1.5), (None, None), (None, None)])
    res = fitter.minimize(n_fits=1, fixed=('c',))
    assert res.values[0, 0] == pytest.approx(1.5, abs=5e-3)        # pinned at the upper limit
    assert res.values[0, 1] == pytest.approx(-1.0, abs=0.05 * 0.3)
    assert res.values[0, 2] == 0.1
    assert res.errors[0, 1] == pytest.approx(0.3, rel=1e-2)


def test_matches_scipy_on_a_curved_valley():
    def f(t):
        return 100 * (t[1] - t[0]**2)**2 + (1 - t[0])**2 + 2.0

    def evaluate(theta, fit):
        return np.array([f(t) for t in theta])
    fitter = BatchedMinimizer(evaluate, ['x', 'y'], start=[-0.5, 0.8], errors=[0.1, 0.1],
                              limits=[(-2., 2.), (-1., 3.)], max_iter=400)
    res = fitter.minimize(n_fits=1)
    ref = optimize.minimize(f, [-0.5, 0.8], method='L-BFGS-B', bounds=[(-2, 2), (-1, 3)])
    assert res.fval[0] == pytest.approx(ref.fun, abs=1e-3)
    assert np.all(np.abs(res.values[0] - 1.) < 0.1 * res.errors[0] + 1e-3)


def test_model_failures_are_avoided():
    def evaluate(theta, fit):
        out = (theta[:, 0] - 1.0)**2 + (theta[:, 1] - 2.0)**2
        out[theta[:, 0] > 1.4] = 1e100          # the engine's sentinel for points it cannot evaluate
        return out
    fitter = BatchedMinimizer(evaluate, ['a', 'b'], start=[0.2, 0.2], errors=[0.5, 0.5],
                              limits=[(None, None), (None, None)])
    res = fitter.minimize(n_fits=3)
    np.testing.assert_allclose(res.values, [[1., 2.]] * 3, atol=0.05)
