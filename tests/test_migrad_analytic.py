"""MIGRAD restated (vega_amd/migrad.py) against outcomes that are known INDEPENDENTLY of this repository: analytic minima,
the error definition of MINUIT (a parameter's error is where the function rises by `up` = 1: covariance = 2 up H^-1, F. James,
MINUIT reference manual, section 1.3 / 7.1), its convergence criterion (EDM < 0.002 tol up: Minuit2 user's guide, MnMigrad) and
its treatment of limits.  iminuit is absent from the image, so no trajectory can be compared call for call; what the restatement
must get right whatever its trajectory is asserted here.  (The fixtures of tests/test_fits_gpu.py were produced by the reference
driving THIS restatement behind the iminuit surface: they pin the drivers around the minimiser, not the minimiser.)"""
import numpy as np
import pytest

from vega_amd import migrad
from vega_amd.migrad import MigradMinimizer

EDM_GOAL = 0.002 * 0.1 * 1.0          # tol = 0.1, up = 1


def _fit(fun, names, start, errors, limits, n_fits=1, **kw):
    calls = {'n': 0}

    def evaluate(theta, fit):
        calls['n'] += theta.shape[0]
        return np.array([fun(row) for row in theta])
    m = MigradMinimizer(evaluate, names, start, errors, limits, **kw)
    return m.minimize(n_fits, prefit_bias=False), calls


@pytest.mark.parametrize('vectorised', [True, False])
def test_quadratic_minimum_errors_and_covariance(vectorised):
    A = np.array([[4.0, 1.0, 0.5], [1.0, 3.0, -0.7], [0.5, -0.7, 2.0]]) * 25.0
    c = np.array([0.3, -0.2, 0.1])
    res, calls = _fit(lambda x: float((x - c) @ A @ (x - c)), ['a', 'b', 'c'], [0., 0., 0.], [0.1, 0.1, 0.1],
                      [(None, None)] * 3, vectorised=vectorised)
    cov = np.linalg.inv(A)              # f = d^T A d: Hessian 2 A, covariance 2 up (2 A)^-1 = A^-1
    assert res.is_valid[0] and not res.hesse_failed[0]
    assert res.edm[0] < EDM_GOAL
    sigma = np.sqrt(np.diag(cov))
    assert np.abs(res.values[0] - c).max() < 1e-3 * sigma.min()
    np.testing.assert_allclose(res.errors[0], sigma, rtol=2e-3)
    np.testing.assert_allclose(res.covariance[0], cov, rtol=5e-3, atol=1e-3 * cov.max())
    assert res.fval[0] < 1e-6
    # an exact quadratic needs the seed (1 + 2 n calls per gradient cycle), one Newton step with its line search and one
    # HESSE (diagonal 2 n + off-diagonal n (n - 1) / 2 calls): a few dozen calls, not hundreds
    assert calls['n'] == res.nfcn[0] and res.nfcn[0] < 70


def test_one_parameter_parabola_reports_sigma():
    res, _ = _fit(lambda x: float(((x[0] - 1.0) / 0.25)**2), ['x'], [0.], [0.1], [(None, None)])
    assert res.values[0, 0] == pytest.approx(1.0, abs=1e-6)
    assert res.errors[0, 0] == pytest.approx(0.25, rel=1e-4)
    assert res.fval[0] < 1e-10 and res.is_valid[0]


def test_minimum_beyond_a_limit_stops_at_the_limit():
    # the unconstrained minimum sits at x = 2, the parameter may not pass 1: MINUIT's sine transform keeps it inside and the
    # fit ends at the limit (to the transform's guard); y is free and reaches its own minimum with the right error
    res, _ = _fit(lambda x: float((x[0] - 2.0)**2 / 0.04 + (x[1] + 0.5)**2 / 0.01), ['x', 'y'], [0.2, 0.], [0.1, 0.1],
                  [(-1.0, 1.0), (None, None)])
    assert res.values[0, 0] == pytest.approx(1.0, abs=2e-3) and res.values[0, 0] <= 1.0
    assert res.values[0, 1] == pytest.approx(-0.5, abs=1e-4)
    assert res.errors[0, 1] == pytest.approx(0.1, rel=5e-3)
    assert res.fval[0] == pytest.approx(1.0 / 0.04, rel=1e-3)
    assert 0.0 < res.errors[0, 0] <= 2.0          # a limited parameter's error never exceeds its range


def test_one_sided_limits_and_a_minimum_inside():
    res, _ = _fit(lambda x: float((x[0] - 0.7)**2 / 0.01 + (x[1] - 3.0)**2 / 0.09), ['x', 'y'], [2.0, 1.0], [0.1, 0.3],
                  [(0.0, None), (None, 10.0)])
    # (EDM < 2e-4 bounds the distance to the minimum by sqrt(2 EDM) = 0.02 sigma; the square-root transform is not linear)
    assert np.all(np.abs(res.values[0] - [0.7, 3.0]) < 0.02 * np.array([0.1, 0.3]))
    np.testing.assert_allclose(res.errors[0], [0.1, 0.3], rtol=1e-2)
    assert res.is_valid[0]


def test_rosenbrock_valley():
    def rosen(x):
        return float(100.0 * (x[1] - x[0]**2)**2 + (1.0 - x[0])**2)
    res, calls = _fit(rosen, ['x', 'y'], [-1.2, 1.0], [0.1, 0.1], [(None, None)] * 2)
    assert res.is_valid[0]
    assert res.edm[0] < EDM_GOAL and res.fval[0] < 10 * EDM_GOAL          # the function is within a few EDM of its minimum
    np.testing.assert_allclose(res.values[0], [1.0, 1.0], atol=0.05)
    # covariance of the valley at the minimum: Hessian [[802, -400], [-400, 200]]
    H = np.array([[802.0, -400.0], [-400.0, 200.0]])
    np.testing.assert_allclose(res.covariance[0], 2.0 * np.linalg.inv(H), rtol=0.1)
    assert calls['n'] < 1000


def test_call_limit_is_reported_not_iterated():
    def rosen(x):
        return float(100.0 * (x[1] - x[0]**2)**2 + (1.0 - x[0])**2)
    res, calls = _fit(rosen, ['x', 'y'], [-1.2, 1.0], [0.1, 0.1], [(None, None)] * 2, maxfcn=40)
    assert not res.is_valid[0]
    assert calls['n'] < 120         # a fit at its call limit is not run again (iminuit: `if fm.is_valid or fm.has_reached_call_limit: break`)


def test_iterate_runs_an_invalid_fit_again_from_its_last_state(monkeypatch):
    """iminuit's `migrad(iterate=5)`: an invalid minimum that is not at the call limit is minimised again from its last state
    (values, errors as steps, error matrix as first metric with dcovar = 0).  The first run of every fit is MARKED invalid
    here; the re-run starts at the minimum with the right metric, so it converges at once - a gradient, one step - and
    without another HESSE (dcovar stays below 0.05)."""
    A = np.array([[3.0, 0.4], [0.4, 2.0]]) * 50.0
    c = np.array([0.2, -0.1])
    runs = []

    class Marked(migrad._Fit):
        def run(self, *a, **k):
            yield from super().run(*a, **k)
            runs.append((self.seed_V is not None, self.nfcn))
            if self.seed_V is None:
                self.result['valid'] = False

    monkeypatch.setattr(migrad, '_Fit', Marked)
    res, calls = _fit(lambda x: float((x - c) @ A @ (x - c)), ['a', 'b'], [0., 0.], [0.1, 0.1], [(None, None)] * 2,
                      vectorised=False)
    assert [seeded for seeded, _ in runs] == [False, True]
    first, second = runs[0][1], runs[1][1]
    assert res.nfcn[0] == first + second == calls['n']
    assert second <= 1 + 2 * 2 * 3 + 4           # start value, <= 3 gradient cycles, a short line search: no HESSE
    assert res.is_valid[0]
    np.testing.assert_allclose(res.values[0], c, atol=1e-5)
    np.testing.assert_allclose(res.covariance[0], np.linalg.inv(A), rtol=5e-3)
    # iterate = 1: the first run's verdict stands
    runs.clear()
    res1, _ = _fit(lambda x: float((x - c) @ A @ (x - c)), ['a', 'b'], [0., 0.], [0.1, 0.1], [(None, None)] * 2,
                   vectorised=False, iterate=1)
    assert len(runs) == 1 and not res1.is_valid[0]
