"""Size-independent properties at the full size of the bench workload (BASELINE configs[2]: joint Lya x Lya +
QSO x Lya, dense 2500^2 + 5000^2 distortion matrices, 1590^2 + 3180^2 inverse covariances, B = 256), where the CPU
oracle is too slow to check every walker: linearity of the model in the BAO amplitude and of the stand-alone
product in its operand, the model-as-data round trip (chi2 = 0), additivity of chi2 over priors, bitwise
repeatability, and walker-order independence.
"""
import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def vega():
    from vega_amd import VegaInterface, synthetic
    from vega_amd.setup import build_problem
    prob = build_problem('configs/joint/main.ini', search_dirs=[GOLDEN])
    for item in prob.items.values():
        item.distortion = synthetic.distortion_matrix(item.model_grid.rp, item.model_grid.rt)
        item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt))
    v = VegaInterface(None, problem=prob, max_batch=256)
    yield v
    v.close()


def _walkers(vega, n, seed):
    from vega_amd import synthetic
    eng = vega.engine
    varied = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO', 'drp_QSO',
              'bias_hcd', 'beta_hcd', 'L0_hcd']
    return synthetic.walkers(eng.low.theta0, eng.names, n, varied=varied, seed=seed)


def test_model_is_affine_in_the_bao_amplitude(vega):
    """model = bao_amp * peak + smooth (reference model.py:186): three amplitudes per walker lie on a line."""
    eng = vega.engine
    theta = _walkers(vega, 64, seed=1)
    slot = eng.low.slot['bao_amp']
    models = []
    for amp in (0.0, 1.0, 2.5):
        t = theta.copy()
        t[:, slot] = amp
        models.append(eng.eval(t, want_model=True)[2])
    m0, m1, m2 = models
    scale = np.abs(m1).max()
    assert np.abs((m2 - m0) - 2.5 * (m1 - m0)).max() <= 1e-12 * scale


def test_model_as_data_round_trip(vega):
    """Swap the data for the model of walker i (through the per-walker mock pool): chi2 of walker i against its own
    model is zero to rounding, and positive against any other walker's."""
    eng = vega.engine
    theta = _walkers(vega, 256, seed=2)
    _, status, model = eng.eval(theta, want_model=True)
    assert not status.any()
    for name, item in vega.problem.items.items():
        sl = eng.model_slices[name]
        eng.set_mock_pool(name, np.ascontiguousarray(model[:, sl][:, item.model_mask]))
    eng.set_mock_index(np.arange(256))
    own = eng.eval(theta, want_model=True)[0]               # the full chain: residuals that vanish identically
    own_quad = eng.eval(theta)[0]                           # chi2 only: the static quadratic form around the reference point
    eng.set_mock_index(np.roll(np.arange(256), 1))
    other = eng.eval(theta)[0]
    eng.set_mock_index(None)
    typical = np.median(other)
    assert np.all(np.abs(own) <= 1e-20 * typical + 1e-18)
    # three terms of the size of chi2(walker's model against the reference point's model) cancel: rounding of that size
    assert np.all(np.abs(own_quad) <= 1e-12 * typical + 1e-9)
    assert np.all(other > 0) and typical > 1.0


def test_repeatable_and_order_independent(vega):
    eng = vega.engine
    theta = _walkers(vega, 256, seed=3)
    a = eng.eval(theta)[0]
    b = eng.eval(theta)[0]
    np.testing.assert_array_equal(a, b)                     # fixed-order reductions: bitwise repeatable
    perm = np.random.default_rng(0).permutation(256)
    c = eng.eval(theta[perm])[0]
    np.testing.assert_allclose(c, a[perm], rtol=1e-12)       # a walker's result does not depend on its neighbours


def test_stand_alone_product_is_linear(vega):
    """y = A x through the MFMA and the streaming kernels at 5000^2: (a x1 + b x2) -> a y1 + b y2; B = 1 equals a
    row of the batched result."""
    eng = vega.engine
    rng = np.random.default_rng(4)
    A = rng.standard_normal((5000, 5000))
    X = rng.standard_normal((64, 5000))
    Y = eng.matmul_host(A, X)
    np.testing.assert_allclose(Y, X @ A.T, rtol=0, atol=1e-11 * np.abs(Y).max())
    combo = 0.3 * X[:32] - 1.7 * X[32:]
    np.testing.assert_allclose(eng.matmul_host(A, combo), 0.3 * Y[:32] - 1.7 * Y[32:], rtol=0, atol=1e-11 * np.abs(Y).max())
    np.testing.assert_allclose(eng.matmul_host(A, X[5:6])[0], Y[5], rtol=0, atol=1e-11 * np.abs(Y).max())


def test_priors_add_to_chi2(vega):
    """chi2 with Gaussian priors = chi2 without + sum ((theta - mean) / sigma)^2 (reference vega_interface.py:423-446)."""
    import copy
    from vega_amd import VegaInterface
    theta = _walkers(vega, 32, seed=5)
    base = vega.engine.eval(theta)[0]
    prob = copy.copy(vega.problem)
    prob.priors = {'beta_LYA': np.array([1.6, 0.2]), 'ap': np.array([1.0, 0.03])}
    with_priors = VegaInterface(None, problem=prob, max_batch=32)
    got = with_priors.engine.eval(theta)[0]
    slot = with_priors.engine.low.slot
    extra = ((theta[:, slot['beta_LYA']] - 1.6) / 0.2)**2 + ((theta[:, slot['ap']] - 1.0) / 0.03)**2
    np.testing.assert_allclose(got, base + extra, rtol=1e-12)
    with_priors.close()
