"""GPU tests of the static quadratic form of chi2 (``vmx_set_quadratic_form``): chi2-only evaluations replace the
distortion product (reference vega/model.py:143-144) and the C^-1 product (vega/vega_interface.py:316) by ONE
half-triangle product with Q' = DM'^T S^T C^-1 S DM' per item.  The results must be those of the full chain - and of
the reference - to rounding, for every batch regime, with mocks, priors, rescaled covariances and changing data, and
clean enough for a minimiser's finite differences.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_problem, synth_joint_problem

pytestmark = pytest.mark.gpu

CHI2_RTOL = 1e-6
VARIED = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO', 'drp_QSO', 'bias_hcd',
          'beta_hcd', 'L0_hcd', 'bao_amp', 'sigmaNL_par', 'par_sigma_smooth']


def _took_quadratic_path(eng, theta):
    eng.set_profiling(True)
    eng.timings(reset=True)
    eng.eval(theta)
    t = eng.timings(reset=True)
    eng.set_profiling(False)
    return t['quadratic_form_product'][1] > 0 and t['distortion_product'][1] == 0 and t['invcov_product'][1] == 0


@pytest.mark.parametrize('batch', [1, 4, 9, 64, 256])
def test_quadratic_form_equals_the_full_chain_and_the_reference(batch):
    from vega_amd import VegaInterface, synthetic
    prob = synth_joint_problem()
    prob.priors = {'beta_LYA': np.array([1.5, 0.1]), 'ap': np.array([1.0, 0.05])}
    try:
        vega = VegaInterface(None, problem=prob, max_batch=batch)
    finally:
        prob.priors = {}
    eng = vega.engine
    assert eng.quadratic_form
    # the reference's fixture (8 walkers) tiled to the batch; the priors are added on the host for the comparison
    exp = np.load(GOLDEN / 'expected_joint_synth.npz')
    names = [str(n) for n in exp['param_names']]
    base = np.stack([eng.theta_from_params(dict(zip(names, row))) for row in exp['theta']])
    reps = -(-batch // 8)
    theta = np.tile(base, (reps, 1))[:batch]
    which = np.tile(np.arange(8), reps)[:batch]
    prior = ((theta[:, eng.low.slot['beta_LYA']] - 1.5) / 0.1)**2 + ((theta[:, eng.low.slot['ap']] - 1.0) / 0.05)**2
    quad = eng.eval(theta)[0]
    full = eng.eval(theta, want_model=True)[0]
    assert _took_quadratic_path(eng, theta)
    np.testing.assert_allclose(quad, full, rtol=1e-11)
    np.testing.assert_allclose(quad, exp['chi2'][which] + prior, rtol=CHI2_RTOL)
    np.testing.assert_array_equal(eng.eval(theta)[0], quad)             # bitwise repeatable
    # seeded walkers further from the reference point
    far = synthetic.walkers(eng.low.theta0, eng.names, batch, varied=VARIED, seed=5, scale=0.1)
    q, sq, _ = eng.eval(far)
    f, sf, _ = eng.eval(far, want_model=True)
    np.testing.assert_array_equal(sq, sf)
    ok = sq == 0
    np.testing.assert_allclose(q[ok], f[ok], rtol=1e-11)
    assert np.all(q[~ok] == 1e100)
    eng.set_quadratic_form(False)
    np.testing.assert_array_equal(eng.eval(theta)[0], full)             # switched off: the full chain, bit for bit
    assert eng.set_quadratic_form(True)
    vega.close()


@pytest.mark.parametrize('kind', ['q', 'factored'])
def test_quadratic_form_with_mocks_rescaled_covariance_and_new_data(kind):
    """Both shapes of the form (include/vegamx.h: vmx_set_quadratic_form_kind): the half-form Q' and || U r0 - F dx ||^2."""
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface, synthetic
    from vega_amd.montecarlo import create_mocks
    prob = synth_joint_problem()
    vega = VegaInterface(None, problem=prob, max_batch=32)
    eng = vega.engine
    eng.set_quadratic_form_kind(kind)
    fid = vega.compute_model()
    mocks = create_mocks(prob, fid, 5, seed=3)
    for name, pool in mocks.items():
        eng.set_mock_pool(name, pool)
    theta = synthetic.walkers(eng.low.theta0, eng.names, 20, varied=VARIED, seed=8, scale=0.01)
    index = np.array([0, 1, 2, 3, 4, -1, 2, 2, 0, 4] * 2, dtype=np.int32)
    eng.set_mock_index(index)
    quad = eng.eval(theta)[0]
    assert eng.last_form() == kind
    full = eng.eval(theta, want_model=True)[0]
    np.testing.assert_allclose(quad, full, rtol=1e-10)
    # chi2 ~ n_data against its own mocks, ~6e7 against the data: the same relative agreement for both
    assert quad[5] > 1e6 and quad[0] < 1e5
    for b in (0, 5, 7):
        pars = dict(zip(eng.names, theta[b]))
        data = None if index[b] < 0 else {n: mocks[n][index[b]] for n in mocks}
        assert quad[b] == pytest.approx(oc.chi2(prob, pars, data_override=data), rel=CHI2_RTOL)
    # a minimiser's finite differences: steps of 1e-3 sigma around a point near the minimum of a mock
    base = np.tile(eng.low.theta0, (9, 1))
    for j, n in enumerate(['ap', 'at', 'bias_eta_LYA', 'beta_LYA']):
        base[1 + 2 * j, eng.low.slot[n]] *= 1 + 1e-6
        base[2 + 2 * j, eng.low.slot[n]] *= 1 - 1e-6
    eng.set_mock_index(np.zeros(9, dtype=np.int32))
    dq = eng.eval(base)[0]
    df = eng.eval(base, want_model=True)[0]
    assert np.abs((dq - dq[0]) - (df - df[0])).max() < 1e-7      # differences of ~1e-3 reproduced to 1e-7 absolute
    eng.set_mock_index(None)
    # new data vectors through the reference's Monte-Carlo switch: the linear terms are refreshed
    for name, view in vega.data.items():
        view.masked_mc_mock = mocks[name][1]
    vega.monte_carlo = True
    assert vega.chi2() == pytest.approx(oc.chi2(prob, data_override={n: mocks[n][1] for n in mocks}), rel=CHI2_RTOL)
    vega.monte_carlo = False
    assert vega.chi2() == pytest.approx(oc.chi2(prob), rel=CHI2_RTOL)
    # rescaled inverse covariance (Monte-Carlo `scale`): the matrices are rebuilt (last: every rebuild factors the
    # covariances again on the host in the factored form)
    for name, item in prob.items.items():
        eng.set_invcov(name, item.chi2_matrix / 4.0)
    np.testing.assert_allclose(eng.eval(theta)[0], full_scaled := eng.eval(theta, want_model=True)[0], rtol=1e-10)
    np.testing.assert_allclose(full_scaled[5], full[5] / 4.0, rtol=1e-9)
    vega.close()


def test_quadratic_form_eligibility_and_golden_configs():
    """Identity distortion / identity covariance items, additive polynomial post-distortion broadband (in the form),
    and the configurations that must fall back to the full chain."""
    from vega_amd import VegaInterface
    # the reference's 4-correlation test config: identity matrices, additive post-distortion broadband on every item
    vega = VegaInterface(None, problem=load_problem('full4'), max_batch=4)
    exp = np.load(GOLDEN / 'expected_full4.npz')
    assert vega.engine.quadratic_form
    theta = vega.engine.low.theta0[None, :]
    assert _took_quadratic_path(vega.engine, theta)
    assert vega.chi2() == pytest.approx(float(exp['chi2']), rel=CHI2_RTOL)
    assert vega.engine.eval(theta, want_model=True)[0][0] == pytest.approx(float(exp['chi2']), rel=CHI2_RTOL)
    vega.close()
    # a global covariance: not eligible, the full chain runs
    prob = synth_joint_problem(with_global_cov=True)
    vega = VegaInterface(None, problem=prob, max_batch=2)
    assert not vega.engine.quadratic_form and not vega.engine.set_quadratic_form(True)
    assert not _took_quadratic_path(vega.engine, vega.engine.low.theta0[None, :])
    vega.close()


def test_single_walker_chain_tables_and_host_side_sum():
    """A single walker through the host entry: from the second call with the same shared (Arinyo / Gaussian-factor)
    parameters on, the P(k,mu) stage runs against the persistent level-2 tables, and the streaming products of the
    quadratic form leave per-block sums that the host adds up (no chi2 kernel).  Same chi2 as the batched paths, the full
    chain and the oracle - with priors, with a mock as data, after the shared parameters change and come back."""
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface, synthetic
    from vega_amd.montecarlo import create_mocks
    prob = synth_joint_problem()
    prob.priors = {'beta_LYA': np.array([1.5, 0.1]), 'ap': np.array([1.0, 0.05])}
    try:
        vega = VegaInterface(None, problem=prob, max_batch=32)
    finally:
        prob.priors = {}
    eng = vega.engine
    theta = synthetic.walkers(eng.low.theta0, eng.names, 6, varied=VARIED, seed=12, scale=0.02)
    prior = ((theta[:, eng.low.slot['beta_LYA']] - 1.5) / 0.1)**2 + ((theta[:, eng.low.slot['ap']] - 1.0) / 0.05)**2
    batch = eng.eval(np.tile(theta, (4, 1))[:20])[0][:6]                 # B = 20: table kernel + contraction epilogue
    full = eng.eval(theta, want_model=True)[0]                           # the full chain
    np.testing.assert_allclose(batch, full, rtol=1e-11)
    singles = np.array([[eng.eval(theta[i:i + 1])[0][0] for i in range(6)] for _ in range(3)])
    # (first pass: the per-walker loops - the shared parameters have to come twice; then the tables)
    np.testing.assert_allclose(singles[0], full, rtol=1e-11)
    np.testing.assert_allclose(singles[1], full, rtol=1e-11)
    np.testing.assert_array_equal(singles[2], singles[1])
    for i in (0, 3):
        assert singles[2][i] == pytest.approx(oc.chi2(vega.problem, dict(zip(eng.names, theta[i]))) + prior[i], rel=CHI2_RTOL)
    # other smoothing for a while, then back
    other = theta.copy()
    other[:, eng.low.slot['par_sigma_smooth']] *= 1.2
    ref_other = eng.eval(other, want_model=True)[0]
    for _ in range(3):
        got = np.array([eng.eval(other[i:i + 1])[0][0] for i in range(6)])
        np.testing.assert_allclose(got, ref_other, rtol=1e-11)
    assert np.abs(got / singles[2] - 1).max() > 1e-7
    for _ in range(3):
        got = np.array([eng.eval(theta[i:i + 1])[0][0] for i in range(6)])
        np.testing.assert_allclose(got, full, rtol=1e-11)
    # a mock as data for the single walker: the linear term's row and the constant follow the mock index
    mocks = create_mocks(prob, vega.compute_model(), 3, seed=4)
    for name, pool in mocks.items():
        eng.set_mock_pool(name, pool)
    eng.set_mock_index(np.array([2], dtype=np.int32))
    with_mock = [eng.eval(theta[1:2])[0][0] for _ in range(3)]
    eng.set_mock_index(np.array([2] * 20, dtype=np.int32))
    np.testing.assert_allclose(with_mock, eng.eval(np.tile(theta[1:2], (20, 1)))[0][0], rtol=1e-10)
    data = {n: mocks[n][2] for n in mocks}
    assert with_mock[2] == pytest.approx(oc.chi2(prob, dict(zip(eng.names, theta[1])), data_override=data) + prior[1], rel=CHI2_RTOL)
    eng.set_mock_index(None)
    # an out-of-range walker keeps its sentinel and its status on this path
    bad = theta[0:1].copy()
    bad[0, eng.low.slot['ap']] = 400.0
    for _ in range(3):
        c, s, _ = eng.eval(bad)
        assert c[0] == 1e100 and s[0] != 0
    np.testing.assert_allclose(eng.eval(theta[0:1])[0][0], full[0], rtol=1e-11)
    vega.close()
