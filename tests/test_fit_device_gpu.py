"""The device-resident MIGRAD fits (include/vegamx.h: vmx_fit_migrad - state machines of vega_amd/csrc/vmx_migrad.h advanced by the
fit kernels, parameter rows / chi2 / states in HBM) against the NumPy lock-step driver of vega_amd/migrad.py through the host
entry, on real engines: the same number of function calls fit by fit (a decision taken differently anywhere changes it), the
same minima, errors and flags; walkers that bring their mock rows per call against the host-stated rows; the scalar fit, fixed
parameters, a fit whose model cannot be evaluated, a global covariance (full chain), and what the driver reports about itself.
Reference semantics: vega/minimizer.py:66-97 (bias pre-fit, then the full MIGRAD), vega/analysis.py:224-308 (one fit per mock)."""
import numpy as np
import pytest

from conftest import GOLDEN, load_problem, synth_joint_problem

pytestmark = pytest.mark.gpu


def _sample(prob, names, limits, errors):
    prob.sample_params = {'limits': dict(zip(names, limits)), 'values': {n: prob.params[n] for n in names},
                          'errors': dict(zip(names, errors)), 'fix': {n: False for n in names}}


_FID = {}


def _run_mc(vega, n_mocks, seed, driver, **kw):
    from vega_amd.montecarlo import MonteCarlo
    vega.freeze_metals()
    mc = MonteCarlo(vega)
    mc.driver = 'python' if driver == 'python' else 'device'
    mc.stream_mocks = driver == 'streamed'          # (the mocks made while their fits run: include/vegamx.h vmx_mock_stream)
    vega.analysis = mc
    if id(vega) not in _FID:        # (one fiducial model for both drivers: a single walker's path through the engine depends on
        _FID[id(vega)] = vega.compute_model()       # what the tables hold from earlier calls - equal to 1e-16, not bitwise)
    res = mc.run_monte_carlo(_FID[id(vega)], num_mocks=n_mocks, seed=seed, **kw)
    return mc, res


def _assert_same_fits(a, b, rtol_err=1e-5, flips=0, rtol_fval=1e-11, max_pull=1e-6):
    """Fit by fit the same calls and the same results.  `flips`: how many fits may take a decision differently - the two drivers
    hand the engine differently composed batches, chi2 of a point then differs in its last bits (another cut of the sums), and a
    long fit on an ill-conditioned problem (hundreds of calls) can cross one of Minuit's thresholds on the other side; such a fit
    must still end at the same minimum within Minuit's own tolerance."""
    same = a.nfcn == b.nfcn
    assert (~same).sum() <= flips, (a.nfcn, b.nfcn)
    np.testing.assert_array_equal(a.n_iter[same], b.n_iter[same])
    np.testing.assert_array_equal(a.is_valid, b.is_valid)
    np.testing.assert_array_equal(a.hesse_failed, b.hesse_failed)
    fin = np.isfinite(a.fval)
    np.testing.assert_array_equal(fin, np.isfinite(b.fval))
    scale = np.where(a.errors > 0, a.errors, 1.)
    pull = np.abs((b.values - a.values) / scale)
    ok = fin & same
    np.testing.assert_allclose(b.fval[ok], a.fval[ok], rtol=rtol_fval)
    assert pull[ok].max() < max_pull          # of the parameter's own error
    np.testing.assert_allclose(b.errors[ok], a.errors[ok], rtol=rtol_err)
    np.testing.assert_allclose(b.covariance[ok], a.covariance[ok], rtol=1e-4, atol=1e-6 * np.abs(a.covariance[ok]).max())
    other = fin & ~same
    if other.any():
        assert pull[other].max() < 0.1 and np.abs(b.fval[other] - a.fval[other]).max() < 2e-3       # (EDM goal 2e-4, up = 1)
        np.testing.assert_allclose(b.errors[other], a.errors[other], rtol=0.15)


def test_device_fits_equal_the_numpy_driver_on_monte_carlo_mocks():
    """48 mocks of the joint auto + cross problem, six parameters with the bias pre-fit: both drivers, fit by fit."""
    from vega_amd import VegaInterface
    prob = synth_joint_problem()
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd']
    _sample(prob, names, [(0.5, 1.5), (0.5, 1.5), (-2., 0.), (0., 5.), (0., 1.), (-0.5, 0.)], [0.01, 0.01, 0.01, 0.1, 0.1, 0.01])
    vega = VegaInterface(None, problem=prob, max_batch=256)
    mc_a, a = _run_mc(vega, 48, 3, 'python')
    mc_b, b = _run_mc(vega, 48, 3, 'device')
    _assert_same_fits(a, b)
    assert b.is_valid.all()
    for name in prob.items:
        np.testing.assert_array_equal(mc_a.mc_mocks[name], mc_b.mc_mocks[name])
    # what the driver says about itself: every evaluation was one of a fit's, the fits ran at their own pace (far fewer rounds
    # than the lock-step driver's stages), nothing was left unfinished, and the GPU waited for the host for a small part of the run
    st = b.driver_stats
    assert st['evaluations'] == int(b.nfcn.sum()) and st['fits_unfinished'] == 0
    assert st['rounds'] < 100 and st['engine_calls'] >= st['rounds']
    assert sum(st['evaluations_by_batch'].values()) == st['evaluations']
    assert st['gpu_idle_seconds_between_rounds'] < 0.25 * st['seconds_rounds']
    assert a.driver_stats is None
    # a rescaled covariance
    mc_c, c = _run_mc(vega, 20, 5, 'python', scale=1.7)
    mc_d, d = _run_mc(vega, 20, 5, 'device', scale=1.7)
    _assert_same_fits(c, d)
    vega.close()


def test_mocks_made_while_their_fits_run():
    """`run_monte_carlo` with the mock stream (the default of the device driver): the normal draws come from a host thread in the
    reference's order (vega/data.py:748-757), the mocks are formed on the device wave by wave and join the fits on a fixed
    schedule.  Same mocks as `create_mocks` to rounding, same fits, the same results again (nothing depends on how fast the
    draws arrive), waves that do not divide the number of mocks, a rescaled covariance."""
    from vega_amd import VegaInterface
    prob = synth_joint_problem()
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd']
    _sample(prob, names, [(0.5, 1.5), (0.5, 1.5), (-2., 0.), (0., 5.), (0., 1.), (-0.5, 0.)], [0.01, 0.01, 0.01, 0.1, 0.1, 0.01])
    vega = VegaInterface(None, problem=prob, max_batch=256)
    n = 150                     # (two full waves of 64 and one of 22)
    mc_a, a = _run_mc(vega, n, 3, 'python')
    mc_b, b = _run_mc(vega, n, 3, 'streamed')
    for name in prob.items:
        assert mc_b.mc_mocks[name].shape == mc_a.mc_mocks[name].shape
        scale = np.abs(mc_a.mc_mocks[name]).max()
        np.testing.assert_allclose(mc_b.mc_mocks[name], mc_a.mc_mocks[name], rtol=0, atol=1e-13 * scale)
    _assert_same_fits(a, b, flips=1)
    st = b.driver_stats
    assert st['evaluations'] == int(b.nfcn.sum()) and st['fits_unfinished'] == 0 and st['rounds'] < 110
    assert st['seconds_waiting_for_draws'] < 0.5 * st['seconds']
    mc_c, c = _run_mc(vega, n, 3, 'streamed')
    for name in prob.items:
        np.testing.assert_array_equal(mc_c.mc_mocks[name], mc_b.mc_mocks[name])
    np.testing.assert_array_equal(c.nfcn, b.nfcn)
    np.testing.assert_array_equal(c.values, b.values)
    np.testing.assert_array_equal(c.fval, b.fval)
    # the mocks read back are the data the fits saw: chi2 of a best fit against its mock through the host entry
    eng = vega.engine
    for f in (0, 70, 149):
        for name in prob.items:
            eng.set_data(name, mc_b.mc_mocks[name][f])
        assert vega.chi2(dict(zip(b.names, b.values[f]))) == pytest.approx(b.fval[f], rel=1e-10)
    for name, item in prob.items.items():
        eng.set_data(name, item.masked_data_vec)
    _, d = _run_mc(vega, 40, 8, 'python', scale=2.5)
    _, e = _run_mc(vega, 40, 8, 'streamed', scale=2.5)
    _assert_same_fits(d, e, flips=1)
    vega.close()


def test_rows_per_call_equal_rows_stated_from_the_host():
    """vmx_eval_device_mocks (walkers bring their pool rows, device memory) against vmx_set_mock_index + the host entry, full
    chain and quadratic form, with chunks on both lanes."""
    import torch
    from vega_amd import VegaInterface
    from vega_amd.montecarlo import create_mocks
    prob = synth_joint_problem()
    vega = VegaInterface(None, problem=prob, max_batch=256)
    eng = vega.engine
    mocks = create_mocks(prob, vega.compute_model(), 7, seed=2)
    for name, pool in mocks.items():
        eng.set_mock_pool(name, pool)
    rng = np.random.default_rng(5)
    B = 200
    theta = np.tile(eng.low.theta0, (B, 1))
    for n in ('ap', 'at', 'beta_LYA'):
        theta[:, eng.low.slot[n]] *= 1 + 0.01 * rng.normal(size=B)
    rows = rng.integers(-1, 7, size=B).astype(np.int32)
    eng.set_mock_index(rows)
    want = eng.eval(theta)[0]
    eng.set_mock_index(None)
    dev = torch.device('cuda', 0)
    d_theta = torch.from_numpy(theta).to(dev)
    d_rows = torch.from_numpy(rows).to(dev)
    d_chi2 = torch.zeros(B, dtype=torch.float64, device=dev)
    for lanes in (1, 2):
        eng.set_lanes(lanes)
        for lo in (0, 100):          # (two calls of 100 walkers: with two lanes one on each)
            eng.eval_device_mocks(d_theta[lo:lo + 100].data_ptr(), 100, d_chi2[lo:lo + 100].data_ptr(), d_rows[lo:lo + 100].data_ptr())
        eng.sync()
        np.testing.assert_allclose(d_chi2.cpu().numpy(), want, rtol=1e-13)      # (100 + 100 walkers against 200: another cut of the sums)
        d_chi2.zero_()
    eng.set_lanes(1)
    # small batches take the same eager chain (no captured graph per batch size)
    for nb in (1, 5, 9, 33):
        eng.eval_device_mocks(d_theta.data_ptr(), nb, d_chi2.data_ptr(), d_rows.data_ptr())
        eng.sync()
        np.testing.assert_allclose(d_chi2[:nb].cpu().numpy(), want[:nb], rtol=1e-12)
    vega.close()


def test_the_scalar_fit_fixed_parameters_and_a_fit_that_cannot_run():
    """`vega.minimize()` (one fit, the data), `fix`, and a start the model cannot be evaluated at, on both drivers."""
    from vega_amd import VegaInterface
    from vega_amd.montecarlo import MonteCarlo
    vega = VegaInterface(None, problem=load_problem('full4'), max_batch=64)
    res = {}
    for driver in ('python', 'device'):
        mc = MonteCarlo(vega)
        mc.driver = driver
        vega.freeze_metals()
        fitter = mc.minimizer(vega.sample_params)
        res[driver] = fitter.minimize(n_fits=1, fixed=mc._fixed)
    _assert_same_fits(res['python'], res['device'])
    assert res['device'].nfcn[0] == 53          # the reference's pinned fit (tests/test_fits_gpu.py holds its value)
    vega.close()

    prob = synth_joint_problem()
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA']
    _sample(prob, names, [(0.5, 1.5), (0.5, 1.5), (-2., 0.), (0., 5.)], [0.01, 0.01, 0.01, 0.1])
    vega = VegaInterface(None, problem=prob, max_batch=64)
    out = {}
    for driver in ('python', 'device'):
        mc = MonteCarlo(vega)
        mc.driver = driver
        sample = {k: dict(v) for k, v in prob.sample_params.items()}
        sample['fix']['at'] = True
        fitter = mc.minimizer(sample)
        start = np.tile([sample['values'][n] for n in names], (3, 1))
        start[1, 1] = np.nan        # the fixed `at` is not a number for this fit: the engine returns its sentinel, 1e100
        out[driver] = fitter.minimize(n_fits=3, start=start, fixed=mc._fixed)
    a, b = out['python'], out['device']
    _assert_same_fits(a, b)
    assert np.all(b.values[[0, 2], 1] == start[[0, 2], 1]) and np.all(b.errors[:, 1] == 0.)
    assert not np.isfinite(b.fval[1]) and not b.is_valid[1] and b.hesse_failed[1] and b.is_valid[[0, 2]].all()
    vega.close()


def test_device_fits_with_a_global_covariance_and_with_metals():
    """The full chain as the fits' objective (a global covariance: no quadratic form, one lane) and a problem with metal terms."""
    from vega_amd import VegaInterface
    prob = synth_joint_problem(with_global_cov=True)
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA']
    _sample(prob, names, [(0.5, 1.5), (0.5, 1.5), (-2., 0.), (0., 5.)], [0.01, 0.01, 0.01, 0.1])
    vega = VegaInterface(None, problem=prob, max_batch=64)
    _, a = _run_mc(vega, 6, 4, 'python')
    _, b = _run_mc(vega, 6, 4, 'device')
    _assert_same_fits(a, b)
    vega.close()

    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    prob = build_problem('configs/auto_metals/main.ini', search_dirs=[GOLDEN])
    for item in prob.items.values():
        item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt))
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'bias_eta_SiII(1260)']
    _sample(prob, names, [(0.5, 1.5), (0.5, 1.5), (-2., 0.), (0., 5.), (-0.1, 0.)], [0.01, 0.01, 0.01, 0.1, 0.001])
    vega = VegaInterface(None, problem=prob, max_batch=64)
    _, a = _run_mc(vega, 5, 9, 'python')
    _, b = _run_mc(vega, 5, 9, 'device')
    # (fits of 200 - 430 calls, the metal bias barely constrained: last-bit differences of chi2 grow to 1e-10 of the minimum's value
    # over such a fit without changing a decision - seen between boxes, whose host BLAS the NumPy driver's arithmetic follows)
    _assert_same_fits(a, b, rtol_err=1e-4, flips=1, rtol_fval=1e-8, max_pull=1e-4)
    vega.close()


def test_seventeen_free_parameters_take_the_largest_state_record():
    """More than 16 free parameters: the fit kernels' 32-parameter instantiation (k_fit_advance<32, 1>, k_fit_emit<32>: a 22 KB
    state record per fit, gradients of 34 points, HESSE's 136 off-diagonal points) against the NumPy driver: five physical
    parameters and the twelve coefficients of the additive post-distortion broadband (chi2 is exactly quadratic in those)."""
    import sys
    from conftest import REPO
    sys.path.insert(0, str(REPO))
    from bench import build_problem
    from vega_amd import VegaInterface
    prob = build_problem('joint_metals')
    bb = sorted(n for n in prob.params if n.startswith('BB-') and 'add post' in n)
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO'] + bb
    assert len(names) == 17
    _sample(prob, names, [(0.5, 1.5), (0.5, 1.5), (-2., 0.), (0., 5.), (0., 1.)] + [(-1., 1.)] * 12, [0.01, 0.01, 0.01, 0.1, 0.1] + [1e-4] * 12)
    vega = VegaInterface(None, problem=prob, max_batch=256)
    _, a = _run_mc(vega, 3, 21, 'python')
    mc, b = _run_mc(vega, 3, 21, 'device')
    # (fits of 960 - 1340 calls each; call for call the same on the boxes seen so far)
    _assert_same_fits(a, b, rtol_err=1e-3, flips=1, rtol_fval=1e-8, max_pull=1e-3)
    assert a.is_valid.all() and mc.driver_stats['fits_unfinished'] == 0
    vega.close()


def test_refused_fit_requests_leave_the_engine_as_it_was():
    """include/vegamx.h: every argument of vmx_fit_migrad is checked before anything runs."""
    from vega_amd import VegaInterface
    prob = synth_joint_problem()
    vega = VegaInterface(None, problem=prob, max_batch=64)
    eng = vega.engine
    before = vega.chi2()
    col = {n: eng.low.slot[n] for n in ('ap', 'at', 'beta_LYA')}
    theta = np.tile(eng.low.theta0, (4, 1))
    good = dict(free=[col['ap'], col['at']], limits=[(0.5, 1.5), (0.5, 1.5)], errors=[0.01, 0.01])
    base = dict(iterate=5, maxfcn=100000, up=1.0, tol=0.1)

    def plan(**stage):
        return dict(base, stages=[dict(good, **stage)])
    refused = [
        (plan(free=[col['ap'], col['ap']]), 'twice'),
        (plan(free=[col['ap'], eng.n_params]), 'column'),
        (plan(limits=[(1.5, 0.5), (0.5, 1.5)]), 'limits'),
        (plan(errors=[0.0, 0.01]), 'step'),
        (dict(base, iterate=0, stages=[good]), 'iterate'),
        (dict(base, up=-1.0, stages=[good]), 'up'),
    ]
    for p, word in refused:
        with pytest.raises(Exception, match=word):
            eng.fit_migrad(p, theta)
    with pytest.raises(Exception, match='const_hint'):
        eng.fit_migrad(plan(), theta, const_hint=7)
    with pytest.raises(Exception, match='pool'):
        eng.fit_migrad(plan(), theta, mock_rows=np.array([0, 0, 0, 3], dtype=np.int32))      # (no mock pool installed)
    assert vega.chi2() == before
    outs, stats = eng.fit_migrad(plan(), theta)                 # ... and the same request, valid, runs
    assert stats['fits_unfinished'] == 0 and (outs[0]['flags'] & 1).all()
    assert vega.chi2() == before
    vega.close()
