"""CPU tests (no GPU): the oracle against the round-2 reference fixtures - BASELINE configs[1] as stated, the
marginalisation coefficients, ``Model.compute`` with caller-supplied spectra, chi2 with a global covariance - and the
host side of the Monte-Carlo driver (global-covariance mocks in the reference's draw order, scale resolution, the
unmasked-Cholesky mode, result tables).
"""
import numpy as np
import pytest

from conftest import GOLDEN, config1_problem, marginalization_problem, MARGINALIZATION_CASES, synth_joint_problem

CHI2_RTOL = 1e-6


def _pars(exp, prefix=''):
    return [{str(n): float(v) for n, v in zip(exp[prefix + 'param_names'], row)} for row in exp[prefix + 'theta']]


def _assert_xi_elementwise(got, ref, mask, what):
    """north_star: <= 1e-8 relative on xi - element by element on the bins the fit uses (with a floor of 1e-12 of the
    vector's scale for the zero crossings), and to the vector's scale on all bins."""
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-8 * scale, what
    np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-8, atol=1e-12 * scale, err_msg=what)


def test_oracle_on_config1_as_stated():
    """configs[1]: auto only, ell <= 4, dense 2500^2 distortion matrix, one evaluation at a time."""
    from oracle import vega_cpu as oc
    prob = config1_problem()
    item = prob.items['lyalya_lyalya']
    assert item.core.xi.ell_max == 4 and item.distortion.shape == (2500, 2500)
    exp = np.load(GOLDEN / 'expected_config1.npz')
    assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-12)
    assert oc.log_lik(prob) == pytest.approx(float(exp['fid/log_lik']), rel=1e-12)
    _assert_xi_elementwise(oc.compute_model(prob)['lyalya_lyalya'], exp['fid/model/lyalya_lyalya'], item.model_mask, 'fid')
    for i, pars in enumerate(_pars(exp)[:2]):
        assert oc.chi2(prob, pars) == pytest.approx(float(exp['chi2'][i]), rel=1e-12)
        _assert_xi_elementwise(oc.compute_model(prob, pars)['lyalya_lyalya'], exp[f'walker{i}/model/lyalya_lyalya'],
                               item.model_mask, f'walker {i}')


@pytest.mark.parametrize('mode', ['cov', 'infit'])
def test_oracle_marginalisation_coefficients(tmp_path, mode):
    from oracle import vega_cpu as oc
    exp = np.load(GOLDEN / 'expected_marg_coeff.npz')
    prob = marginalization_problem(tmp_path, MARGINALIZATION_CASES['rtmax'], in_fit=mode == 'infit')
    chi2, coeff = oc.chi2(prob, return_marg_coeff=True)
    assert chi2 == pytest.approx(float(exp[f'{mode}/fid/chi2']), rel=1e-10)
    ref = exp[f'{mode}/fid/coeff']
    np.testing.assert_allclose(coeff['lyalya_lyalya'], ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    pars = _pars(exp, f'{mode}/')[0]
    chi2, coeff = oc.chi2(prob, pars, return_marg_coeff=True)
    assert chi2 == pytest.approx(float(exp[f'{mode}/walker0/chi2']), rel=1e-10)
    ref = exp[f'{mode}/walker0/coeff']
    np.testing.assert_allclose(coeff['lyalya_lyalya'], ref, rtol=1e-9, atol=1e-9 * np.abs(ref).max())
    assert oc.chi2(prob, {'ap': 1e3}, return_marg_coeff=True) == (1e100, None)
    # compute_model(marg_coeff=...) adds templates . coefficients (reference vega_interface.py:243-246)
    item = prob.items['lyalya_lyalya']
    with_t = exp[f'{mode}/fid/model_plain'] + item.marg_templates.dot(exp[f'{mode}/fid/coeff'])
    np.testing.assert_allclose(with_t, exp[f'{mode}/fid/model_with_templates'], rtol=0,
                               atol=1e-13 * np.abs(with_t).max())


def test_oracle_model_compute_with_caller_spectra():
    from conftest import load_problem
    from oracle import vega_cpu as oc
    exp = np.load(GOLDEN / 'expected_model_compute.npz')
    prob = load_problem('joint_metals')
    walker = _pars(exp)[0]
    for name, item in prob.items.items():
        oc.reset_metal_cache(prob)
        got = oc.model_compute(prob, item, oc.local_params(prob))
        _assert_xi_elementwise(got, exp[f'fiducial_spectra/{name}'], item.model_mask, name)
        oc.reset_metal_cache(prob)
        got = oc.model_compute(prob, item, oc.local_params(prob, walker), pk_full=exp['pk_full'],
                               pk_smooth=exp['pk_smooth'])
        _assert_xi_elementwise(got, exp[f'own_spectra/{name}'], item.model_mask, name + ' own spectra')


def test_global_covariance_file_chi2_and_mocks(tmp_path):
    """`global-cov-file` ingestion, chi2 / log-likelihood with it, and the reference's global mocks for seed 7
    (vega/analysis.py:164-222): one randn over all correlations through the Cholesky factor of the masked global
    covariance - not item by item."""
    from oracle import vega_cpu as oc
    from vega_amd.montecarlo import create_global_mocks, create_mocks, split_global
    exp = np.load(GOLDEN / 'expected_global_mc.npz')
    prob = synth_joint_problem(with_global_cov=True, tmp_path=tmp_path)
    assert prob.global_cov.shape == (7500, 7500)
    assert oc.chi2(prob) == pytest.approx(float(exp['data/chi2']), rel=1e-10)
    assert oc.log_lik(prob) == pytest.approx(float(exp['data/log_lik']), rel=1e-10)
    fid = oc.compute_model(prob)
    mocks = create_global_mocks(prob, fid, 2, seed=7)
    for i in range(2):
        ref = exp[f'mock{i}/global']
        np.testing.assert_allclose(mocks[i], ref, rtol=0, atol=1e-12 * np.abs(ref).max())
        assert oc.chi2(prob, data_override=mocks[i]) == pytest.approx(float(exp[f'mock{i}/chi2_fid']), rel=1e-9)
    parts = split_global(prob, mocks)
    assert [p.shape for p in parts.values()] == [(2, 1590), (2, 3180)]
    np.testing.assert_array_equal(np.concatenate(list(parts.values()), axis=1), mocks)
    with pytest.raises(ValueError):
        create_mocks(prob, fid, 1, seed=7)          # per-item draws would silently drop the cross-covariance
    # a rescaled covariance: the factor is cached with the first scale it is built with, as the reference's is
    prob._global.pop('cholesky')
    scaled = create_global_mocks(prob, fid, 1, seed=7, scale=0.25)
    ref = exp['scaled/mock0/global']
    np.testing.assert_allclose(scaled[0], ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    # forecast: the fiducial without noise
    np.testing.assert_array_equal(create_global_mocks(prob, fid, 1, forecast=True)[0],
                                  np.concatenate([fid[n] for n in prob.items])[prob.global_masks()['data_mask']])


def test_mock_scale_resolution_and_unmasked_cholesky():
    """reference vega/analysis.py:147-156 (scale: None -> cov_rescale, number, dict with default 1) and
    vega/data.py:727-757 (`cholesky-masked-cov = False`: randn over the full data size, the mask applied afterwards)."""
    from conftest import load_problem
    from vega_amd import synthetic
    from vega_amd.montecarlo import create_mocks, item_scales
    prob = load_problem('joint')
    names = list(prob.items)
    for item in prob.items.values():
        item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt))
    assert item_scales(prob, None) == {n: 1. for n in names}
    prob.items[names[0]].cov_rescale = 0.5
    assert item_scales(prob, None) == {names[0]: 0.5, names[1]: 1.}
    assert item_scales(prob, 2) == {n: 2. for n in names}
    assert item_scales(prob, {names[1]: 3.}) == {names[0]: 1., names[1]: 3.}
    fid = {n: np.zeros(it.data_vec.size) for n, it in prob.items.items()}
    try:
        prob.items[names[1]].cholesky_masked_cov = False
        got = create_mocks(prob, fid, 2, seed=5, scale={names[1]: 3.})
        np.random.seed(5)
        want = {n: [] for n in names}
        a, b = prob.items[names[0]], prob.items[names[1]]
        la = np.linalg.cholesky(1. * a.cov[:, a.data_mask][a.data_mask, :])
        lb = np.linalg.cholesky(3. * b.cov)
        for _ in range(2):
            want[names[0]].append(la.dot(np.random.randn(a.data_size)))
            want[names[1]].append(lb.dot(np.random.randn(b.data_vec.size))[b.data_mask])
        for n in names:
            np.testing.assert_allclose(got[n], np.array(want[n]), rtol=0, atol=1e-15)
        prob.items[names[1]].cov = None
        with pytest.raises(ValueError):
            create_mocks(prob, fid, 1)              # no identity stand-in for a missing covariance
    finally:
        prob.items[names[0]].cov_rescale = None
        prob.items[names[1]].cholesky_masked_cov = True
        for item in prob.items.values():
            item.set_covariance(None)
            item.__dict__.pop('_cholesky', None)


def test_monte_carlo_tables_follow_the_reference_layout(tmp_path):
    """reference vega/output.py:442-520 + vega/data.py:749-753 + vega/analysis.py:279-297: mocks on the full data grid
    with NaN outside the mask; a failed fit has no Bestfit row and chisq = NaN."""
    from types import SimpleNamespace
    from conftest import load_problem
    from vega_amd import fitslite
    from vega_amd.output import write_monte_carlo
    prob = load_problem('joint')
    rng = np.random.default_rng(1)
    mocks = {n: rng.standard_normal((3, it.data_size)) for n, it in prob.items.items()}
    analysis = SimpleNamespace(
        vega=SimpleNamespace(problem=prob), mc_mocks=mocks,
        mc_bestfits={'ap': rng.standard_normal((2, 2)), 'at': rng.standard_normal((2, 2))},
        mc_covariances=list(rng.standard_normal((2, 2, 2))), mc_chisq=[1.0, np.nan, 3.0],
        mc_valid_minima=[True, False, True], mc_valid_hesse=[True, False, True], mc_failed_mask=[False, True, False])
    path = write_monte_carlo(analysis, tmp_path, cpu_id=3, overwrite=True)
    assert path.name == 'monte_carlo_3.fits'
    hdul = fitslite.open(path)
    by_name = {h.header['EXTNAME']: h for h in hdul[1:]}
    best = by_name['BESTFIT'] if 'BESTFIT' in by_name else by_name['Bestfit']
    assert np.asarray(best.data['values']).shape == (2, 2)          # [parameters][successful fits]
    np.testing.assert_allclose(np.asarray(best.data['values'])[0], analysis.mc_bestfits['ap'][:, 0])
    info = by_name['FITINFO'] if 'FITINFO' in by_name else by_name['FitInfo']
    assert np.isnan(np.asarray(info.data['chisq'])[1])
    table = by_name['MOCKS'] if 'MOCKS' in by_name else by_name['Mocks']
    for n, it in prob.items.items():
        col = np.asarray(table.data[n])
        assert col.shape == (3, it.data_vec.size)
        np.testing.assert_array_equal(col[:, it.data_mask], mocks[n])
        assert np.isnan(col[:, ~it.data_mask]).all()


def test_minimizer_reports_a_sentinel_gradient_as_a_failed_fit():
    """A finite-difference stencil that hits the 1e100 sentinel must not turn into 'gradient zero, converged'."""
    from vega_amd.minimizer import BatchedMinimizer

    def evaluate(x, fits):
        out = ((x - 0.3)**2).sum(axis=1) * 50
        out[x[:, 0] > 0.5] = 1e100         # fit 1 starts right below the wall: its first stencil crosses it
        return out

    m = BatchedMinimizer(evaluate, ['a', 'b'], [0.0, 0.0], [0.1, 0.1], [(None, None), (None, None)])
    res = m.minimize(n_fits=2, start=[[0.0, 0.0], [0.49999, 0.0]], prefit_bias=False)
    assert res.is_valid[0] and abs(res.values[0] - 0.3).max() < 1e-3
    assert not res.is_valid[1] and not np.isfinite(res.edm[1])


def test_oracle_on_the_xi_stage_options_of_round_2(tmp_path):
    """`rescale-coords-systematics`, `old_growth_func`, `fht_lowring = False` (reference correlation_func.py:470-475,
    :681-684, :75-80, :405-444; pktoxi.py:42,53) against the unmodified reference."""
    from conftest import options2_problem
    from oracle import vega_cpu as oc
    exp = np.load(GOLDEN / 'expected_options2.npz')
    prob = options2_problem(tmp_path, 'cross')
    pipe = prob.items['lyalya_qso'].core
    assert pipe.xi.rescale_coords_systematics and pipe.xi.old_growth and not pipe.xi.fht_lowring
    assert oc.chi2(prob) == pytest.approx(float(exp['cross/fid/chi2']), rel=1e-9)
    mask = prob.items['lyalya_qso'].model_mask
    _assert_xi_elementwise(oc.compute_model(prob)['lyalya_qso'], exp['cross/fid/model'], mask, 'cross fid')
    pars = _pars(exp, 'cross/')
    assert oc.chi2(prob, pars[0]) == pytest.approx(float(exp['cross/chi2'][0]), rel=1e-9)
    prob = options2_problem(tmp_path, 'auto')
    assert oc.chi2(prob) == pytest.approx(float(exp['auto/fid/chi2']), rel=1e-9)
    _assert_xi_elementwise(oc.compute_model(prob)['lyalya_lyalya'], exp['auto/fid/model'],
                           prob.items['lyalya_lyalya'].model_mask, 'auto fid')
    assert oc.chi2(prob, {'ap': 1.03, 'at': 0.96, 'uv_shotnoise_amp': 0.03}) == pytest.approx(float(exp['auto/walker/chi2']), rel=1e-9)


def test_oracle_rescaled_covariance_with_marginalize_in_fit(tmp_path):
    """`marginalize-in-fit` + a Monte-Carlo covariance scale (reference vega/vega_interface.py:282-292, :311-313): the oracle
    against what the unmodified reference computed for the mock of tests/golden/make_golden.py::dump_marg_mc."""
    from oracle import vega_cpu as oc
    exp = np.load(GOLDEN / 'expected_marg_mc.npz')
    prob = marginalization_problem(tmp_path, MARGINALIZATION_CASES['rtmax'], in_fit=True)
    mock = {'lyalya_lyalya': exp['mock']}
    chi2, coeff = oc.chi2(prob, data_override=mock, cov_scale=float(exp['scale']), return_marg_coeff=True)
    assert chi2 == pytest.approx(float(exp['fid/chi2']), rel=1e-9)
    scale = np.abs(exp['fid/coeff']).max()
    np.testing.assert_allclose(coeff['lyalya_lyalya'], exp['fid/coeff'], rtol=0, atol=5e-6 * scale)
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    assert oc.chi2(prob, pars, data_override=mock, cov_scale=float(exp['scale'])) == pytest.approx(float(exp['walker0/chi2']), rel=1e-9)


def test_contiguous_share_of_a_mock_file():
    """reference bin/run_vega_mc_fits_mpi.py:134-141: the first n % size ranks take one more; the shares tile [0, n)"""
    from vega_amd.montecarlo import contiguous_share
    for n, size in ((10, 4), (3, 8), (16, 8), (1, 1), (0, 2)):
        shares = [contiguous_share(n, size, r) for r in range(size)]
        assert shares[0][0] == 0 and shares[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(shares, shares[1:]))
        lens = [b - a for a, b in shares]
        assert max(lens) - min(lens) <= 1 and lens == sorted(lens, reverse=True)
    assert contiguous_share(10, 4, 1) == (3, 6) and contiguous_share(10, 4, 3) == (8, 10)


def test_model_only_correlations_match_the_reference(tmp_path):
    """Correlations without a data file (reference vega/correlation_item.py:40-42, :120-136, vega/vega_interface.py:110-137,
    :208-235): the caller's coordinates, no distortion matrix, no mask - the oracle on the lowered problem against the
    unmodified reference (expected_model_only.npz: fiducial point and three walkers, a constant and a per-bin redshift)."""
    from conftest import model_only_problem
    from oracle import vega_cpu as oc
    prob, _ = model_only_problem(tmp_path)
    exp = np.load(GOLDEN / 'expected_model_only.npz')
    assert all(not item.has_data and item.distortion is None for item in prob.items.values())
    fid = oc.compute_model(prob)
    names = [str(n) for n in exp['param_names']]
    for name in prob.items:
        ref = exp[f'fid/{name}']
        np.testing.assert_allclose(fid[name], ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    for i, row in enumerate(exp['theta']):
        got = oc.compute_model(prob, dict(zip(names, row)))
        for name in prob.items:
            ref = exp[f'walker{i}/{name}']
            np.testing.assert_allclose(got[name], ref, rtol=0, atol=1e-9 * np.abs(ref).max())
    # without coordinates the constructor says what it needs
    from vega_amd.setup import build_problem
    with pytest.raises(NotImplementedError, match='coordinates'):
        build_problem('configs/modelonly/main.ini', search_dirs=[tmp_path, GOLDEN])
