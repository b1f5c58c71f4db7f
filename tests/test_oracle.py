"""CPU tests that pin the oracle (oracle/) against the reference's own known answers and against
outputs of the unmodified reference (tests/golden/expected_*.npz, made by make_golden.py)."""
import numpy as np
import pytest
from math import isclose

from oracle import vega_cpu as oc
from oracle.fftlog import P2xi
from conftest import load_problem, GOLDEN


def _params(names, row):
    return {str(n): float(v) for n, v in zip(names, row)}


def test_pinned_log_likelihood():
    """reference tests/test_vega.py:10-14 (isclose default rel_tol 1e-9) and the value the
    unmodified reference produces in the build container (bit-for-bit)."""
    prob = load_problem('full4')
    exp = np.load(GOLDEN / 'expected_full4.npz')
    ll = oc.log_lik(prob)
    assert isclose(ll, -8766.997108462287)
    assert ll == float(exp['log_lik'])
    assert oc.chi2(prob) == float(exp['chi2'])
    model = oc.compute_model(prob)
    for name in prob.items:
        np.testing.assert_array_equal(model[name], exp[f'model/{name}'])


@pytest.mark.parametrize('tag', ['joint', 'joint_metals'])
def test_walkers_match_reference(tag):
    prob = load_problem(tag)
    exp = np.load(GOLDEN / f'expected_{tag}.npz')
    names = exp['param_names']
    assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-13)
    for i in (0, 3, 7):
        pars = _params(names, exp['theta'][i])
        assert oc.chi2(prob, pars) == pytest.approx(float(exp['chi2'][i]), rel=1e-12)
        model = oc.compute_model(prob, pars)
        for name in prob.items:
            ref = exp[f'walker{i}/model/{name}']
            np.testing.assert_allclose(model[name], ref, rtol=1e-11, atol=1e-16)


def test_stage_taps_match_reference():
    prob = load_problem('joint')
    exp = np.load(GOLDEN / 'expected_joint.npz')
    taps = {}
    oc.compute_model(prob, taps=taps)
    for name in prob.items:
        for comp in ('peak', 'smooth'):
            t = taps[name][comp]
            assert t['pk_mean'] == pytest.approx(float(exp[f'fid/taps/{name}/{comp}/pk_mean']), rel=1e-14)
            np.testing.assert_allclose(t['xi_core'], exp[f'fid/taps/{name}/{comp}/xi_core'], rtol=1e-12, atol=1e-18)
            np.testing.assert_allclose(t['xi_distorted'], exp[f'fid/taps/{name}/{comp}/xi_distorted'],
                                       rtol=1e-12, atol=1e-18)
            pk_ells = exp[f'fid/taps/{name}/{comp}/pk_ells']
            for i, ell in enumerate((0, 2, 4, 6)):
                np.testing.assert_allclose(t['pk_ell'][ell], pk_ells[i], rtol=1e-13, atol=1e-20)
        for ell in (0, 2, 4, 6):
            r_fft, xi_fft = taps[name]['smooth']['xi_fft'][ell]
            np.testing.assert_allclose(r_fft, exp[f'fid/taps/{name}/smooth/r_fft_{ell}'], rtol=1e-15)
            np.testing.assert_allclose(xi_fft, exp[f'fid/taps/{name}/smooth/xi_fft_{ell}'], rtol=1e-12, atol=1e-20)


def test_metals_sum_matches_reference():
    prob = load_problem('joint_metals')
    exp = np.load(GOLDEN / 'expected_joint_metals.npz')
    taps = {}
    oc.compute_model(prob, taps=taps)
    for name in prob.items:
        np.testing.assert_allclose(taps[name]['xi_metals'], exp[f'fid/xi_metals/{name}'], rtol=1e-12, atol=1e-20)


def test_synthetic_distortion_and_covariance():
    from vega_amd import synthetic
    prob = load_problem('joint')
    exp = np.load(GOLDEN / 'expected_joint_synth.npz')
    try:
        for item in prob.items.values():
            item.distortion = synthetic.distortion_matrix(item.model_grid.rp, item.model_grid.rt)
            item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt))
        assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-9)
        pars = _params(exp['param_names'], exp['theta'][1])
        assert oc.chi2(prob, pars) == pytest.approx(float(exp['chi2'][1]), rel=1e-9)
        model = oc.compute_model(prob, pars)
        for name in prob.items:
            np.testing.assert_allclose(model[name], exp[f'walker1/model/{name}'], rtol=1e-10, atol=1e-16)
    finally:
        for item in prob.items.values():
            item.distortion = None
            item.set_covariance(None)


def test_picca_golden_vectors():
    """reference tests/test_vega.py:21-44: 7 auto + 7 cross picca vectors, np.allclose defaults,
    through the in-repo Hamilton FFTLog (old_fftlog) incl. relativistic / asymmetry terms."""
    bench = np.load(GOLDEN / 'inputs' / 'picca_bench_data.npz')
    exp = np.load(GOLDEN / 'expected_picca.npz')
    for kind, main, hdu in (('auto', 'configs/picca/main.ini', 1), ('cross', 'configs/picca/main_cross.ini', 2)):
        prob = load_problem(main, fiducial_overrides=(('Omega_de', None),))
        model = oc.compute_model(prob)
        for name, xi in model.items():
            picca = bench[f'{hdu}/{kind}_{name}']
            assert np.allclose(xi, picca), (kind, name)
            np.testing.assert_allclose(xi, exp[f'{kind}_{name}'], rtol=1e-10, atol=1e-14)


# ---- P(k, mu) known answers: constants of reference tests/test_pk.py -----------------------------
def _pk_setup():
    from vega_amd.tables import read_tables
    from vega_amd.setup import PkOptions, XiOptions, Pipeline, Tracer
    tab = read_tables(GOLDEN / 'inputs' / 'PlanckDR16.npz')[0]
    k, pk_full, pk_smooth = tab.data['K'], tab.data['PK'], tab.data['PKSB']
    z_fid, z_eff = tab.header['ZREF'], 2.25
    pk_fid = pk_full * ((1 + z_fid) / (1. + z_eff))**2
    grid = oc.PkGrid(k, 1000)
    lya, qso = Tracer('LYA', 'continuous'), Tracer('QSO', 'discrete')

    def pipe(t1, t2, name, **kw):
        dummy = np.zeros(1)
        return Pipeline(t1, t2, name, PkOptions(**kw), XiOptions(), False,
                        dummy, dummy, dummy, dummy, dummy)
    return grid, pk_full, pk_smooth, pk_fid, lya, qso, pipe


def test_bias_beta_algebra():
    """reference tests/test_pk.py:10-52"""
    assert oc.bias_beta({'bias_LYA': -0.12, 'beta_LYA': 1.6}, 'LYA', 'LYA') == (-0.12, 1.6, -0.12, 1.6)
    b1, be1, _, _ = oc.bias_beta({'bias_eta_LYA': -0.2, 'beta_LYA': 1.6, 'growth_rate': 0.97}, 'LYA', 'LYA')
    assert b1 == pytest.approx(-0.2 * 0.97 / 1.6) and be1 == 1.6
    b1, be1, _, _ = oc.bias_beta({'bias_eta_LYA': -0.2, 'bias_LYA': -0.12, 'growth_rate': 0.97}, 'LYA', 'LYA')
    assert b1 == -0.12 and be1 == pytest.approx(-0.2 * 0.97 / -0.12)
    pars = {'bias_LYA': -0.12, 'beta_LYA': 1.6, 'bias_eta_QSO': 1, 'beta_QSO': 0.25, 'growth_rate': 0.97}
    b1, be1, b2, be2 = oc.bias_beta(pars, 'LYA', 'QSO')
    assert (b1, be1, be2) == (-0.12, 1.6, 0.25) and b2 == pytest.approx(0.97 / 0.25)
    pars = {'bias_eta_LYA': -0.2, 'beta_LYA': 1.6, 'bias_eta_QSO': 1, 'bias_QSO': 3.7, 'growth_rate': 0.97}
    b1, be1, b2, be2 = oc.bias_beta(pars, 'LYA', 'QSO')
    assert b1 == pytest.approx(-0.2 * 0.97 / 1.6) and b2 == 3.7 and be2 == pytest.approx(0.97 / 3.7)


def test_pk_known_answers():
    """Every known answer of reference tests/test_pk.py (pytest.approx default rel 1e-6)."""
    grid, pk_full, pk_smooth, pk_fid, lya, qso, pipe = _pk_setup()
    approx = pytest.approx

    # auto_pk (:153-231)
    base = {'bias_LYA': -0.12, 'beta_LYA': 1.6, 'peak': False}
    p = pipe(lya, lya, 'lyaxlya', use_gk=False)
    kaiser = (1 + 1.6 * grid.mu**2) * (1 + 1.6 * grid.mu**2) * (-0.12 * -0.12)
    assert np.sum(kaiser) == approx(37.13279)
    assert np.allclose(oc.power_spectrum(p, grid, pk_smooth, pk_fid, base), pk_smooth * kaiser)
    assert np.sum(oc._gk(p.pk, grid, 'lyaxlya', {'par binsize lyaxlya': 2, 'per binsize lyaxlya': 3})) \
        == approx(470301.136422)
    gk = oc._gk(p.pk, grid, 'lyaxlya', base)
    assert np.sum(gk) == approx(450783.949889)
    p = pipe(lya, lya, 'lyaxlya')
    assert np.allclose(oc.power_spectrum(p, grid, pk_smooth, pk_fid, base), pk_smooth * kaiser * gk)
    assert np.mean(oc.power_spectrum(p, grid, pk_smooth, pk_fid, base, fast_metals=True)) == approx(1228.9847366)

    # UV / HCD effective biases (:55-93)
    full = pipe(lya, lya, 'lyaxlya', hcd_model='Rogers', uvb=True, small_scale_nl='arinyo',
                fullshape_smoothing='gauss')
    uv = {'bias_gamma': 0.1125, 'bias_prim': -0.66, 'lambda_uv': 300}
    b_uv, be_uv = oc._uv_heii(full.pk, grid, -0.12, 1.6, uv)
    assert np.sum(b_uv) == approx(-35.268497) and np.sum(be_uv) == approx(1138.77689)
    hcd = {'bias_hcd': -0.05, 'beta_hcd': 0.5, 'L0_hcd': 10}
    b_e, be_e = oc._hcd(full.pk, grid, 'LYAxLYA', -0.12, 1.6, hcd)
    assert np.sum(b_e) == approx(-116031.686) and np.sum(be_e) == approx(1179867.64849)
    from vega_amd.setup import load_fvoigt_table
    voigt = pipe(lya, lya, 'lyaxlya', hcd_model='fvoigt',
                 fvoigt_table=load_fvoigt_table('fvoigt_models/Fvoigt_exp.txt', [GOLDEN / 'inputs']))
    b_e, be_e = oc._hcd(voigt.pk, grid, 'LYAxLYA', -0.12, 1.6, hcd)
    assert np.sum(b_e) == approx(-121782.768388) and np.sum(be_e) == approx(1142662.6535)
    sinc = pipe(lya, lya, 'lyaxlya', hcd_model='sinc')
    b_e, be_e = oc._hcd(sinc.pk, grid, 'LYAxLYA', -0.12, 1.6, dict(hcd, L0_sinc=10))
    assert np.sum(b_e) == approx(-118530.3944) and np.sum(be_e) == approx(1166657.39777)

    # peak NL (:96-117)
    assert np.sum(oc._peak_nl(grid, {'sigmaNL_par': 6.36984, 'sigmaNL_per': 3.24})) == approx(390698.51738)
    assert np.sum(oc._peak_nl(grid, {'sigmaNL_par': 6.36984, 'growth_rate': 0.97})) == approx(390747.02382)
    assert np.sum(oc._peak_nl(grid, {'sigmaNL_per': 3.24, 'growth_rate': 0.97})) == approx(390645.39796)

    # small-scale non-linear terms (:120-130)
    ar = {'dnl_arinyo_q1': 0.8558, 'dnl_arinyo_kv': 1.11454, 'dnl_arinyo_av': 0.5378,
          'dnl_arinyo_bv': 1.607, 'dnl_arinyo_kp': 19.47}
    assert np.sum(oc._arinyo(grid, pk_fid, 'LYA', 'LYA', ar)) == approx(680327.61617)
    assert np.sum(oc._mcdonald(grid)) == approx(632262.53194)

    # full-shape smoothing (:133-142)
    sm = {'par_sigma_smooth': 2, 'per_sigma_smooth': 2.5}
    assert np.sum(oc._fullshape_gauss(grid, 'LYA', 'LYA', sm)) == approx(404166.27948)
    assert np.sum(oc._fullshape_exp(grid, dict(sm, par_exp_smooth=2, per_exp_smooth=2.5))) == approx(333204.95791)

    # velocity dispersion (:145-151)
    vd = {'sigma_velo_disp_gauss_QSO': 6.8, 'sigma_velo_disp_lorentz_QSO': 7.2}
    assert np.sum(oc._velocity_dispersion('gauss', grid, lya, qso, vd)) == approx(435379.6457)
    assert np.sum(oc._velocity_dispersion('lorentz', grid, lya, qso, vd)) == approx(446899.3964)

    # full auto P(k, mu) (:216-231)
    pars = dict(base, **uv, **hcd, **ar, sigmaNL_par=6.36984, sigmaNL_per=3.24,
                par_sigma_smooth=2, per_sigma_smooth=2.5)
    pars['peak'] = True
    assert np.mean(oc.power_spectrum(full, grid, pk_full - pk_smooth, pk_fid, pars)) == approx(2.8794436016)
    pars['peak'] = False
    assert np.mean(oc.power_spectrum(full, grid, pk_smooth, pk_fid, pars)) == approx(19.67878957)

    # full cross P(k, mu) (:234-266)
    cross = pipe(lya, qso, 'lyaxqso', hcd_model='Rogers', uvb=True, fullshape_smoothing='gauss',
                 velocity_dispersion='lorentz')
    pars = {'bias_LYA': -0.12, 'beta_LYA': 1.6, 'bias_QSO': 3.7, 'beta_QSO': 0.26, **uv, **hcd,
            'sigmaNL_par': 6.36984, 'sigmaNL_per': 3.24, 'par_sigma_smooth': 2, 'per_sigma_smooth': 2.5,
            'sigma_velo_disp_lorentz_QSO': 7.2}
    pars['peak'] = True
    assert np.mean(oc.power_spectrum(cross, grid, pk_full - pk_smooth, pk_fid, pars)) == approx(-2.9406788865)
    pars['peak'] = False
    assert np.mean(oc.power_spectrum(cross, grid, pk_smooth, pk_fid, pars)) == approx(-401.0937936)


def test_fftlog_matrix_is_the_transform():
    """The explicit operator used to build the engine's static matrices equals the FFT path."""
    grid, pk_full, pk_smooth, pk_fid, lya, qso, pipe = _pk_setup()
    rng = np.random.default_rng(1)
    f = pk_smooth * (1 + 0.1 * rng.standard_normal(pk_smooth.size))
    for ell in (0, 2):
        t = P2xi(grid.k, l=ell)
        r, xi = t(f)
        sel = (r > 0.5) & (r < 1000.)      # the bins only ever ask for separations of a few to ~600 Mpc/h
        np.testing.assert_allclose((t.matrix() @ f)[sel], xi[sel], rtol=0, atol=1e-12 * np.abs(xi[sel]).max())


def test_uv_shotnoise_and_instrumental_systematics_match_reference():
    """Additive terms pinned by the unmodified reference (tests/golden/expected_extras.npz)."""
    prob = load_problem('auto_extras')
    exp = np.load(GOLDEN / 'expected_extras.npz')
    assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-12)
    np.testing.assert_allclose(oc.compute_model(prob)['lyalya_lyalya'], exp['fid/model'], rtol=1e-11, atol=1e-16)
    pars = _params(exp['param_names'], exp['theta'][0])
    assert oc.chi2(prob, pars) == pytest.approx(float(exp['walker0/chi2']), rel=1e-12)
    np.testing.assert_allclose(oc.compute_model(prob, pars)['lyalya_lyalya'], exp['walker0/model'],
                               rtol=1e-11, atol=1e-16)


def test_fast_metals_caches_match_reference():
    """`fast_metals = True` (reference metals.py:144-207): metal x metal correlations frozen at the first
    evaluation and the pair-blind per-call cache of the main x metal correlations - the oracle reproduces the
    reference's sequence (fiducial first, then walkers) bit for bit."""
    prob = load_problem('joint_metals_fast')
    exp = np.load(GOLDEN / 'expected_joint_metals_fast.npz')
    oc.reset_metal_cache(prob)
    assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-13)
    names = [str(n) for n in exp['param_names']]
    for i, row in enumerate(exp['theta']):
        pars = dict(zip(names, row))
        assert oc.chi2(prob, pars) == pytest.approx(float(exp['chi2'][i]), rel=1e-13)
        model = oc.compute_model(prob, pars)
        for name in prob.items:
            ref = exp[f'walker{i}/model/{name}']
            assert np.abs(model[name] - ref).max() <= 1e-13 * np.abs(ref).max()
    # a different first evaluation freezes different metal x metal terms: the mode is history dependent
    oc.reset_metal_cache(prob)
    first = dict(zip(names, exp['theta'][0]))
    oc.chi2(prob, first)
    assert oc.chi2(prob) != pytest.approx(float(exp['fid/chi2']), rel=1e-9)
    oc.reset_metal_cache(prob)


def test_mock_binning_matches_reference():
    """`mock-bin-size` + `mock-los-smoothing = growth` (reference power_spectrum.py:143-160)."""
    prob = load_problem('auto_mockbin')
    exp = np.load(GOLDEN / 'expected_mockbin.npz')
    assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-13)
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    assert oc.chi2(prob, pars) == pytest.approx(float(exp['walker0/chi2']), rel=1e-13)
    model = oc.compute_model(prob, pars)['lyalya_lyalya']
    assert np.abs(model - exp['walker0/model']).max() <= 1e-13 * np.abs(model).max()


@pytest.mark.parametrize('mode', ['amplitude', 'growth'])
def test_mock_binning_that_follows_a_varying_parameter_matches_reference(mode):
    """`mock-los-smoothing = amplitude | growth` with `los_smooth_amp` / `growth_rate` different from walker to walker
    (reference power_spectrum.py:143-160): the binning kernel is then a function of the walker."""
    import copy
    prob = copy.deepcopy(load_problem('auto_mockbin'))
    prob.items['lyalya_lyalya'].core.pk.mock_los_smoothing = mode
    exp = np.load(GOLDEN / 'expected_mockbin_sampled.npz')
    names = [str(n) for n in exp[f'{mode}/param_names']]
    for i, row in enumerate(exp[f'{mode}/theta']):
        pars = dict(zip(names, map(float, row)))
        assert oc.chi2(prob, pars) == pytest.approx(float(exp[f'{mode}/chi2'][i]), rel=1e-12)
        model = oc.compute_model(prob, pars)['lyalya_lyalya']
        assert np.abs(model - exp[f'{mode}/model'][i]).max() <= 1e-13 * np.abs(model).max()
    assert len(set(np.round(exp[f'{mode}/chi2'], 6))) == 4


def test_fht_extrap_matches_reference():
    """`fht_extrap = True` (reference vega/pktoxi.py:41,141) on a model whose spectra keep non-zero end samples; with the test
    configuration's own model the reference returns NaN (0 / 0 end segments) - so does the oracle."""
    import copy
    exp = np.load(GOLDEN / 'expected_fht_extrap.npz')
    prob = copy.deepcopy(load_problem('auto'))
    core = prob.items['lyalya_lyalya'].core
    core.xi.fht_extrap = True
    with np.errstate(all='ignore'):
        assert np.isnan(oc.chi2(prob)) and np.isnan(exp['default_model/chi2'])
    core.pk.small_scale_nl, core.pk.fullshape_smoothing, core.pk.use_gk = None, None, False
    names = [str(n) for n in exp['param_names']]
    for i, row in enumerate(exp['theta']):
        pars = dict(zip(names, map(float, row)))
        assert oc.chi2(prob, pars) == pytest.approx(float(exp['chi2'][i]), rel=1e-12)
        model = oc.compute_model(prob, pars)['lyalya_lyalya']
        assert np.abs(model - exp['model'][i]).max() <= 1e-13 * np.abs(model).max()


def test_fits_ingestion_matches_reference(tmp_path):
    """Static-state ingestion (reference vega/data.py:285-473, utils.py:271-298): distortion matrix and covariance
    read from vector columns of a FITS data file.  The reference read the same file (same writer, same generator)
    for tests/golden/expected_fits_ingest.npz."""
    from conftest import fits_ingest_problem
    from vega_amd import synthetic
    prob = fits_ingest_problem(tmp_path)
    exp = np.load(GOLDEN / 'expected_fits_ingest.npz')
    item = prob.items['lyalya_lyalya']
    assert item.data_size == int(exp['data_size'])
    np.testing.assert_array_equal(item.distortion.toarray(),
                                  synthetic.distortion_matrix(item.model_grid.rp, item.model_grid.rt))
    assert item.log_cov_det == pytest.approx(float(exp['log_cov_det']), rel=1e-12)
    assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-10)
    assert oc.log_lik(prob) == pytest.approx(float(exp['fid/log_lik']), rel=1e-10)
    model = oc.compute_model(prob)['lyalya_lyalya']
    assert np.abs(model - exp['fid/model']).max() <= 1e-12 * np.abs(model).max()
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    assert oc.chi2(prob, pars) == pytest.approx(float(exp['walker0/chi2']), rel=1e-10)


def test_direct_pk_with_metal_terms_matches_reference():
    """`direct_pk` + `no-metal-decomp = False` (reference model.py:188-207 -> :120-123): the metal terms are part of the
    direct model, on the caller's spectrum (tests/golden/make_golden.py::dump_direct_pk_metals)."""
    prob = load_problem('auto_metals')
    prob.items['lyalya_lyalya'].metal_opts['no_metal_decomp'] = False
    try:
        exp = np.load(GOLDEN / 'expected_direct_pk_metals.npz')
        pk = exp['direct_pk']
        assert oc.chi2(prob, direct_pk=pk) == pytest.approx(float(exp['fid/chi2']), rel=1e-12)
        model = oc.compute_model(prob, direct_pk=pk)['lyalya_lyalya']
        assert np.abs(model - exp['fid/model']).max() <= 1e-13 * np.abs(model).max()
        names = [str(n) for n in exp['param_names']]
        for i, row in enumerate(exp['theta']):
            got = oc.chi2(prob, dict(zip(names, row)), direct_pk=pk * (1 + 0.01 * (i + 1)))
            assert got == pytest.approx(float(exp['chi2'][i]), rel=1e-12)
    finally:
        prob.items['lyalya_lyalya'].metal_opts['no_metal_decomp'] = True


def test_metal_decomposition_matches_reference():
    """`no-metal-decomp = False` (reference model.py:120-123, :181-186): metals per component."""
    prob = load_problem('auto_metals')
    prob.items['lyalya_lyalya'].metal_opts['no_metal_decomp'] = False
    exp = np.load(GOLDEN / 'expected_metal_decomp.npz')
    assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-12)
    model = oc.compute_model(prob)['lyalya_lyalya']
    assert np.abs(model - exp['fid/model']).max() <= 1e-13 * np.abs(model).max()
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    assert oc.chi2(prob, pars) == pytest.approx(float(exp['chi2'][0]), rel=1e-12)
    prob.items['lyalya_lyalya'].metal_opts['no_metal_decomp'] = True


@pytest.mark.parametrize('tag', ['auto', 'cross', 'auto_rp'])
def test_new_metals_matches_reference(tmp_path, tag):
    """`new_metals = True` (reference vega/metals.py:83-112, :389-752): the matrices and effective coordinates built by
    vega_amd/metal_matrices.py from a stacked-delta file / object catalogue against the ones the unmodified reference
    built from the same files (both on the restated picca wavelengths and comoving distance), then chi2."""
    from conftest import new_metals_problem
    prob, name = new_metals_problem(tmp_path, tag)
    exp = np.load(GOLDEN / 'expected_new_metals.npz')
    item = prob.items[name]
    pairs = [tuple(str(p).split('|')) for p in exp[f'{tag}/pairs']]
    assert [m.names for m in item.metals] == pairs
    probe = np.cos(0.013 * np.arange(item.model_grid.size))
    for i, metal in enumerate(item.metals):
        np.testing.assert_allclose(metal.pipeline.r, exp[f'{tag}/pair{i}/r'], rtol=1e-13)
        np.testing.assert_allclose(metal.pipeline.mu, exp[f'{tag}/pair{i}/mu'], rtol=0, atol=1e-13)
        np.testing.assert_allclose(metal.pipeline.z, exp[f'{tag}/pair{i}/z'], rtol=1e-13)
        want = exp[f'{tag}/pair{i}/applied']
        assert np.abs(metal.matrix.dot(probe) - want).max() <= 1e-13 * np.abs(want).max()
    assert oc.chi2(prob) == pytest.approx(float(exp[f'{tag}/chi2']), rel=1e-11)
    model = oc.compute_model(prob)[name]
    assert np.abs(model - exp[f'{tag}/model']).max() <= 1e-12 * np.abs(model).max()
    pars = {str(n): float(v) for n, v in zip(exp[f'{tag}/param_names'], exp[f'{tag}/theta'][0])}
    assert oc.chi2(prob, pars) == pytest.approx(float(exp[f'{tag}/walker0/chi2']), rel=1e-11)


def test_new_bias_evolution_matches_reference(tmp_path):
    """`new-bias-evolution` (reference correlation_func.py:238-299) with the cosmology of the data file's header:
    against the unmodified reference on the same file (picca's D_H restated on both sides)."""
    from conftest import new_bias_evol_problem
    prob = new_bias_evol_problem(tmp_path)
    exp = np.load(GOLDEN / 'expected_new_bias_evol.npz')
    assert prob.items['lyalya_qso'].core.rel_z_evol_1 is not None
    assert oc.chi2(prob) == pytest.approx(float(exp['fid/chi2']), rel=1e-11)
    model = oc.compute_model(prob)['lyalya_qso']
    assert np.abs(model - exp['fid/model']).max() <= 1e-13 * np.abs(model).max()
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    assert oc.chi2(prob, pars) == pytest.approx(float(exp['walker0/chi2']), rel=1e-11)


def test_blinding_matches_reference(tmp_path):
    """Blinded data (reference vega/data.py:305-339: DA_BLIND replaces DA for `desi_dr3`) and parameter-level
    blinding (vega_interface.py:389-421, utils.py:375-393: p += pi - exp(v^2), seen by the model and the priors),
    against the unmodified reference on the same file with `_rnsps` set to the same offsets."""
    from conftest import blinding_problem
    from vega_amd import synthetic
    from vega_amd.setup import init_blinding
    prob = blinding_problem(tmp_path)
    exp = np.load(GOLDEN / 'expected_blinding.npz')
    item = prob.items['lyalya_lyalya']
    assert item.blind and item.blinding_strat == 'desi_dr3'
    assert init_blinding(prob.items, prob.sample_params) == (True, [])
    assert oc.chi2(prob) == pytest.approx(float(exp['plain/chi2']), rel=1e-12)
    assert oc.log_lik(prob) == pytest.approx(float(exp['plain/log_lik']), rel=1e-12)
    prob.blinding_offsets = synthetic.blinding_offsets()
    assert oc.chi2(prob) == pytest.approx(float(exp['offsets/chi2']), rel=1e-12)
    assert oc.log_lik(prob) == pytest.approx(float(exp['offsets/log_lik']), rel=1e-12)
    assert oc.prior_chi2(prob) == pytest.approx(float(exp['offsets/prior_chi2']), rel=1e-12)
    model = oc.compute_model(prob)['lyalya_lyalya']
    assert np.abs(model - exp['offsets/model']).max() <= 1e-13 * np.abs(model).max()
    for row, want in zip(exp['theta'], exp['offsets/walker_chi2']):
        pars = {str(n): float(v) for n, v in zip(exp['param_names'], row)}
        assert oc.chi2(prob, pars) == pytest.approx(float(want), rel=1e-12)


def test_blinding_checks_mirror_the_reference(tmp_path):
    """vega_interface.py:853-886: sampling a blinded parameter on `desi_dr3` data stops with the reference's own
    message (it has no offsets file for that strategy); the full-shape scale parameters must stay fixed."""
    from conftest import blinding_problem
    from vega_amd.setup import init_blinding, blinding_transform
    exp = np.load(GOLDEN / 'expected_blinding.npz')
    prob = blinding_problem(tmp_path, sample_extra='growth_rate = True')
    with pytest.raises(ValueError) as err:
        init_blinding(prob.items, prob.sample_params)
    assert str(err.value) == str(exp['sampled_blinded_error'])
    sampled = {'limits': {'ap_full': (0.5, 1.5)}}
    with pytest.raises(ValueError, match='ap_full must be fixed'):
        init_blinding(prob.items, sampled)
    with pytest.raises(ValueError, match='bias_QSO and beta_QSO'):
        init_blinding(prob.items, {'limits': {'bias_QSO': (0, 1), 'beta_QSO': (0, 1)}})
    scale, shift = blinding_transform(['ap', 'ap_full', 'beta_LYA'], {'ap': 0.3})
    assert scale.tolist() == [1.0, 0.0, 1.0] and shift.tolist() == [np.pi - np.exp(0.09), 1.0, 0.0]


@pytest.mark.parametrize('tag', ['rtmax', 'allrmin', 'fitscales'])
def test_small_scale_marginalization_matches_reference(tmp_path, tag):
    """Small-scale marginalisation (reference vega/correlation_item.py:175-268, vega/data.py:96-128, :762-828): the
    set-up time covariance update, against chi2 / log-likelihood of the unmodified reference on the same file."""
    from conftest import marginalization_problem, MARGINALIZATION_CASES
    prob = marginalization_problem(tmp_path, MARGINALIZATION_CASES[tag])
    exp = np.load(GOLDEN / 'expected_marginalization.npz')
    item = prob.items['lyalya_lyalya']
    assert np.trace(item.cov_marg_update) == pytest.approx(float(exp[f'{tag}/cov_update_trace']), rel=1e-10)
    # the variance the result file writes (<name>_VAR): the reference's is a live view of the UPDATED covariance's diagonal
    np.testing.assert_allclose(item.variance, exp[f'{tag}/variance'], rtol=1e-9)
    assert oc.chi2(prob) == pytest.approx(float(exp[f'{tag}/chi2']), rel=1e-8)
    assert oc.log_lik(prob) == pytest.approx(float(exp[f'{tag}/log_lik']), rel=1e-8)


@pytest.mark.parametrize('tag', ['rtmax', 'allrmin', 'fitscales'])
def test_marginalize_in_fit_matches_reference(tmp_path, tag):
    """`marginalize-in-fit` (reference vega_interface.py:282-292, :546-579): the oracle fits the template coefficients
    from the residual and adds the templates to the model, as the reference does; the product path uses the
    equivalent static matrix P^T C^-1 P (CorrItem.chi2_matrix) - both against the reference's values."""
    from conftest import marginalization_problem, MARGINALIZATION_CASES
    prob = marginalization_problem(tmp_path, MARGINALIZATION_CASES[tag], in_fit=True)
    exp = np.load(GOLDEN / 'expected_marginalization.npz')
    np.testing.assert_allclose(prob.items['lyalya_lyalya'].variance, exp[f'{tag}/infit/variance'], rtol=1e-12)      # (no update in fit)
    assert oc.chi2(prob) == pytest.approx(float(exp[f'{tag}/infit/chi2']), rel=1e-9)
    assert oc.log_lik(prob) == pytest.approx(float(exp[f'{tag}/infit/log_lik']), rel=1e-9)
    pars = {str(n): float(v) for n, v in zip(exp[f'{tag}/infit/param_names'], exp[f'{tag}/infit/theta'][0])}
    assert oc.chi2(prob, pars) == pytest.approx(float(exp[f'{tag}/infit/walker0/chi2']), rel=1e-9)
    # the static form the engine is given
    item = prob.items['lyalya_lyalya']
    diff = item.masked_data_vec - oc.compute_model(prob, pars)['lyalya_lyalya'][item.model_mask]
    assert diff.dot(item.chi2_matrix.dot(diff)) == pytest.approx(float(exp[f'{tag}/infit/walker0/chi2']), rel=1e-9)


def test_direct_pk_matches_reference():
    """`direct_pk` (reference vega_interface.py:208-248 -> model.py:188-207): one component from the caller's spectrum,
    no metals with the default no-metal-decomp, additive broadband entering once."""
    prob = load_problem('joint_metals')
    exp = np.load(GOLDEN / 'expected_direct_pk.npz')
    assert oc.chi2(prob, direct_pk=exp['direct_pk']) == pytest.approx(float(exp['fid/chi2']), rel=1e-13)
    names = [str(n) for n in exp['param_names']]
    for i, row in enumerate(exp['theta']):
        got = oc.chi2(prob, dict(zip(names, row)), direct_pk=exp['direct_pk'] * (1 + 0.01 * (i + 1)))
        assert got == pytest.approx(float(exp['chi2'][i]), rel=1e-13)
