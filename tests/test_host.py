"""CPU tests of the host logic: config lowering, static operators, synthetic tensors, sharding."""
import numpy as np
import pytest
from scipy import interpolate

from conftest import load_problem, GOLDEN


class _FakeEngine:
    def __init__(self):
        self.tables = {}

    def _gk_table(self, a, b, c=0.0, d=0.0):
        return self.tables.setdefault((a, b), len(self.tables))


def test_lowering_resolves_the_reference_lookups():
    from vega_amd import engine as E
    prob = load_problem('joint_metals')
    low = E.Lowering(prob)
    fake = _FakeEngine()
    auto = prob.items['lyalya_lyalya']
    peak = low.pipeline(fake, auto.core, 'peak')
    smooth = low.pipeline(fake, auto.core, 'smooth')
    assert peak.scale_mode == E.SCALE_AP_AT and smooth.scale_mode == E.SCALE_UNIT
    assert peak.peak_nl == 1 and smooth.peak_nl == 0
    assert peak.hcd_model == E.HCD['Rogers'] and peak.uvb == 1 and peak.nl_model == E.NL['arinyo']
    assert peak.arinyo_power == 1.0 and peak.n_ell == 4 and peak.same_tracer == 1
    # bias_eta + beta given -> bias derived (reference utils.py:71-73)
    assert peak.tracer[0].bias_slot == -1 and peak.tracer[0].bias_eta_slot >= 0
    # 'par_sigma_smooth' present -> one squared Gaussian term (reference power_spectrum.py:520-535)
    assert peak.n_smooth == 1 and peak.smooth_weight[0] == 1.0
    cross = prob.items['lyalya_qso']
    xs = low.pipeline(fake, cross.core, 'smooth')
    assert xs.arinyo_power == 0.5 and xs.vd_kind == E.VD['lorentz'] and xs.radiation == 1
    assert xs.drp_slot == low.slot['drp_QSO'] and xs.tracer[1].discrete == 1
    assert len(fake.tables) == 1
    # metal pairs: no HCD / UV / NL, unit scaling, Kaiser without biases
    m = auto.metals[0]
    md = low.pipeline(fake, m.pipeline, 'full', fast_metals=True)
    assert md.scale_mode == E.SCALE_UNIT and md.hcd_model == 0 and md.nl_model == 0 and md.fast_metals == 1
    assert [p.names for p in auto.metals][:5] == [('SiII(1190)', 'LYA'), ('SiII(1193)', 'LYA'),
                                                  ('SiIII(1207)', 'LYA'), ('SiII(1260)', 'LYA'),
                                                  ('SiII(1190)', 'SiII(1190)')]
    assert len(auto.metals) == 15 and len(cross.metals) == 4
    assert sum(p.double_count for p in auto.metals) == 6 + 4


def test_unsupported_options_fail_loudly():
    from vega_amd import engine as E
    prob = load_problem('auto')
    low = E.Lowering(prob)
    core = prob.items['lyalya_lyalya'].core
    # a binning kernel that follows a SAMPLED parameter: lowered to a per-walker factor, the table keeps the transverse one
    core.pk.mock_bin_size, core.pk.mock_los_smoothing = 2.0, 'amplitude'
    prob.params['los_smooth_amp'] = 0.3
    prob.sample_params['limits']['los_smooth_amp'] = (0., 1.)
    try:
        low2 = E.Lowering(prob)
        fake = _FakeEngine()
        d = low2.pipeline(fake, core, 'smooth')
        assert d.mock_los_slot == low2.slot['los_smooth_amp'] and d.mock_los_size == 2.0
        del prob.sample_params['limits']['los_smooth_amp']
        d = E.Lowering(prob).pipeline(_FakeEngine(), core, 'smooth')
        assert d.mock_los_slot == -1            # (not sampled: folded into the static table)
    finally:
        core.pk.mock_bin_size, core.pk.mock_los_smoothing = None, None
        del prob.params['los_smooth_amp']
        prob.sample_params['limits'].pop('los_smooth_amp', None)
    core.xi.single_multipole = 3
    try:
        with pytest.raises(ValueError):
            low.pipeline(_FakeEngine(), core, 'smooth')
    finally:
        core.xi.single_multipole = -1


def test_hamilton_operator_is_the_legacy_transform():
    from vega_amd.fftlog_op import hamilton_xi_operator
    from oracle.vega_cpu import hamilton_multipoles, PkGrid
    prob = load_problem('auto')
    grid = PkGrid(prob.k, 1000)
    pk2d = prob.pk_smooth * (1 + 1.5 * grid.mu**2)**2 * np.exp(-(prob.k * grid.mu)**2)
    ar = np.linspace(4., 300., 500)
    ref = hamilton_multipoles(ar, prob.k, pk2d, (0, 2, 4), grid.mu, 1e-3)
    for i, ell in enumerate((0, 2, 4)):
        from scipy import special
        pk_ell = np.sum(1e-3 * special.legendre(ell)(grid.mu) * pk2d, axis=0) * (2 * ell + 1)
        op, x0, h, n = hamilton_xi_operator(prob.k, ell)
        c = op @ pk_ell
        u = (np.log(ar) - x0) / h
        j = np.floor(u).astype(int)
        t = u - j
        s = (c[j] * (1 - t)**3 + c[j + 1] * (3 * t**3 - 6 * t**2 + 4)
             + c[j + 2] * (-3 * t**3 + 3 * t**2 + 3 * t + 1) + c[j + 3] * t**3) / 6
        assert np.abs(s - ref[i]).max() <= 1e-12 * np.abs(ref[i]).max()


def test_masks_and_sizes_match_the_reference_fixture_facts():
    prob = load_problem('full4')
    sizes = {n: (it.model_grid.size, it.data_size) for n, it in prob.items.items()}
    assert sizes['lyalya_lyalya'] == (2500, 1590) and sizes['lyalya_qso'] == (5000, 3180)


def test_xi_operator_reproduces_fftlog_plus_cubic_spline():
    """OP_ell . P_ell evaluated with four B-spline taps == scipy interp1d(cubic) of the FFTLog."""
    from vega_amd.fftlog_op import xi_operator
    from oracle.fftlog import P2xi
    prob = load_problem('auto')
    f = prob.pk_smooth * np.exp(-prob.k**2 * 4) * 0.02
    xs = np.log(np.linspace(1.5, 700., 4001))
    for ell in (0, 2, 4, 6):
        op, x0, h, n = xi_operator(prob.k, ell)
        assert op.shape == (n + 2, n)
        r, xi = P2xi(prob.k, l=ell)(f)
        ref = interpolate.interp1d(np.log(r), xi, kind='cubic')(xs)
        c = op @ f
        u = (xs - x0) / h
        j = np.floor(u).astype(int)
        t = u - j
        s = (c[j] * (1 - t)**3 + c[j + 1] * (3 * t**3 - 6 * t**2 + 4)
             + c[j + 2] * (-3 * t**3 + 3 * t**2 + 3 * t + 1) + c[j + 3] * t**3) / 6
        assert np.abs(s - ref).max() <= 1e-12 * np.abs(ref).max()


def test_synthetic_tensors_are_deterministic_and_well_formed():
    from vega_amd import synthetic
    prob = load_problem('auto')
    grid = prob.items['lyalya_lyalya'].model_grid
    dm = synthetic.distortion_matrix(grid.rp[:400], grid.rt[:400])
    assert np.array_equal(dm, synthetic.distortion_matrix(grid.rp[:400], grid.rt[:400]))
    np.testing.assert_allclose(dm.sum(axis=1), 0.7, rtol=1e-12)
    assert 0.2 < (dm == 0).mean() < 0.6
    cov = synthetic.covariance(grid.rp[:300], grid.rt[:300])
    assert np.allclose(cov, cov.T) and np.linalg.eigvalsh(cov).min() > 0
    theta = synthetic.walkers(np.array([1.0, 0.0, 9e99]), ['ap', 'x', 'qso_rad_lifetime'], 5)
    assert theta.shape == (5, 3) and np.all(theta[:, 2] == 9e99) and np.ptp(theta[:, 0]) > 0
    # the generators keep their results per grid (read-only, the same object again; another grid or argument is another entry;
    # the kept inverse is the item's own inverse), and a fresh generation after forget() gives the same bits
    assert synthetic.distortion_matrix(grid.rp[:400], grid.rt[:400]) is dm and not dm.flags.writeable
    assert synthetic.distortion_matrix(grid.rp[:400], grid.rt[:400], dense_fraction=0.5) is not dm
    with pytest.raises(ValueError):
        dm[0, 0] = 1.0
    mask = np.arange(300) % 3 != 0
    inv = synthetic.inverse_masked_covariance(grid.rp[:300], grid.rt[:300], mask)
    np.testing.assert_array_equal(inv, np.linalg.inv(cov[:, mask][mask, :]))
    kept = dm.copy()
    synthetic.forget()
    again = synthetic.distortion_matrix(grid.rp[:400], grid.rt[:400])
    assert again is not dm and np.array_equal(again, kept)


def test_table_bundle_round_trip(tmp_path):
    from vega_amd.tables import Table, write_bundle, read_tables
    t = Table({'RPMIN': 0.0, 'NP': 50, 'NAXIS1': 8}, {'RP': np.arange(5.), 'DM': np.eye(3)})
    write_bundle(tmp_path / 'x.npz', [t])
    back = read_tables(tmp_path / 'x.npz')[0]
    assert back.header == {'RPMIN': 0.0, 'NP': 50} and back.names == ['RP', 'DM']
    np.testing.assert_array_equal(back.data['DM'], np.eye(3))


def test_fast_metal_plan_follows_the_reference_caches():
    """metals_plan.fast_metal_plan: metal x metal pairs become static vectors, main x metal pairs with equal betas
    share the first such pair's pipeline (the reference's per-call cache is blind to the tracer pair), unless a
    sampled parameter could separate their betas later."""
    import copy
    from vega_amd import metals_plan
    prob = load_problem('joint_metals_fast')
    assert metals_plan.needs_freeze(prob) and not metals_plan.needs_freeze(load_problem('joint_metals'))
    calls = []

    def metal_xi(name, mi):
        calls.append((name, mi))
        return np.full(prob.items[name].model_grid.size, float(mi))

    plan, pinned = metals_plan.fast_metal_plan(prob, dict(prob.params), metal_xi)
    assert set(pinned) == {f'beta_{m}' for m in ('SiII(1190)', 'SiII(1193)', 'SiIII(1207)', 'SiII(1260)')}
    auto, cross = plan['lyalya_lyalya'], plan['lyalya_qso']
    pairs = prob.items['lyalya_lyalya'].metals
    kinds = [k for k, _ in auto]
    assert kinds.count('pipeline') == 1 and kinds.count('share') == 3 and kinds.count('static') == 11
    for (kind, arg), pair in zip(auto, pairs):
        assert (kind == 'static') == (not pair.cross_with_main)
        if kind == 'share':
            assert auto[arg][0] == 'pipeline' and pairs[arg].cross_with_main
    assert [k for k, _ in cross] == ['pipeline', 'share', 'share', 'share']
    assert len(calls) == 11
    # a different beta separates a pair from its leader
    pars = dict(prob.params)
    pars['beta_SiII(1193)'] = 0.7
    kinds = [k for k, _ in metals_plan.fast_metal_plan(prob, pars, metal_xi)[0]['lyalya_qso']]
    assert kinds.count('pipeline') == 2
    # equal values but one of the betas is sampled: no sharing with that pair
    prob2 = copy.deepcopy(prob)
    prob2.sample_params['limits']['beta_SiII(1193)'] = (0., 5.)
    kinds = [k for k, _ in metals_plan.fast_metal_plan(prob2, dict(prob2.params), metal_xi)[0]['lyalya_qso']]
    assert kinds.count('pipeline') == 2 and kinds.count('share') == 2


def test_monte_carlo_tables_round_trip_in_the_reference_layout(tmp_path):
    """vega_amd.output.write_monte_carlo: the HDUs / columns of reference vega/output.py:442-520, read back with
    the FITS reader that parses the reference's own files."""
    from types import SimpleNamespace
    from vega_amd import fitslite, output
    rng = np.random.default_rng(0)
    n_mocks, names = 5, ['ap', 'at', 'bias_eta_LYA']
    values, errors = rng.normal(size=(n_mocks, 3)), rng.uniform(0.1, 1, size=(n_mocks, 3))
    analysis = SimpleNamespace(
        mc_mocks={'lyalya_lyalya': rng.normal(size=(n_mocks, 40)), 'lyalya_qso': rng.normal(size=(n_mocks, 60))},
        mc_bestfits={n: np.stack([values[:, j], errors[:, j]], axis=1) for j, n in enumerate(names)},
        mc_covariances=[rng.normal(size=(3, 3)) for _ in range(n_mocks)],
        mc_chisq=list(rng.uniform(90, 110, n_mocks)), mc_valid_minima=[True, True, False, True, True],
        mc_valid_hesse=[True] * n_mocks, mc_failed_mask=[False, False, True, False, False])
    path = output.write_monte_carlo(analysis, tmp_path / 'monte_carlo', cpu_id=3)
    assert path.name == 'monte_carlo_3.fits' and path.stat().st_size % 2880 == 0
    hdul = fitslite.open(path)
    assert [h.header.get('EXTNAME') for h in hdul[1:]] == ['BESTFIT', 'FITINFO', 'MOCKS']      # (upper case, as astropy stores hdu.name)
    best = hdul[1].data
    assert [s.strip() for s in best['names']] == names
    np.testing.assert_array_equal(best['values'], values.T)
    np.testing.assert_array_equal(best['errors'], errors.T)
    cov = np.array(analysis.mc_covariances).reshape(n_mocks * 3, 3).T
    np.testing.assert_array_equal(best['covariance'], cov)
    info = hdul[2].data
    np.testing.assert_array_equal(info['chisq'], analysis.mc_chisq)
    assert list(info['valid_minima']) == [True, True, False, True, True] and list(info['failed_mask']) == [False, False, True, False, False]
    np.testing.assert_array_equal(hdul[3].data['lyalya_qso'], analysis.mc_mocks['lyalya_qso'])
    with pytest.raises(OSError):
        output.write_monte_carlo(analysis, tmp_path / 'monte_carlo', cpu_id=3)


def test_file_cosmology_and_new_bias_evolution(tmp_path):
    """A data file that carries a picca cosmology (OMEGAM ... header keywords, reference vega/data.py:360-366) is
    accepted; with `new-bias-evolution = True` the cross-correlation pipelines get one redshift grid per tracer
    (reference correlation_func.py:238-274), the auto-correlation keeps the mean redshift."""
    import re
    from vega_amd import synthetic
    from vega_amd.setup import build_problem, picca_dist_hubble
    from vega_amd.tables import read_tables
    cfg = tmp_path / 'configs' / 'cosmo'
    cfg.mkdir(parents=True)
    main = (GOLDEN / 'configs' / 'joint' / 'main.ini').read_text()
    main = re.sub(r'ini files = .*', 'ini files = configs/cosmo/lyalya_lyalya.ini configs/cosmo/lyalya_qso.ini', main)
    (cfg / 'main.ini').write_text(main)
    header = {'OMEGAM': 0.3147, 'OMEGAR': 7.9e-5}
    for name, bundle in (('lyalya_lyalya', 'cf_lya-exp.npz'), ('lyalya_qso', 'xcf_lya-exp.npz')):
        text = (GOLDEN / 'configs' / 'joint' / f'{name}.ini').read_text()
        path = synthetic.write_data_file(tmp_path / f'{name}.fits', read_tables(GOLDEN / 'inputs' / bundle),
                                         with_distortion=False, with_covariance=False, extra_header=header)
        text = re.sub(r'filename = .*', f'filename = {path}', text, count=1)
        (cfg / f'{name}.ini').write_text(text.replace('[model]', '[model]\nnew-bias-evolution = True'))
    prob = build_problem('configs/cosmo/main.ini', search_dirs=[tmp_path, GOLDEN])
    auto, cross = prob.items['lyalya_lyalya'].core, prob.items['lyalya_qso'].core
    assert auto.rel_z_evol_1 is None and cross.rel_z_evol_1 is not None
    cosmo = {'Omega_m': 0.3147, 'Omega_k': 0., 'Omega_r': 7.9e-5, 'wl': -1.}
    shift = cross.r * cross.mu / (2 * picca_dist_hubble(cross.z, cosmo))
    forest = (1 + cross.z + shift) / (1 + prob.z_eff)
    quasar = (1 + cross.z - shift) / (1 + prob.z_eff)
    assert cross.tracer1.type == 'continuous' and cross.tracer2.type == 'discrete'
    np.testing.assert_allclose(cross.rel_z_evol_1, forest, rtol=1e-15)
    np.testing.assert_allclose(cross.rel_z_evol_2, quasar, rtol=1e-15)
    assert np.abs(shift).max() > 0.05       # the split is not a rounding effect
