"""GPU parity of every model option the engine accepts, against the CPU oracle on the same inputs.

The golden fixtures cover the reference's production-like configuration; the variants below switch each
remaining option of SURVEY.md section 8a on (HCD / non-linear / smoothing / velocity-dispersion kinds, HeII,
damping, bias-evolution models, scale parametrisations, broadband kinds, metal options and matrices,
alternative (bias, beta) specifications) and compare xi and chi2 with the oracle at the fiducial point and at
seeded walkers.
"""
import copy

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

XI_RTOL = 1e-8
CHI2_RTOL = 1e-6


def _fresh(tag):
    from vega_amd.setup import build_problem
    return build_problem(f'configs/{tag}/main.ini', search_dirs=[GOLDEN])


def _check(prob, n_walkers=3, vary=None, seed=5):
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface, synthetic
    vega = VegaInterface(None, problem=prob, max_batch=max(n_walkers, 1))
    eng = vega.engine
    theta = synthetic.walkers(eng.low.theta0, eng.names, n_walkers, varied=vary, seed=seed)
    theta = np.vstack([eng.low.theta0[None, :], theta])
    chi2, status = vega.chi2_batch(theta, return_status=True)
    models = vega.compute_model_batch(theta)
    assert not status.any()
    for i in range(theta.shape[0]):
        pars = dict(zip(eng.names, theta[i]))
        ref_model = oc.compute_model(prob, pars)
        for name in prob.items:
            scale = np.abs(ref_model[name]).max()
            err = np.abs(models[name][i] - ref_model[name]).max() / scale
            assert err <= XI_RTOL, f'walker {i} {name}: scaled error {err:.3e}'
        assert chi2[i] == pytest.approx(oc.chi2(prob, pars), rel=CHI2_RTOL)
    vega.close()


def test_sinc_hcd_and_mcdonald():
    prob = _fresh('auto')
    pk = prob.items['lyalya_lyalya'].core.pk
    pk.hcd_model, pk.small_scale_nl, pk.uvb = 'sinc', 'mcdonald', False
    prob.params['L0_sinc'] = 7.0
    _check(prob)


def test_fvoigt_hcd():
    from vega_amd.setup import load_fvoigt_table
    prob = _fresh('joint')
    table = load_fvoigt_table('fvoigt_models/Fvoigt_exp.txt', [GOLDEN / 'inputs'])
    for item in prob.items.values():
        item.core.pk.hcd_model = 'fvoigt'
        item.core.pk.fvoigt_table = table
    prob.params['L0_fvoigt'] = 0.8
    _check(prob, n_walkers=2)


def test_relativistic_and_asymmetry_odd_multipoles():
    prob = _fresh('joint')
    cross = prob.items['lyalya_qso'].core
    cross.xi.relativistic = cross.xi.asymmetry = True
    prob.params.update({'Arel1': -13.5, 'Arel3': 1.0, 'Aasy0': 1.0, 'Aasy2': 1.0, 'Aasy3': 1.0})
    _check(prob, n_walkers=2)


def test_exp_smoothing_gauss_velocity_dispersion_croom():
    prob = _fresh('joint')
    cross = prob.items['lyalya_qso'].core
    cross.pk.fullshape_smoothing = 'exp'
    cross.pk.velocity_dispersion = 'gauss'
    cross.xi.evol_model['QSO'] = 'croom'
    prob.params.update({'par_exp_smooth': 1.5, 'per_exp_smooth': 2.5, 'sigma_velo_disp_gauss_QSO': 4.0,
                        'croom_par0': 0.53, 'croom_par1': 0.289})
    _check(prob)


def test_heii_damping_skip_nl_in_peak_ell4():
    prob = _fresh('auto')
    core = prob.items['lyalya_lyalya'].core
    core.pk.heii, core.pk.damping_scale, core.pk.damping_power = True, 0.05, 2
    core.pk.skip_nl_in_peak = True
    core.xi.ell_max = 4
    prob.params.update({'bias_gamma_e': 0.05, 'lambda_HeII': 150.})
    _check(prob)


@pytest.mark.parametrize('mode', ['aiso_epsilon', 'phi_alpha_smooth', 'full_shape_alpha'])
def test_scale_parametrisations(mode):
    prob = _fresh('joint')
    if mode == 'aiso_epsilon':
        prob.scale.parametrisation = 'aiso_epsilon'
        prob.params.update({'aiso': 1.02, 'epsilon': 0.03})
    elif mode == 'phi_alpha_smooth':
        prob.scale.parametrisation = 'phi_alpha'
        prob.scale.smooth_scaling = True
        prob.params.update({'phi': 1.03, 'alpha': 0.98, 'phi_smooth': 0.97, 'alpha_smooth': 1.01})
    else:
        prob.scale.parametrisation = 'ap_at'
        prob.scale.full_shape = prob.scale.full_shape_alpha = True
        prob.params.update({'ap_full': 1.04, 'at_full': 0.97})
    _check(prob, n_walkers=2)


def test_per_tracer_smoothing_and_alternative_bias_specs():
    prob = _fresh('joint')
    for key in ('par_sigma_smooth', 'per_sigma_smooth'):
        del prob.params[key]
    prob.params.update({'par_sigma_smooth_LYA': 2.2, 'per_sigma_smooth_LYA': 2.8,
                        'par_sigma_smooth_QSO': 1.1, 'per_sigma_smooth_QSO': 3.3})
    # (bias, beta) for LYA and (bias, bias_eta) for QSO instead of (bias_eta, beta)
    growth = prob.params['growth_rate']
    prob.params['bias_LYA'] = prob.params.pop('bias_eta_LYA') * growth / prob.params['beta_LYA']
    prob.params['bias_QSO'] = prob.params['bias_eta_QSO'] * growth / prob.params.pop('beta_QSO')
    # only sigmaNL_par given: the transverse one follows from the growth rate
    del prob.params['sigmaNL_per']
    _check(prob)


def test_broadband_kinds():
    from vega_amd.setup import BroadbandTerm
    prob = _fresh('auto')
    item = prob.items['lyalya_lyalya']
    name = item.name
    item.broadband = [
        BroadbandTerm(f'BB-{name}-0 mul pre rp,rt', 'broadband', 'mul', 'pre', 'rp,rt', (0, 1, 1), (0, 2, 2)),
        BroadbandTerm(f'BB-{name}-1 add pre r,mu', 'broadband', 'add', 'pre', 'r,mu', (-2, 0, 1), (0, 2, 2)),
        BroadbandTerm(f'BB-{name}-2 mul post r,mu', 'broadband', 'mul', 'post', 'r,mu', (0, 1, 1), (0, 0, 1)),
        BroadbandTerm(f'BB-{name}-3-broadband_sky', 'broadband_sky', 'add', 'post', 'rp,rt', (0, 0, 1), (0, 0, 1)),
    ]
    rng = np.random.default_rng(2)
    for term in item.broadband[:3]:
        for i in range(term.r1[0], term.r1[1] + 1, term.r1[2]):
            for j in range(term.r2[0], term.r2[1] + 1, term.r2[2]):
                prob.params[f'{term.name} ({i},{j})'] = 1e-3 * rng.standard_normal()
    prob.params[f'BB-{name}-3-broadband_sky-scale-sky'] = 0.009
    prob.params[f'BB-{name}-3-broadband_sky-sigma-sky'] = 30.
    _check(prob)


def test_metal_options_and_dense_metal_matrices():
    from scipy import sparse
    prob = _fresh('joint_metals')
    rng = np.random.default_rng(9)
    for item in prob.items.values():
        item.metal_opts['single_metal_beta'] = True
        item.metal_opts['separate_metal_auto_biases'] = True
        for pair in item.metals:
            n_out, n_in = item.model_grid.size, pair.pipeline.r.size
            dense = np.eye(n_out, n_in) * 0.8
            rows = rng.integers(0, n_out, 4000)
            cols = rng.integers(0, n_in, 4000)
            dense[rows, cols] += 0.02 * rng.standard_normal(4000)
            pair.matrix = sparse.csr_array(dense)
            if not pair.cross_with_main and pair.names[0] != pair.names[1]:
                prob.params[pair.auto_bias_names[0]] = 1.0 + 0.1 * rng.standard_normal()
    prob.params['beta_metals'] = 0.6
    _check(prob, n_walkers=1)


def test_metals_with_biases_inside_pk_and_hcd_division():
    prob = _fresh('auto_metals')
    item = prob.items['lyalya_lyalya']
    item.metal_opts['fast_metal_bias'] = False
    _check(copy.deepcopy(prob), n_walkers=1)
    # fast-metal-bias Kaiser term with an HCD model in the metal P(k): the effective-beta division
    item.metal_opts['fast_metal_bias'] = True
    for pair in item.metals:
        pair.pipeline.pk.hcd_model = 'Rogers'
        pair.pipeline.pk.uvb = True
    _check(prob, n_walkers=1)


def test_uv_shotnoise_and_instrumental_systematics():
    """Against the oracle and against the reference's own output (tests/golden/expected_extras.npz)."""
    from vega_amd import VegaInterface
    prob = _fresh('auto_extras')
    exp = np.load(GOLDEN / 'expected_extras.npz')
    vega = VegaInterface(None, problem=prob, max_batch=2)
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    assert vega.chi2(pars) == pytest.approx(float(exp['walker0/chi2']), rel=CHI2_RTOL)
    got = vega.compute_model(pars)['lyalya_lyalya']
    assert np.abs(got - exp['walker0/model']).max() <= XI_RTOL * np.abs(exp['walker0/model']).max()
    vega.close()
    _check(prob, n_walkers=2)


def test_fast_metals_frozen_caches_match_reference():
    """`fast_metals = True`: the reference's own sequence (chi2 at the fiducial point first - which freezes the
    metal x metal terms - then walkers) through the engine, against the reference's outputs.  After the freeze the
    engine runs 6 pipelines instead of 23: 4 core + one main x metal pipeline per item (equal betas), and the 11
    metal x metal terms are static vectors."""
    from vega_amd import VegaInterface
    from conftest import load_problem
    exp = np.load(GOLDEN / 'expected_joint_metals_fast.npz')
    vega = VegaInterface(None, problem=load_problem('joint_metals_fast'), max_batch=4)
    assert len(vega.engine.pipe_index) == 23
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    assert len(vega.engine.pipe_index) == 6
    model = vega.compute_model()
    for name in vega.corr_items:
        ref = exp[f'fid/model/{name}']
        assert np.abs(model[name] - ref).max() <= XI_RTOL * np.abs(ref).max()
    names = [str(n) for n in exp['param_names']]
    theta = np.stack([vega.engine.theta_from_params(dict(zip(names, row))) for row in exp['theta']])
    chi2, status = vega.chi2_batch(theta, return_status=True)
    assert not status.any()
    np.testing.assert_allclose(chi2, exp['chi2'], rtol=CHI2_RTOL)
    models = vega.compute_model_batch(theta)
    for i in range(theta.shape[0]):
        for name in vega.corr_items:
            ref = exp[f'walker{i}/model/{name}']
            assert np.abs(models[name][i] - ref).max() <= XI_RTOL * np.abs(ref).max(), (i, name)
    # an unsampled beta that made two pairs share a pipeline may not move afterwards
    with pytest.raises(ValueError, match='frozen metal terms'):
        vega.chi2({'beta_SiII(1193)': 0.7})
    vega.close()


@pytest.mark.parametrize('tag', ['joint_metals', 'auto_metals'])
def test_static_metal_basis_is_exact(tag):
    """freeze_static_metals(): polynomial metal pairs become xi = Y0 + (b1 + b2) Y1 + b1 b2 Y2 with static vectors
    (metal matrix included).  Same chi2 / models as the per-walker pipelines and as the oracle, for walkers that
    move every sampled-type parameter including the metal biases and betas."""
    from scipy import sparse
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface, synthetic
    prob = _fresh(tag)
    rng = np.random.default_rng(3)
    for item in prob.items.values():          # dense-ish metal matrices: the static form must carry them
        for pair in item.metals[::3]:
            n_out, n_in = item.model_grid.size, pair.pipeline.r.size
            dense = np.eye(n_out, n_in) * 0.9
            dense[rng.integers(0, n_out, 3000), rng.integers(0, n_in, 3000)] += 0.03 * rng.standard_normal(3000)
            pair.matrix = sparse.csr_array(dense)
    vega = VegaInterface(None, problem=prob, max_batch=8)
    eng = vega.engine
    n_before = len(eng.pipe_index)
    varied = [n for n in eng.names if n.startswith(('bias_eta_', 'beta_')) or n in ('ap', 'at', 'bao_amp', 'bias_hcd')]
    theta = synthetic.walkers(eng.low.theta0, eng.names, 6, varied=varied, seed=8)
    before_chi2, _, before_model = eng.eval(theta, want_model=True)
    vega.freeze_static_metals()
    eng = vega.engine
    n_metal_pipes = sum(1 for item in prob.items.values() for _ in item.metals)
    assert len(eng.pipe_index) < n_before and len(eng.pipe_index) >= n_before - n_metal_pipes
    chi2, status, model = eng.eval(theta, want_model=True)
    assert not status.any()
    np.testing.assert_allclose(chi2, before_chi2, rtol=1e-9)
    assert np.abs(model - before_model).max() <= 1e-11 * np.abs(before_model).max()
    for i in (0, 5):
        pars = dict(zip(eng.names, theta[i]))
        assert chi2[i] == pytest.approx(oc.chi2(prob, pars), rel=CHI2_RTOL)
        ref = oc.compute_model(prob, pars)
        for name, sl in eng.model_slices.items():
            assert np.abs(model[i, sl] - ref[name]).max() <= XI_RTOL * np.abs(ref[name]).max()
    with pytest.raises(ValueError, match='frozen metal terms'):
        vega.chi2({'alpha_SiII(1190)': 1.3})
    vega.close()


def test_static_metal_basis_after_fast_metals():
    """fast_metals + freeze_static_metals(): the merged main x metal pairs (leader and the pairs reading its pipeline)
    also become static Kaiser bases; results equal those of the frozen-cache engine before the step."""
    from vega_amd import VegaInterface, synthetic
    from conftest import load_problem
    vega = VegaInterface(None, problem=load_problem('joint_metals_fast'), max_batch=8)
    vega.freeze_metals()
    eng = vega.engine
    assert len(eng.pipe_index) == 6
    varied = [n for n in eng.names if n.startswith('bias_eta_') or n in ('beta_LYA', 'beta_QSO', 'ap', 'at', 'bao_amp')]
    theta = synthetic.walkers(eng.low.theta0, eng.names, 6, varied=varied, seed=12)
    before_chi2, _, before_model = eng.eval(theta, want_model=True)
    vega.freeze_static_metals()
    eng = vega.engine
    assert len(eng.pipe_index) == 5          # 4 core + QSO x metal (Lorentzian velocity dispersion: not polynomial)
    chi2, status, model = eng.eval(theta, want_model=True)
    assert not status.any()
    np.testing.assert_allclose(chi2, before_chi2, rtol=1e-9)
    assert np.abs(model - before_model).max() <= 1e-11 * np.abs(before_model).max()
    vega.close()


def test_mock_binning():
    """`mock-bin-size` (reference power_spectrum.py:143-160) folded into the static G table: against the reference's
    own output, and the other line-of-sight options against the oracle."""
    from vega_amd import VegaInterface
    prob = _fresh('auto_mockbin')
    exp = np.load(GOLDEN / 'expected_mockbin.npz')
    vega = VegaInterface(None, problem=prob, max_batch=2)
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    assert vega.chi2(pars) == pytest.approx(float(exp['walker0/chi2']), rel=CHI2_RTOL)
    got = vega.compute_model(pars)['lyalya_lyalya']
    assert np.abs(got - exp['walker0/model']).max() <= XI_RTOL * np.abs(exp['walker0/model']).max()
    vega.close()
    for los in (None, 'only-los'):
        prob = _fresh('auto_mockbin')
        prob.items['lyalya_lyalya'].core.pk.mock_los_smoothing = los
        _check(prob, n_walkers=1)
    # `amplitude`: the line-of-sight bin follows `los_smooth_amp` - static while that parameter is not sampled
    prob = _fresh('auto_mockbin')
    prob.items['lyalya_lyalya'].core.pk.mock_los_smoothing = 'amplitude'
    prob.params['los_smooth_amp'] = 0.4
    _check(prob, n_walkers=1, vary=['bias_eta_LYA', 'beta_LYA', 'ap', 'at', 'bias_hcd'])


@pytest.mark.parametrize('mode', ['amplitude', 'growth'])
def test_mock_binning_that_follows_a_sampled_parameter(mode):
    """`mock-los-smoothing = amplitude | growth` with a SAMPLED `los_smooth_amp` / `growth_rate` (reference
    power_spectrum.py:143-160): the line-of-sight factor sinc(k_par L (1 + p) / 2) is the walker's own - a factor of the mu
    loop (vmx_pipe_desc::mock_los_slot), the static table keeping the transverse one.  Against the unmodified reference,
    four walkers with four values of the parameter in one batch."""
    from vega_amd import VegaInterface
    prob = _fresh('auto_mockbin')
    name = 'los_smooth_amp' if mode == 'amplitude' else 'growth_rate'
    prob.items['lyalya_lyalya'].core.pk.mock_los_smoothing = mode
    if mode == 'amplitude':
        prob.params['los_smooth_amp'] = 0.3
    prob.sample_params['limits'][name] = (0., 2.)
    exp = np.load(GOLDEN / 'expected_mockbin_sampled.npz')
    vega = VegaInterface(None, problem=prob, max_batch=4)
    names = [str(n) for n in exp[f'{mode}/param_names']]
    theta = np.stack([vega.engine.theta_from_params(dict(zip(names, map(float, row)))) for row in exp[f'{mode}/theta']])
    chi2, status = vega.chi2_batch(theta, return_status=True)
    assert not status.any()
    np.testing.assert_allclose(chi2, exp[f'{mode}/chi2'], rtol=CHI2_RTOL)
    models = vega.compute_model_batch(theta)['lyalya_lyalya']
    for i in range(4):
        ref = exp[f'{mode}/model'][i]
        assert np.abs(models[i] - ref).max() <= XI_RTOL * np.abs(ref).max(), i
    # the scalar entry (one walker, its own batch) agrees with the batch
    assert vega.chi2(dict(zip(names, map(float, exp[f'{mode}/theta'][2])))) == pytest.approx(chi2[2], rel=1e-12)
    vega.close()


def test_metal_decomposition():
    """`no-metal-decomp = False` (reference model.py:120-123, :181-186): every metal pair enters twice - its
    smooth-spectrum pipeline, and its peak-spectrum pipeline (with the peak's broadening) times bao_amp.  Against
    the unmodified reference."""
    from vega_amd import VegaInterface
    prob = _fresh('auto_metals')
    prob.items['lyalya_lyalya'].metal_opts['no_metal_decomp'] = False
    exp = np.load(GOLDEN / 'expected_metal_decomp.npz')
    vega = VegaInterface(None, problem=prob, max_batch=2)
    assert len(vega.engine.pipe_index) == 2 + 15        # (+ 15 peak pipelines that are not indexed)
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    got = vega.compute_model()['lyalya_lyalya']
    assert np.abs(got - exp['fid/model']).max() <= XI_RTOL * np.abs(exp['fid/model']).max()
    names = [str(n) for n in exp['param_names']]
    theta = np.stack([vega._theta(dict(zip(names, row))) for row in exp['theta']])
    np.testing.assert_allclose(vega.chi2_batch(theta), exp['chi2'], rtol=CHI2_RTOL)
    got = vega.compute_model(dict(zip(names, exp['theta'][0])))['lyalya_lyalya']
    assert np.abs(got - exp['walker0/model']).max() <= XI_RTOL * np.abs(exp['walker0/model']).max()
    vega.freeze_static_metals()                         # nothing is eligible: the engine stays as it is
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    vega.close()


def test_direct_pk_with_metal_terms():
    """`direct_pk` + `no-metal-decomp = False` (reference model.py:188-207 -> :120-123): the metal terms are part of the direct
    model, computed on the caller's spectrum - the smooth-spectrum entries of the pairs (`vmx_metal_desc::in_direct`), on their
    per-walker pipelines (the interface rebuilds the engine without the static basis of the template's spectra at the first
    such call).  Against the unmodified reference; ordinary evaluations before and after are untouched."""
    from vega_amd import VegaInterface
    prob = _fresh('auto_metals')
    prob.items['lyalya_lyalya'].metal_opts['no_metal_decomp'] = False
    exp = np.load(GOLDEN / 'expected_direct_pk_metals.npz')
    pk = exp['direct_pk']
    vega = VegaInterface(None, problem=prob, max_batch=4)
    assert vega.chi2() == pytest.approx(float(exp['plain/chi2']), rel=CHI2_RTOL)
    assert vega.engine.static_poly
    assert vega.chi2(direct_pk=pk) == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    assert not vega.engine.static_poly                  # the engine was rebuilt for the caller's spectrum
    got = vega.compute_model(direct_pk=pk)['lyalya_lyalya']
    assert np.abs(got - exp['fid/model']).max() <= XI_RTOL * np.abs(exp['fid/model']).max()
    names = [str(n) for n in exp['param_names']]
    for i, row in enumerate(exp['theta']):
        assert vega.chi2(dict(zip(names, row)), direct_pk=pk * (1 + 0.01 * (i + 1))) == pytest.approx(float(exp['chi2'][i]), rel=CHI2_RTOL)
    assert vega.chi2() == pytest.approx(float(exp['plain/chi2']), rel=CHI2_RTOL)
    vega.close()


@pytest.mark.parametrize('tag', ['cross', 'auto_rp'])
def test_new_metals_through_the_engine(tmp_path, tag):
    """`new_metals = True`: metal matrices built at set-up (vega_amd/metal_matrices.py; reference
    vega/metals.py:389-752) and applied by the engine like matrices read from a file - chi2 and model against the
    unmodified reference on the same stacked-delta file / catalogue."""
    from conftest import new_metals_problem
    from vega_amd import VegaInterface
    prob, name = new_metals_problem(tmp_path, tag)
    exp = np.load(GOLDEN / 'expected_new_metals.npz')
    pars = {str(n): float(v) for n, v in zip(exp[f'{tag}/param_names'], exp[f'{tag}/theta'][0])}
    # the matrices are Kronecker products: applied as two small products per walker (default) or uploaded dense
    for kron in (True, False):
        vega = VegaInterface(None, problem=prob, max_batch=12, kron_metals=kron)
        assert vega.chi2() == pytest.approx(float(exp[f'{tag}/chi2']), rel=CHI2_RTOL)
        got = vega.compute_model()[name]
        assert np.abs(got - exp[f'{tag}/model']).max() <= XI_RTOL * np.abs(exp[f'{tag}/model']).max()
        assert vega.chi2(pars) == pytest.approx(float(exp[f'{tag}/walker0/chi2']), rel=CHI2_RTOL)
        # a batch large enough for the grouped (MFMA) path
        batch = vega.chi2_batch([pars] * 11 + [None])
        np.testing.assert_allclose(batch[:11], float(exp[f'{tag}/walker0/chi2']), rtol=CHI2_RTOL)
        assert batch[11] == pytest.approx(float(exp[f'{tag}/chi2']), rel=CHI2_RTOL)
        vega.close()


def test_single_multipole():
    """`single_multipole = ell`: the model is xi_ell(r') alone, without its Legendre factor (reference pktoxi.py:122-155)."""
    for ell in (0, 2, 4):
        prob = _fresh('auto')
        prob.items['lyalya_lyalya'].core.xi.single_multipole = ell
        _check(prob, n_walkers=1)


def test_new_bias_evolution_against_the_reference(tmp_path):
    """`new-bias-evolution` with the cosmology of the data file: engine against the unmodified reference."""
    from conftest import new_bias_evol_problem
    from vega_amd import VegaInterface
    exp = np.load(GOLDEN / 'expected_new_bias_evol.npz')
    vega = VegaInterface(None, problem=new_bias_evol_problem(tmp_path), max_batch=2)
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    got = vega.compute_model()['lyalya_qso']
    assert np.abs(got - exp['fid/model']).max() <= XI_RTOL * np.abs(exp['fid/model']).max()
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    assert vega.chi2(pars) == pytest.approx(float(exp['walker0/chi2']), rel=CHI2_RTOL)
    vega.close()


def test_new_bias_evolution_with_a_file_cosmology():
    """`new-bias-evolution` (reference correlation_func.py:238-299): the QSO and the forest of a cross-correlation
    evolve with z -/+ rp / (2 D_H(z)), D_H from the picca cosmology of the data file's header (restated, picca being
    absent); engine against oracle here, the reference itself in the test above)."""
    from vega_amd.setup import picca_dist_hubble
    prob = _fresh('joint')
    cosmo = {'Omega_m': 0.315, 'Omega_k': 0., 'Omega_r': 7.9e-5, 'wl': -1.}
    # D_H of a flat LCDM cosmology at z = 2.3, Mpc/h
    e_z = np.sqrt(0.315 * 3.3**3 + 7.9e-5 * 3.3**4 + (1 - 0.315 - 7.9e-5))
    assert picca_dist_hubble(np.array([2.3]), cosmo)[0] == pytest.approx(2997.92458 / e_z, rel=1e-6)
    cross = prob.items['lyalya_qso'].core
    cross.xi.new_bias_evol = True
    z, z_eff = cross.z, prob.z_eff
    shift = cross.r * cross.mu / (2 * picca_dist_hubble(z, cosmo))
    rel_q, rel_f = (1 + z - shift) / (1 + z_eff), (1 + z + shift) / (1 + z_eff)
    assert cross.tracer2.type == 'discrete'
    cross.rel_z_evol_1, cross.rel_z_evol_2 = rel_f, rel_q
    prob.params['alpha_QSO'] = 1.44      # different exponents make the split visible
    _check(prob, n_walkers=2)


def _extrap_problem():
    """The auto-correlation with `fht_extrap = True` on a model whose spectra keep non-zero end samples (no small-scale
    non-linear term, no full-shape smoothing, no binning kernel; the walkers carry sigmaNL = 0) - tests/golden/make_golden.py:
    dump_fht_extrap."""
    prob = _fresh('auto')
    core = prob.items['lyalya_lyalya'].core
    core.pk.small_scale_nl, core.pk.fullshape_smoothing, core.pk.use_gk = None, None, False
    core.xi.fht_extrap = True
    return prob


def test_fht_extrap_against_the_reference():
    """`fht_extrap = True` (reference vega/pktoxi.py:41,141): power-law pads of the FFTLog input, formed per walker on the
    device (k_pk_extrap) and multiplied by the padded operator's columns."""
    from vega_amd import VegaInterface
    from vega_amd import fftlog_op
    exp = np.load(GOLDEN / 'expected_fht_extrap.npz')
    prob = _extrap_problem()
    vega = VegaInterface(None, problem=prob, max_batch=4)
    eng = vega.engine
    assert eng.fht_extrap and eng.fftlog_pads == (617, 617)
    names = [str(n) for n in exp['param_names']]
    theta = np.stack([eng.theta_from_params(dict(zip(names, map(float, row)))) for row in exp['theta']])
    chi2, status = vega.chi2_batch(theta, return_status=True)
    assert not status.any()
    np.testing.assert_allclose(chi2, exp['chi2'], rtol=CHI2_RTOL)
    models = vega.compute_model_batch(theta)['lyalya_lyalya']
    for i in range(3):
        assert np.abs(models[i] - exp['model'][i]).max() <= XI_RTOL * np.abs(exp['model'][i]).max(), i
    # On the radii of this fit the pads are worth 1e-9 of the model (the template reaches 1152 h/Mpc): the stage taps show
    # them at work.  P_ell rows: the samples, then the pads as numpy forms them; spline coefficients: the padded operator.
    k = np.asarray(prob.k)
    nk, (lo, hi) = k.size, eng.fftlog_pads
    nkp = (nk + lo + hi + 31) // 32 * 32
    n_cols = len(eng.pk_multipoles(B=3))
    rows = eng.debug_read(0, 0, 4 * 3 * n_cols * nkp).reshape(4, n_cols * 3, nkp)
    ncp = (nk + 2 + 31) // 32 * 32
    coef = eng.debug_read(2, 0, 4 * 3 * n_cols * ncp).reshape(4, n_cols * 3, ncp)
    moved = 0.0
    for e, ell in enumerate((0, 2, 4, 6)):
        op = fftlog_op.xi_operator(k, ell, extrap=True)[0]
        op0 = fftlog_op.xi_operator(k, ell)[0]
        for c in range(n_cols * 3):
            f = rows[e, c, :nk]
            with np.errstate(all='ignore'):
                left = f[0] * (f[1] / f[0]) ** (-np.arange(1.0, lo + 1))
                right = f[-1] * (f[-1] / f[-2]) ** np.arange(1.0, hi + 1)
            np.testing.assert_allclose(rows[e, c, nk:nk + lo], left, rtol=1e-12, atol=0)
            np.testing.assert_allclose(rows[e, c, nk + lo:nk + lo + hi], right, rtol=1e-12, atol=0)
            want = op @ rows[e, c, :nk + lo + hi]
            got = coef[e, c, :nk + 2]
            window = got != 0.0             # (the product computes the coefficient rows the batch's bins can read)
            assert window.sum() > 100
            assert np.abs(got - want)[window].max() <= 1e-11 * np.abs(want[window]).max()
            moved = max(moved, np.abs(want - op0 @ f).max() / np.abs(want).max())
    assert moved > 1e-6            # (the pads do change the transform - at the small radii no bin of this fit reads)
    vega.close()
    # the test configuration's own model smooths its spectra to exact zeros at the last wavenumbers: 0 / 0 end segments,
    # NaN in the reference (recorded in the fixture) - here the non-finite status and the error value
    assert np.isnan(exp['default_model/chi2'])
    prob = _fresh('auto')
    prob.items['lyalya_lyalya'].core.xi.fht_extrap = True
    vega = VegaInterface(None, problem=prob, max_batch=1)
    chi2, status = vega.chi2_batch(vega.engine.low.theta0[None, :], return_status=True)
    assert status[0] != 0 and chi2[0] == 1e100
    vega.close()


def test_direct_pk_with_the_odd_multipole_terms():
    """`direct_pk` on a cross-correlation with the relativistic and asymmetry terms (reference model.py:188-207 ->
    correlation_func.py:491-551: the caller's spectrum is the terms' pk_lin): the Hamilton splines as operators of the
    spectrum (`vmx_pipeline_set_odd_operator`), the walkers' coefficient rows formed at `vmx_set_direct_pk`.  Against the
    oracle (as the odd-multipole terms themselves: no reference fixture holds them), two walkers with two spectra; the
    ordinary evaluation before and after is untouched."""
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface, synthetic
    prob = _fresh('joint')
    cross = prob.items['lyalya_qso'].core
    cross.xi.relativistic = cross.xi.asymmetry = True
    prob.params.update({'Arel1': -13.5, 'Arel3': 1.0, 'Aasy0': 1.0, 'Aasy2': 1.0, 'Aasy3': 1.0})
    vega = VegaInterface(None, problem=prob, max_batch=4)
    names = vega.engine.names
    theta = synthetic.walkers(vega.engine.low.theta0, names, 2, seed=61, varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA',
                                                                                  'beta_QSO', 'drp_QSO', 'Arel1', 'Aasy0', 'Aasy3'])
    plain = vega.chi2()
    assert plain == pytest.approx(oc.chi2(prob), rel=CHI2_RTOL)
    x = np.log(np.asarray(prob.k) / 0.05)
    for i in range(2):
        pars = dict(zip(names, theta[i]))
        pk = np.asarray(prob.pk_full) * (1 + 0.02 * (i + 1) * np.exp(-0.5 * x**2))     # a bump around k = 0.05 h/Mpc
        ref_model = oc.compute_model(prob, pars, direct_pk=pk)
        got = vega.compute_model(pars, direct_pk=pk)
        for name in prob.items:
            assert np.abs(got[name] - ref_model[name]).max() <= XI_RTOL * np.abs(ref_model[name]).max(), (i, name)
        assert vega.chi2(pars, direct_pk=pk) == pytest.approx(oc.chi2(prob, pars, direct_pk=pk), rel=CHI2_RTOL)
        # (the odd terms do depend on the spectrum: with the template's spectrum in their place the model moves)
        without = oc.compute_model(prob, pars, direct_pk=np.asarray(prob.pk_full))['lyalya_qso']
        assert np.abs(ref_model['lyalya_qso'] - without).max() > 1e-6 * np.abs(without).max()
    assert vega.chi2() == plain
    vega.close()
