"""GPU parity tests of round 2 (``-m gpu``), through the C ABI, against fixtures of the unmodified reference:

* BASELINE configs[1] as stated (auto only, ell <= 4, dense 2500^2 distortion matrix, one walker per call);
* the MFMA product kernels (B > 8: `k_gemm_nt44` for the distortion products, `k_gemm_nt` for the C^-1 products)
  directly against the reference fixture with dense distortion matrices and covariances;
* the marginalisation coefficients through ``chi2 / log_lik(return_marg_coeff=True)`` and
  ``compute_model(marg_coeff=...)`` - the call the reference's PolyChord adapter makes;
* ``vega.models[name].compute(pars, pk_full, pk_smooth)``;
* Monte-Carlo mocks and chi2 with a global covariance;
* the walker-sharding collective on a real engine with an RCCL (nccl) process group of one rank.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, config1_problem, load_problem, marginalization_problem, MARGINALIZATION_CASES, \
    synth_joint_problem

pytestmark = pytest.mark.gpu

CHI2_RTOL = 1e-6


def _pars(exp, prefix=''):
    return [{str(n): float(v) for n, v in zip(exp[prefix + 'param_names'], row)} for row in exp[prefix + 'theta']]


def _assert_xi(got, ref, mask, what):
    """north_star: xi <= 1e-8 relative - element-wise on the bins the fit uses (floor: 1e-12 of the vector's scale,
    for the zero crossings), and to the vector's scale on every bin."""
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-8 * scale, what
    np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-8, atol=1e-12 * scale, err_msg=what)


def test_config1_as_stated_single_walker_calls():
    """BASELINE configs[1]: Lya x Lya auto only, ell = 0, 2, 4, dense 2500^2 distortion matrix, batch = 1 - the
    latency-bound chain (fused single-walker distortion product) against the reference and the oracle."""
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface
    prob = config1_problem()
    mask = prob.items['lyalya_lyalya'].model_mask
    exp = np.load(GOLDEN / 'expected_config1.npz')
    vega = VegaInterface(None, problem=prob, max_batch=1)
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    assert vega.log_lik() == pytest.approx(float(exp['fid/log_lik']), rel=1e-9)
    _assert_xi(vega.compute_model()['lyalya_lyalya'], exp['fid/model/lyalya_lyalya'], mask, 'config1 fid')
    for i, pars in enumerate(_pars(exp)):
        assert vega.chi2(pars) == pytest.approx(float(exp['chi2'][i]), rel=CHI2_RTOL)
        _assert_xi(vega.compute_model(pars)['lyalya_lyalya'], exp[f'walker{i}/model/lyalya_lyalya'], mask, f'walker {i}')
    pars = {'ap': 0.97, 'at': 1.04, 'beta_LYA': 1.9, 'bias_hcd': -0.04}
    assert vega.chi2(pars) == pytest.approx(oc.chi2(prob, pars), rel=CHI2_RTOL)
    vega.close()
    # the same walkers as one batch of 4 (small-batch streaming kernels)
    vega = VegaInterface(None, problem=prob, max_batch=4)
    np.testing.assert_allclose(vega.chi2_batch(_pars(exp)), exp['chi2'], rtol=CHI2_RTOL)
    vega.close()


@pytest.mark.parametrize('batch', [9, 64, 256])
def test_mfma_products_against_the_reference_fixture(batch):
    """The 8 reference walkers of the dense-matrix fixture tiled to B > 8, so that the distortion products run on the
    four-block fp64 MFMA kernel (`k_gemm_nt44`, one grouped launch, split-K) and the C^-1 products on the 16x16x4
    kernel (`k_gemm_nt`): chi2 and every model vector directly against the reference's values."""
    from vega_amd import VegaInterface
    prob = synth_joint_problem()
    exp = np.load(GOLDEN / 'expected_joint_synth.npz')
    vega = VegaInterface(None, problem=prob, max_batch=batch)
    names = [str(n) for n in exp['param_names']]
    base = np.stack([vega.engine.theta_from_params(dict(zip(names, row))) for row in exp['theta']])
    reps = -(-batch // base.shape[0])
    theta = np.tile(base, (reps, 1))[:batch]
    which = np.tile(np.arange(base.shape[0]), reps)[:batch]
    chi2, status, model = vega.engine.eval(theta, want_model=True)
    assert not status.any()
    np.testing.assert_allclose(chi2, exp['chi2'][which], rtol=CHI2_RTOL)
    for b in range(batch):
        for name, sl in vega.engine.model_slices.items():
            _assert_xi(model[b, sl], exp[f'walker{which[b]}/model/{name}'], prob.items[name].model_mask,
                       f'B={batch} walker {b} {name}')
    vega.close()


@pytest.mark.parametrize('mode', ['cov', 'infit'])
def test_marginalisation_coefficients_through_the_public_surface(tmp_path, mode):
    from vega_amd import VegaInterface
    exp = np.load(GOLDEN / 'expected_marg_coeff.npz')
    prob = marginalization_problem(tmp_path, MARGINALIZATION_CASES['rtmax'], in_fit=mode == 'infit')
    vega = VegaInterface(None, problem=prob, max_batch=16)
    name = 'lyalya_lyalya'

    item = prob.items[name]

    def close(got, ref, pars=None):
        # against the reference's fixture: the reference builds the map as inv(A) . G (vega/data.py:762-828) with an
        # ill-conditioned A, so the matrix itself moves by ~1e-7 of the coefficients' scale between hosts (NumPy on
        # the REFERENCE's model with this host's matrix differs from the fixture by the same amount) ...
        np.testing.assert_allclose(got, ref, rtol=0, atol=5e-6 * np.abs(ref).max())
        # ... and tightly against the same map applied by NumPy to the engine's own model
        if pars is not False:
            model = vega.compute_model(pars)[name]
            want = item.marg_diff2coeff.dot(item.masked_data_vec - model[item.model_mask])
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-10 * np.abs(want).max())

    # a model error before anything was computed: (1e100, None), reference vega_interface.py:273-279
    assert vega.chi2({'ap': 1e3}, return_marg_coeff=True) == (1e100, None)
    chi2, coeff = vega.chi2(return_marg_coeff=True)
    assert chi2 == pytest.approx(float(exp[f'{mode}/fid/chi2']), rel=CHI2_RTOL)
    assert set(coeff) == {name}
    close(coeff[name], exp[f'{mode}/fid/coeff'])
    # ... and the public form on a model the caller holds (reference compute_marg_coeff, vega_interface.py:546-579)
    held = vega.compute_marg_coeff(vega.compute_model())
    assert set(held) == {name} and vega.corr_num_marg_modes == {name: vega.problem.items[name].num_marg_modes}
    np.testing.assert_allclose(held[name], coeff[name], rtol=0, atol=1e-10 * np.abs(coeff[name]).max())
    # the reference's PolyChord closure (vega/samplers/polychord.py:106-113)
    names = list(prob.sample_params['limits'])

    def log_lik(theta):
        params = {n: float(theta[i]) for i, n in enumerate(names)}
        return vega.log_lik(params, return_marg_coeff=True)

    ll, flat = log_lik([prob.params[n] for n in names])
    assert ll == pytest.approx(float(exp[f'{mode}/fid/log_lik']), rel=1e-8)
    close(flat, exp[f'{mode}/fid/coeff_flat'])
    pars = _pars(exp, f'{mode}/')[0]
    chi2, coeff_w = vega.chi2(pars, return_marg_coeff=True)
    assert chi2 == pytest.approx(float(exp[f'{mode}/walker0/chi2']), rel=CHI2_RTOL)
    close(coeff_w[name], exp[f'{mode}/walker0/coeff'], pars)
    # after a model error the first coefficients ever computed come back (reference :285-286, :276-277)
    bad_chi2, bad_coeff = vega.chi2({'ap': 1e3}, return_marg_coeff=True)
    assert bad_chi2 == 1e100 == float(exp[f'{mode}/bad/chi2'])
    close(bad_coeff[name], exp[f'{mode}/bad/coeff'], False)
    # compute_model(marg_coeff=...) adds the distorted templates (reference :243-246)
    model = vega.compute_model(marg_coeff=coeff)[name]
    plain = vega.compute_model()[name]
    assert np.abs(plain - exp[f'{mode}/fid/model_plain']).max() <= 1e-8 * np.abs(plain).max()
    own = plain + item.marg_templates.dot(coeff[name])
    assert np.abs(model - own).max() <= 1e-12 * np.abs(own).max()
    ref = exp[f'{mode}/fid/model_with_templates']
    assert np.abs(model - ref).max() <= 5e-6 * np.abs(ref).max()       # (carries the coefficients' host dependence)
    # batched form: 12 walkers, two of them failing
    theta = np.stack([vega._theta(), vega._theta(pars)] * 6)
    theta[3, vega.engine.low.slot['ap']] = 1e3
    chi2_b, status, coeff_b = vega.chi2_batch(theta, return_status=True, return_marg_coeff=True)
    assert status[3] and chi2_b[3] == 1e100 and np.isnan(coeff_b[name][3]).all()
    close(coeff_b[name][0], exp[f'{mode}/fid/coeff'])
    close(coeff_b[name][11], exp[f'{mode}/walker0/coeff'], pars)
    vega.close()


def test_model_compute_per_item_entry():
    """reference vega/model.py:157-187 through ``vega.models[name].compute(pars, pk_full, pk_smooth)``."""
    import copy
    from vega_amd import VegaInterface
    exp = np.load(GOLDEN / 'expected_model_compute.npz')
    prob = load_problem('joint_metals')
    vega = VegaInterface(None, problem=prob, max_batch=2)
    walker = _pars(exp)[0]
    base = vega.chi2()
    for name, item in prob.items.items():
        pars = copy.deepcopy(vega.params)
        xi = vega.models[name].compute(pars, vega.fiducial['pk_full'], vega.fiducial['pk_smooth'])
        assert pars['peak'] is False
        _assert_xi(xi, exp[f'fiducial_spectra/{name}'], item.model_mask, name)
        xi = vega.models[name].compute(copy.deepcopy(walker), exp['pk_full'], exp['pk_smooth'])
        _assert_xi(xi, exp[f'own_spectra/{name}'], item.model_mask, name + ' own spectra')
    assert vega.chi2() == pytest.approx(base, rel=1e-14)        # the fiducial spectra are back
    vega.close()


def test_global_covariance_monte_carlo(tmp_path):
    """Mocks from the global covariance (reference vega/analysis.py:164-222) through the driver: the reference's
    vectors for seed 7, chi2 of the fiducial parameters against them through the monte_carlo switch (reference
    vega_interface.py:294-304) and through the per-walker mock index, then a few fits."""
    from vega_amd import VegaInterface
    exp = np.load(GOLDEN / 'expected_global_mc.npz')
    prob = synth_joint_problem(with_global_cov=True, tmp_path=tmp_path)
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA']
    prob.mc_config = {'params': {}, 'sample': {
        'limits': {'ap': (0.5, 1.5), 'at': (0.5, 1.5), 'bias_eta_LYA': (-2., 0.), 'beta_LYA': (0., 5.)},
        'values': {n: prob.params[n] for n in names},
        'errors': {'ap': 0.01, 'at': 0.01, 'bias_eta_LYA': 0.01, 'beta_LYA': 0.1},
        'fix': {n: False for n in names}}}
    vega = VegaInterface(None, problem=prob, max_batch=64)
    assert vega.chi2() == pytest.approx(float(exp['data/chi2']), rel=CHI2_RTOL)
    assert vega.log_lik() == pytest.approx(float(exp['data/log_lik']), rel=1e-8)
    assert vega.run_monte_carlo(num_mocks=2, seed=7, run_mc_fits=False) is None
    whole = vega.analysis.mc_mocks['global']
    for i in range(2):
        ref = exp[f'mock{i}/global']
        np.testing.assert_allclose(whole[i], ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    vega.monte_carlo = True
    for i in range(2):
        vega.analysis.current_mc_mock = whole[i]
        assert vega.chi2() == pytest.approx(float(exp[f'mock{i}/chi2_fid']), rel=CHI2_RTOL)
        assert vega.log_lik() == pytest.approx(float(exp[f'mock{i}/log_lik_fid']), rel=1e-8)
    vega.monte_carlo = False
    assert vega.chi2() == pytest.approx(float(exp['data/chi2']), rel=CHI2_RTOL)
    # fits: every mock against the global inverse covariance; [monte carlo] sampling table by default
    res = vega.run_monte_carlo(num_mocks=6, seed=7)
    assert res.names == names and res.is_valid.all()
    n_data = sum(it.data_size for it in prob.items.values())
    assert np.all(np.abs(res.fval - (n_data - len(names))) < 5 * np.sqrt(2 * n_data))
    np.testing.assert_allclose(vega.analysis.mc_mocks['global'][:2], whole, rtol=0, atol=1e-12)
    truth = np.array([prob.params[n] for n in names])
    assert np.abs((res.values - truth) / res.errors).max() < 5
    vega.close()


def test_fitting_existing_global_mocks(tmp_path):
    """`bin/run_vega_mc_fits_mpi.py` (:11-79, :127-163): mocks of the global masked data vector READ from a file (HDU MOCKS,
    column 'global'), optionally cut to two slices, fitted - here the very mocks `run_monte_carlo` draws for seed 7, written
    with the driver's own writer inside a longer vector: the fits must be those of the drawn mocks, fit for fit, and a rank's
    contiguous share must be its rows of them."""
    from vega_amd import VegaInterface, fitslite
    from vega_amd.montecarlo import contiguous_share, fit_mocks_sharded
    prob = synth_joint_problem(with_global_cov=True, tmp_path=tmp_path)
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA']
    prob.mc_config = {'params': {}, 'sample': {
        'limits': {'ap': (0.5, 1.5), 'at': (0.5, 1.5), 'bias_eta_LYA': (-2., 0.), 'beta_LYA': (0., 5.)},
        'values': {n: prob.params[n] for n in names},
        'errors': {'ap': 0.01, 'at': 0.01, 'bias_eta_LYA': 0.01, 'beta_LYA': 0.1},
        'fix': {n: False for n in names}}}
    vega = VegaInterface(None, problem=prob, max_batch=64)
    ref = vega.run_monte_carlo(num_mocks=5, seed=7)
    whole = np.array(vega.analysis.mc_mocks['global'])
    n1 = next(iter(prob.items.values())).data_size
    # the mocks inside a longer vector: [7 foreign | item 1 | 11 foreign | item 2 | 3 foreign], through a file
    rng = np.random.default_rng(0)
    longer = np.concatenate([rng.standard_normal((5, 7)), whole[:, :n1], rng.standard_normal((5, 11)), whole[:, n1:],
                             rng.standard_normal((5, 3))], axis=1)
    path = tmp_path / 'mocks.fits'
    fitslite.write_tables(str(path), [('MOCKS', [('global', f'{longer.shape[1]}D', longer)])])
    table = [h for h in fitslite.open(str(path))[1:] if h.header['EXTNAME'] == 'MOCKS'][0]
    mocks = np.asarray(table.data['global'])
    np.testing.assert_array_equal(mocks, longer)
    slices = (7, 7 + n1, 7 + n1 + 11, longer.shape[1] - 3)
    vega.monte_carlo = True
    res = vega.analysis.fit_global_mocks(mocks, *slices)
    np.testing.assert_array_equal(vega.analysis.mc_mocks['global'], whole)
    np.testing.assert_array_equal(res.values, ref.values)
    np.testing.assert_array_equal(res.fval, ref.fval)
    np.testing.assert_array_equal(res.is_valid, ref.is_valid)
    assert len(vega.analysis.mc_chisq) == 5 and vega.analysis.mc_bestfits['ap'].shape == (5, 2)
    with pytest.raises(ValueError):
        vega.analysis.fit_global_mocks(mocks)                  # (uncut: not this problem's vector)
    # rank 1 of 2: rows 3..4 (the first rank takes the remainder), its own result file
    assert [contiguous_share(5, 2, r) for r in range(2)] == [(0, 3), (3, 5)]
    mc, part, block = fit_mocks_sharded(vega, mocks, slices, rank=1, world_size=2, output_dir=tmp_path / 'mc')
    assert block == (3, 5) and (tmp_path / 'mc' / 'monte_carlo_1.fits').exists()
    np.testing.assert_allclose(part.values, ref.values[3:5], rtol=1e-9)
    np.testing.assert_allclose(part.fval, ref.fval[3:5], rtol=1e-9)
    vega.close()


def test_walker_sharding_on_a_real_engine_with_an_rccl_group_of_one():
    """`chi2_sharded` (one all_gather per batch) with backend "nccl" (= RCCL) and device tensors around a real engine;
    more ranks are covered on CPU with gloo (tests/test_parallel.py) and by the round driver's N-GPU bench."""
    import torch
    import torch.distributed as dist
    from vega_amd import VegaInterface, synthetic
    from vega_amd.parallel import chi2_sharded
    vega = VegaInterface(None, problem=load_problem('joint'), max_batch=32)
    eng = vega.engine
    theta = synthetic.walkers(eng.low.theta0, eng.names, 45, varied=['ap', 'at', 'beta_LYA', 'bias_hcd'], seed=9)
    direct = vega.chi2_batch(theta)
    created = not dist.is_initialized()
    if created:
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ.setdefault('MASTER_PORT', '29547')
        dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        got = chi2_sharded(vega.chi2_batch, theta, device=torch.device('cuda', 0))
        np.testing.assert_array_equal(got, direct)
    finally:
        if created:
            dist.destroy_process_group()
    vega.close()


def test_xi_stage_options_and_model_pk(tmp_path):
    """`rescale-coords-systematics` (radiation term and UV shot noise on the rescaled coordinates), `old_growth_func`,
    `fht_lowring = False` and `model_pk` (the models are the multipoles of the core power spectrum) against the
    unmodified reference."""
    from conftest import options2_problem
    from vega_amd import VegaInterface
    exp = np.load(GOLDEN / 'expected_options2.npz')
    prob = options2_problem(tmp_path, 'cross')
    vega = VegaInterface(None, problem=prob, max_batch=4)
    mask = prob.items['lyalya_qso'].model_mask
    assert vega.chi2() == pytest.approx(float(exp['cross/fid/chi2']), rel=CHI2_RTOL)
    _assert_xi(vega.compute_model()['lyalya_qso'], exp['cross/fid/model'], mask, 'cross fid')
    pars = _pars(exp, 'cross/')
    np.testing.assert_allclose(vega.chi2_batch(pars), exp['cross/chi2'], rtol=CHI2_RTOL)
    for i, w in enumerate(pars):
        _assert_xi(vega.compute_model(w)['lyalya_qso'], exp[f'cross/walker{i}/model'], mask, f'cross walker {i}')
    vega.close()
    prob = options2_problem(tmp_path, 'auto')
    vega = VegaInterface(None, problem=prob, max_batch=2)
    assert vega.chi2() == pytest.approx(float(exp['auto/fid/chi2']), rel=CHI2_RTOL)
    _assert_xi(vega.compute_model()['lyalya_lyalya'], exp['auto/fid/model'], prob.items['lyalya_lyalya'].model_mask, 'auto')
    assert vega.chi2({'ap': 1.03, 'at': 0.96, 'uv_shotnoise_amp': 0.03}) == pytest.approx(float(exp['auto/walker/chi2']), rel=CHI2_RTOL)
    vega.close()
    prob = options2_problem(tmp_path, 'model_pk')
    vega = VegaInterface(None, problem=prob, max_batch=2)
    assert vega.model_pk
    w = {'ap': 1.03, 'bias_eta_LYA': -0.21, 'beta_LYA': 1.5, 'bao_amp': 0.8, 'sigmaNL_par': 7.0}
    for tag, pars in (('fid', None), ('walker', w)):
        model = vega.compute_model(pars)
        for name in prob.items:
            ref = exp[f'model_pk/{tag}/{name}']
            assert model[name].shape == ref.shape == (4, prob.k.size)
            for ell in range(4):
                scale = np.abs(ref[ell]).max()
                assert np.abs(model[name][ell] - ref[ell]).max() <= 1e-10 * scale, (tag, name, ell)
    vega.close()


def test_sharded_monte_carlo_launcher_on_a_real_engine(tmp_path):
    """scripts/run_mc_sharded.py (the reference's bin/run_vega_mc_mpi.py:17-71) with one rank on the GPU: config with
    [control] run_montecarlo / num_mc_mocks / mc_seed, [monte carlo] sampling table and [mc parameters], a data file that
    carries distortion matrix and covariance; the fiducial comes from a fit to the data, the result file has the
    reference's layout and the fits recover the template parameters."""
    import sys
    from conftest import REPO, mc_launcher_config
    from vega_amd import fitslite
    sys.path.insert(0, str(REPO / 'scripts'))
    import run_mc_sharded
    mc_launcher_config(tmp_path, num_mocks=6, seed=3)
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    mc, res, block = run_mc_sharded.run('configs/mc/main.ini', search_dirs=[tmp_path, GOLDEN], max_batch=128,
                                        print_func=lambda message: None)
    assert block == (0, 6) and res.names == ['ap', 'at', 'bias_eta_LYA'] and res.is_valid.all()
    out = tmp_path / 'out' / 'monte_carlo' / 'monte_carlo.fits'
    assert out.is_file()
    tabs = {h.header['EXTNAME'].upper(): h for h in fitslite.open(out)[1:]}
    assert np.asarray(tabs['BESTFIT'].data['values']).shape == (3, 6)
    assert np.asarray(tabs['MOCKS'].data['lyalya_lyalya']).shape == (6, 2500)
    # the mocks were drawn around the model at (fit to the data) + [mc parameters]: the fits find those values again
    vega = mc.vega
    assert vega.bestfit.names == ['ap', 'at']
    truth = {**vega.bestfit.as_dict(), 'beta_LYA': 1.8}
    for j, n in enumerate(res.names):
        want = truth.get(n, vega.params[n])
        assert np.abs((res.values[:, j] - want) / res.errors[:, j]).max() < 5, n
    vega.close()


def test_model_only_correlations_on_the_engine(tmp_path):
    """Correlations without a data file (reference vega/correlation_item.py:40-42, :120-136, vega/vega_interface.py:110-137,
    :208-235): `VegaInterface(main, coordinates={name: Coordinates(...)})` computes the models on the caller's coordinates - no
    distortion matrix, no mask - and refuses chi2, as the reference asserts.  Against the unmodified reference at 1e-8."""
    from conftest import model_only_problem
    from vega_amd import VegaInterface
    _, coordinates = model_only_problem(tmp_path)
    exp = np.load(GOLDEN / 'expected_model_only.npz')
    vega = VegaInterface('configs/modelonly/main.ini', search_dirs=[tmp_path, GOLDEN], max_batch=8, coordinates=coordinates)
    assert not vega._has_data and all(view is None for view in vega.data.values())
    fid = vega.compute_model()
    names = [str(n) for n in exp['param_names']]
    for name in vega.corr_items:
        ref = exp[f'fid/{name}']
        assert fid[name].shape == ref.shape
        np.testing.assert_allclose(fid[name], ref, rtol=0, atol=1e-8 * np.abs(ref).max())
    walkers = [dict(zip(names, row)) for row in exp['theta']]
    batch = vega.compute_model_batch(walkers)
    for i, w in enumerate(walkers):
        one = vega.compute_model(w)
        for name in vega.corr_items:
            ref = exp[f'walker{i}/{name}']
            np.testing.assert_allclose(one[name], ref, rtol=0, atol=1e-8 * np.abs(ref).max())
            np.testing.assert_allclose(batch[name][i], ref, rtol=0, atol=1e-8 * np.abs(ref).max())
    for call in (vega.chi2, vega.log_lik, vega.minimize):
        with pytest.raises(AssertionError, match='data'):
            call()
    vega.close()
