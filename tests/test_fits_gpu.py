"""The fit drivers around the engine against the reference's own drivers (SURVEY section 8f "next" #1, #2): `minimize`
(bias pre-fit + full MIGRAD, reference vega/minimizer.py:39-103), `chi2_scan` (vega/analysis.py:53-122),
`initialize_monte_carlo` (vega/vega_interface.py:505-544) and `run_monte_carlo` (vega/analysis.py:224-308).

`expected_fits.npz` was written by the UNMODIFIED reference walking those drivers with the MIGRAD restatement of
vega_amd/migrad.py behind the iminuit surface (tools/refshim/iminuit): it pins everything around the minimiser - grids,
pinned parameters, start values, seeding of the mocks, what is kept.  The expected fit values in it are therefore
SELF-GENERATED as far as the minimiser goes: an error inside vega_amd/migrad.py would be on both sides of these comparisons.
MIGRAD's own arithmetic is anchored elsewhere - by the reference's golden fit value (tests/test_vega.py:18), below, and by
outcomes known independently of this repository (analytic minima, MINUIT's error definition, EDM criterion, limits, the
`iterate` loop: tests/test_migrad_analytic.py, CPU), with `method='bfgs'` + SciPy on the oracle as a second witness
(tests/test_minimizer_gpu.py).
"""
from math import isclose

import numpy as np
import pytest

from conftest import GOLDEN, fits_problem, load_problem

pytestmark = pytest.mark.gpu


def test_migrad_reproduces_the_references_pinned_fit_value():
    """reference tests/test_vega.py:16-18: `isclose(vega.bestfit.fmin.fval, 0.6409716347033996)` (rel 1e-9) - MIGRAD's
    stopping point, 1.1e-4 above the bounded minimum 0.6408605 - through the HIP engine and the MIGRAD restatement."""
    from vega_amd import VegaInterface
    vega = VegaInterface(None, problem=load_problem('full4'), max_batch=64)
    res = vega.minimize()
    assert res.names == ['bias_eta_LYA', 'beta_LYA']
    assert isclose(res.fval[0], 0.6409716347033996)
    assert vega.chi2(res.as_dict()) == pytest.approx(res.fval[0], rel=1e-12)
    assert res.is_valid[0] and res.nfcn[0] < 80
    vega.close()


def test_minimize_scan_and_monte_carlo_against_the_references_drivers(tmp_path):
    from vega_amd import VegaInterface
    exp = np.load(GOLDEN / 'expected_fits.npz')
    prob = fits_problem(tmp_path)
    vega = VegaInterface(None, problem=prob, max_batch=256)
    names = [str(n) for n in exp['fit/names']]
    # the fit to the data: same trajectory, same stopping point
    res = vega.minimize()
    assert res.names == names
    np.testing.assert_allclose(res.fval[0], float(exp['fit/fval']), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(res.values[0], exp['fit/values'], rtol=1e-6)
    np.testing.assert_allclose(res.errors[0], exp['fit/errors'], rtol=1e-4)
    np.testing.assert_allclose(res.covariance[0], exp['fit/covariance'], rtol=1e-3, atol=1e-9)
    # (the full fit's evaluations, as the reference's last Minuit object counts them, + the 18 of the bias pre-fit)
    assert int(res.nfcn[0]) == int(exp['fit/nfcn']) + 18
    # chi2 scan: grids, order, pinned values, every grid point's fit
    scan = vega.chi2_scan()
    grid_names = [str(n) for n in exp['scan/grid_names']]
    assert list(vega.analysis.grids) == grid_names
    for g in grid_names:
        np.testing.assert_array_equal(vega.analysis.grids[g], exp[f'scan/grid/{g}'])
    keys = [str(k) for k in exp['scan/keys']]
    got = np.array([[row[k] for k in keys] for row in scan])
    ref = exp['scan/results']
    assert got.shape == ref.shape
    for c, k in enumerate(keys):
        if k == 'fval':
            np.testing.assert_allclose(got[:, c], ref[:, c], rtol=1e-4, atol=1e-8)
        else:
            np.testing.assert_allclose(got[:, c], ref[:, c], rtol=1e-5)
    # ... and into the result file as run_vega writes it (reference vega/scripts/run_vega.py:32-46, vega/output.py:291-349)
    from vega_amd import fitslite
    vega.output.outfile, vega.output.overwrite = str(tmp_path / 'scan_result'), True
    params = {**vega.params, **res.as_dict()}
    path = vega.output.write_results(vega.bestfit_model, params, vega.minimizer, vega.bestfit_corr_stats, scan)
    sc = [h for h in fitslite.open(path)[1:] if h.header['EXTNAME'] == 'SCAN'][0]
    for c, k in enumerate(keys):
        np.testing.assert_array_equal(sc.data[k][:len(scan)], got[:, c])
    for g in grid_names:
        assert (sc.header[g + '_min'], sc.header[g + '_max'], sc.header[g + '_num_bins']) == \
            (vega.analysis.grids[g][0], vega.analysis.grids[g][-1], len(vega.analysis.grids[g]))
    vega.close()

    # Monte Carlo: fiducial from a fit to the data + [mc parameters], one mock installed as data
    vega = VegaInterface(None, problem=fits_problem(tmp_path), max_batch=256)
    mocks = vega.initialize_monte_carlo(print_func=lambda message: None)
    for name in prob.items:
        ref = exp[f'mcinit/mock/{name}']
        keep = np.isfinite(ref)
        np.testing.assert_array_equal(np.isfinite(mocks[name]), keep)
        np.testing.assert_allclose(mocks[name][keep], ref[keep], rtol=0, atol=2e-7 * np.abs(ref[keep]).max())
    assert vega.monte_carlo
    assert vega.chi2() == pytest.approx(float(exp['mcinit/chi2']), rel=1e-4)
    assert vega.log_lik() == pytest.approx(float(exp['mcinit/log_lik']), rel=1e-6)
    vega.close()

    vega = VegaInterface(None, problem=fits_problem(tmp_path), max_batch=256)
    fid = vega.get_fiducial_for_monte_carlo(print_func=lambda message: None)
    for name in prob.items:
        ref = exp[f'mc/fiducial/{name}']
        assert np.abs(fid[name] - ref).max() <= 2e-7 * np.abs(ref).max()
    # the mocks are drawn around the REFERENCE's fiducial here, so that the fits below are compared on the same data
    res = vega.run_monte_carlo({name: exp[f'mc/fiducial/{name}'] for name in prob.items}, num_mocks=2, seed=5)
    mc = vega.analysis
    mc_names = [str(n) for n in exp['mc/names']]
    assert res.names == mc_names and res.is_valid.all()
    for name in prob.items:
        ref = exp[f'mc/mocks/{name}'][:, prob.items[name].data_mask]       # (the reference keeps mocks on the full grid)
        np.testing.assert_allclose(np.asarray(mc.mc_mocks[name]), ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    bestfits = np.array([mc.mc_bestfits[n] for n in mc_names])
    # (chi2 values that differ in the 15th digit can send one line search of a fit down another branch: the stopping points
    # then differ by a small fraction of MIGRAD's own EDM tolerance - 1e-3 of the reported errors here - not bit for bit)
    assert (np.abs(bestfits[:, :, 0] - exp['mc/bestfits'][:, :, 0]) <= 2e-3 * exp['mc/bestfits'][:, :, 1]).all()
    np.testing.assert_allclose(bestfits[:, :, 1], exp['mc/bestfits'][:, :, 1], rtol=5e-3)
    np.testing.assert_allclose(mc.mc_chisq, exp['mc/chisq'], rtol=1e-7)
    # the launchers' last line: vega.output.write_monte_carlo(rank) (reference bin/run_vega_mc_mpi.py:67-71, output.py:510-520)
    from vega_amd import fitslite
    assert vega.output.analysis is mc
    vega.output.outfile, vega.output.overwrite = str(tmp_path / 'results' / 'fit.fits'), True
    path = vega.output.write_monte_carlo(cpu_id=3)
    assert path == tmp_path / 'results' / 'monte_carlo' / 'monte_carlo_3.fits'
    tabs = {h.header['EXTNAME']: h for h in fitslite.open(path)[1:]}
    assert set(tabs) == {'BESTFIT', 'FITINFO', 'MOCKS'} and list(tabs['FITINFO'].data['valid_minima']) == [True, True]
    np.testing.assert_array_equal(np.asarray(tabs['FITINFO'].data['chisq']), mc.mc_chisq)
    vega.close()


def test_mc_start_from_fit_reads_the_bestfit_table(tmp_path):
    """`[control] mc_start_from_fit` (reference vega/vega_interface.py:465-472): the Monte-Carlo fiducial takes the best-fit
    values of an existing fit file - its BESTFIT table with `names` / `values` columns (vega/postprocess/fit_results.py:47-53) -
    under the [mc parameters]; no initial fit is run."""
    from vega_amd import VegaInterface, fitslite
    prob = fits_problem(tmp_path)
    fit_file = tmp_path / 'previous_fit.fits'
    fit_names = np.array(['ap', 'at', 'bias_eta_LYA', 'beta_LYA'])
    fit_values = np.array([1.031, 0.972, -0.2051, 1.71])
    fitslite.write_tables(str(fit_file), [
        ('MODEL', [('dummy', 'D', np.zeros(3))]),
        ('BESTFIT', [('names', '12A', fit_names), ('values', 'D', fit_values), ('errors', 'D', 0.01 * np.ones(4))])], overwrite=True)
    prob.main_config['control']['mc_start_from_fit'] = str(fit_file)
    vega = VegaInterface(None, problem=prob, max_batch=8)
    calls = []
    vega.minimize = lambda *a, **k: calls.append(1)         # (must not be called: the fit file replaces the initial fit)
    said = []
    fid = vega.get_fiducial_for_monte_carlo(print_func=said.append)
    assert not calls and any('Reading input fit' in s for s in said)
    want = dict(zip(fit_names, fit_values))
    want.update(prob.mc_config['params'])                   # [mc parameters] beta_LYA = 1.8 wins over the fit's 1.71
    assert want['beta_LYA'] == 1.8
    ref = vega.compute_model(want)
    for name in prob.items:
        # (a repeated single-walker call may run against the tables the first one left: the last bit can differ)
        np.testing.assert_allclose(fid[name], ref[name], rtol=1e-13, atol=1e-16)
    other = vega.compute_model(dict(zip(fit_names, fit_values)))
    assert any(np.abs(other[n] - ref[n]).max() > 1e-6 * np.abs(ref[n]).max() for n in prob.items)
    vega.close()


def test_chi2_scan_with_the_vectorised_minimiser(tmp_path):
    """`chi2_scan(method='bfgs')`: the same grid and pinned parameters through the second minimiser (Minuit's conventions, not
    its trajectory) - the minima agree with MIGRAD's within a fraction of the reported errors."""
    from vega_amd import VegaInterface
    vega = VegaInterface(None, problem=fits_problem(tmp_path), max_batch=256)
    a = vega.chi2_scan()
    grids = {k: v.copy() for k, v in vega.analysis.grids.items()}
    b = vega.chi2_scan(method='bfgs')
    assert list(vega.analysis.grids) == list(grids) and len(a) == len(b) == 6
    for ra, rb in zip(a, b):
        for g in grids:
            assert ra[g] == rb[g]                           # the pinned grid values
        assert rb['fval'] == pytest.approx(ra['fval'], rel=1e-3, abs=1e-3)
    vega.close()


def test_bestfit_statistics_and_result_file_against_the_reference(tmp_path):
    """What the reference's `minimize` leaves next to the fit (vega/vega_interface.py:593-643) and what `run_vega` writes
    (vega/scripts/run_vega.py:7-51, vega/output.py:37-289) on the reference's own test configuration:
    `expected_fit_stats.npz` = the unmodified reference's per-correlation sizes / chi2 / reduced chi2 / p-values, totals and
    best-fit models (make_golden.dump_fit_stats).  The result file is read back as vega/postprocess/fit_results.py reads it."""
    import vega_amd
    from vega_amd import fitslite
    exp = np.load(GOLDEN / 'expected_fit_stats.npz')
    main = (GOLDEN / 'configs/full4/main.ini').read_text()
    main = main.replace('filename = lyalya_lyalya__lyalya_lyalyb__lyalya_qso__lyalyb_qso', f'filename = {tmp_path}/fit_result')
    cfg = tmp_path / 'main.ini'
    cfg.write_text(main)
    lines = []
    vega = vega_amd.run_vega(str(cfg), search_dirs=[GOLDEN], print_func=lines.append, max_batch=64)
    names = [str(n) for n in exp['names']]
    assert list(vega.bestfit_corr_stats) == names
    assert isclose(vega.chisq, float(exp['chisq']), rel_tol=1e-8) and vega.total_data_size == int(exp['total_data_size'])
    assert isclose(vega.reduced_chisq, float(exp['reduced_chisq']), rel_tol=1e-8) and isclose(vega.p_value, float(exp['p_value']))
    for name in names:
        st = vega.bestfit_corr_stats[name]
        assert st['masked_size'] == int(exp[f'stats/{name}/masked_size']) and st['bestfit_marg_coeff'] is None
        assert st['chisq'] == pytest.approx(float(exp[f'stats/{name}/chisq']), rel=1e-6, abs=1e-10)
        assert st['reduced_chisq'] == pytest.approx(float(exp[f'stats/{name}/reduced_chisq']), rel=1e-6, abs=1e-13)
        assert st['p_value'] == pytest.approx(float(exp[f'stats/{name}/p_value']))
        ref = exp[f'model/{name}']
        np.testing.assert_allclose(vega.bestfit_model[name], ref, rtol=0, atol=1e-8 * np.abs(ref).max())
    assert any(line.startswith('Total chi^2/(ndata-nparam): 0.6/(9540-2)') for line in lines)
    # vega.minimizer under the reference's names (vega/minimizer.py:105-187), as run_vega and user scripts read it
    m = vega.minimizer
    assert m.fit is vega.bestfit and list(m.values) == [str(n) for n in exp['fit/names']]
    assert m.values == vega.bestfit.as_dict() and isclose(m.fmin.fval, 0.6409716347033996) and m.fmin.is_valid and m.minuit.valid
    assert m.covariance.shape == (2, 2) and m.errors['beta_LYA'] == vega.bestfit.errors[0, 1] and m.params[0].name == 'bias_eta_LYA'

    from vega_amd.fit_results import FitResults
    res = FitResults(tmp_path / 'fit_result.fits')
    assert list(res.names) == [str(n) for n in exp['fit/names']] and list(res.correlations) == names
    np.testing.assert_allclose(res.mean, exp['fit/values'], rtol=1e-6)
    assert isclose(res.chisq, 0.6409716347033996) and res.valid is True
    assert res.num_data_points == 9540 and isclose(res.reduced_chisq, vega.reduced_chisq) and isclose(res.p_value, vega.p_value)
    for name in names:
        c = res.correlations[name]
        np.testing.assert_array_equal(c.model, vega.bestfit_model[name])
        assert c.chisq == vega.bestfit_corr_stats[name]['chisq'] and c.size == 1590 * (1 + name.endswith('qso'))
    hdus = fitslite.open(str(tmp_path / 'fit_result.fits'))
    h = [x for x in hdus[1:] if x.header['EXTNAME'] == 'MODEL_LYALYA_QSO'][0]
    assert h.header['bias_eta_LYA'] == vega.bestfit.as_dict()['bias_eta_LYA'] and h.header['ap'] == 1.05
    vega.close()


def test_compute_sensitivity_against_the_reference():
    """`VegaInterface.compute_sensitivity` (reference vega/vega_interface.py:956-1075; fixture: make_golden.dump_sensitivity, the
    unmodified reference on the auto + cross items): central differences of the four parts of every correlation - the final
    and the raw core model of the peak and of the smooth component - and the Fisher information per bin.  Here the 2 P points
    go through the engine in batches, the components' final models come from the model's being affine in bao_amp, the raw ones from the
    per-pipeline stage taps.  (Differences of 1e-15-accurate models over a step of 0.1 sigma: compared at 1e-9 of the largest
    entry of each array.)"""
    from vega_amd import VegaInterface
    exp = np.load(GOLDEN / 'expected_sensitivity.npz')
    nominal = {str(p): (float(v), float(e)) for p, (v, e) in zip(exp['params'], exp['nominal'])}
    vega = VegaInterface(None, problem=load_problem('joint'), max_batch=64)
    lines = []
    sens = vega.compute_sensitivity(nominal=nominal, frac=0.1, print_func=lines.append)
    assert sens is vega.sensitivity and sens['nominal'] == nominal and len(lines) == len(nominal) + 1
    for name in (str(n) for n in exp['names']):
        assert list(sens['partials'][name]) == list(nominal)
        for pname in nominal:
            ref = exp[f'partials/{name}/{pname}']
            got = sens['partials'][name][pname]
            assert got.shape == ref.shape == (2, 2, vega.problem.items[name].model_grid.size)
            for i in range(2):
                for j in range(2):
                    np.testing.assert_allclose(got[i, j], ref[i, j], rtol=0, atol=1e-9 * max(np.abs(ref[i, j]).max(), 1e-300))
        assert ['/'.join(k) for k in sens['fisher'][name]] == [str(k) for k in exp[f'fisher_keys/{name}']]
        mask = vega.problem.items[name].data_mask
        for key, got in sens['fisher'][name].items():
            if 'ap' not in key:
                continue
            ref = exp[f'fisher/{name}/{key[0]}/{key[1]}']
            assert np.isnan(got[:, ~mask]).all() and np.isnan(ref[:, ~mask]).all()
            np.testing.assert_allclose(got[:, mask], ref[:, mask], rtol=0, atol=1e-8 * max(np.abs(ref[:, mask]).max(), 1e-300))
    # the auto-correlation does not know the quasar parameter: exact zeros there, as in the reference
    assert not sens['partials']['lyalya_lyalya']['drp_QSO'].any() and exp['partials/lyalya_lyalya/drp_QSO'].any() == False  # noqa: E712
    vega.close()


def test_model_components_and_write_cf_against_the_reference(tmp_path):
    """`save-components` / `write_cf` (reference vega/model.py:41-45, :113-115, :151-153; vega/output.py:375-440): the raw core
    correlations of the peak and smooth spectra and the components' final models of the unmodified reference at one walker
    (make_golden.dump_components) against `VegaInterface.model_components` - one two-walker evaluation, bao_amp = 0 and 1, and
    the per-pipeline stage taps - and the Xi_<name> HDUs `write_results` makes of them."""
    from vega_amd import VegaInterface, fitslite
    from vega_amd.output import Output
    exp = np.load(GOLDEN / 'expected_components.npz')
    pars = {str(n): float(v) for n, v in zip(exp['param_names'], exp['theta'][0])}
    prob = load_problem('joint')
    vega = VegaInterface(None, problem=prob, max_batch=8)
    comps = vega.model_components(pars)
    full = vega.compute_model(pars)
    names = [str(n) for n in exp['names']]
    assert list(comps) == names
    for name in names:
        np.testing.assert_allclose(full[name], exp[f'model/{name}'], rtol=0, atol=1e-8 * np.abs(exp[f'model/{name}']).max())
        for key in ('xi', 'xi_distorted'):
            for part in ('peak', 'smooth'):
                ref = exp[f'{key}/{name}/{part}']
                got = comps[name][key][part]['core']
                assert got.shape == ref.shape
                np.testing.assert_allclose(got, ref, rtol=0, atol=1e-8 * np.abs(ref).max())
        # the parts add up to the model: bao_amp * peak + smooth (vega/model.py:185)
        total = pars['bao_amp'] * comps[name]['xi_distorted']['peak']['core'] + comps[name]['xi_distorted']['smooth']['core']
        np.testing.assert_allclose(total, full[name], rtol=0, atol=1e-14 * np.abs(full[name]).max())
    out = Output({'filename': str(tmp_path / 'with_cf'), 'write_cf': 'True'}, prob.items)
    out.output_cf = True
    path = out.write_results(full, pars, models=comps)
    hdus = {h.header['EXTNAME']: h for h in fitslite.open(path)[1:]}
    assert list(hdus) == ['MODEL_' + n.upper() for n in names] + ['XI_' + n.upper() for n in names]
    h = hdus['XI_LYALYA_QSO']
    assert list(h.columns.names) == ['raw_peak_core', 'raw_smooth_core', 'distorted_peak_core', 'distorted_smooth_core']
    np.testing.assert_array_equal(h.data['raw_smooth_core'], comps['lyalya_qso']['xi']['smooth']['core'])
    np.testing.assert_array_equal(h.data['distorted_peak_core'], comps['lyalya_qso']['xi_distorted']['peak']['core'])
    with pytest.raises(ValueError):
        out.write_results(full, pars, models=None)
    vega.close()
    # with metal terms (default no-metal-decomp = True): the model keeps the 'core' entries alone, the metal terms sit inside the
    # smooth component's final model - the reference saved these with `fast_metal_bias = False` (vega/metals.py:242)
    metals = VegaInterface(None, problem=load_problem('auto_metals'), max_batch=8)
    pars = {str(n): float(v) for n, v in zip(exp['metals/param_names'], exp['metals/theta'][0])}
    comps = metals.model_components(pars)['lyalya_lyalya']
    for key in ('xi', 'xi_distorted'):
        for part in ('peak', 'smooth'):
            ref = exp[f'metals/{key}/{part}']
            np.testing.assert_allclose(comps[key][part]['core'], ref, rtol=0, atol=1e-8 * np.abs(ref).max())
    np.testing.assert_allclose(metals.compute_model(pars)['lyalya_lyalya'], exp['metals/model'], rtol=0,
                               atol=1e-8 * np.abs(exp['metals/model']).max())
    metals.close()
