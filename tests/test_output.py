"""The fit result file (the data formats either side of the path) - reference vega/output.py:37-349,
read back as vega/postprocess/fit_results.py:45-130 and `mc_start_from_fit` read it.  CPU: the writer and the HIERARCH
cards of `fitslite`; GPU (tests/test_fits_gpu.py): the statistics `minimize` leaves behind against the reference's."""
import numpy as np
import pytest

from conftest import load_problem


class _Fit:
    names = ['bias_eta_LYA', 'beta_LYA']
    values = np.array([[-0.2, 1.67]])
    errors = np.array([[0.01, 0.05]])
    covariance = np.array([[[1e-4, -2e-5], [-2e-5, 2.5e-3]]])
    fval = np.array([0.64])
    is_valid = np.array([True])
    hesse_failed = np.array([False])
    has_accurate_covar = np.array([True])


def test_write_results_layout_and_hierarch_cards(tmp_path):
    from vega_amd import fitslite
    from vega_amd.output import Output
    prob = load_problem('full4')
    rng = np.random.default_rng(3)
    models = {name: rng.standard_normal(item.dist_grid.size) for name, item in prob.items.items()}
    params = {'ap': 1.0, 'at': 0.99, 'bias_eta_LYA': -0.2, 'sigma_velo_disp_lorentz_QSO': 6.86, 'growth_rate': 0.97}
    stats = {name: {'masked_size': int(item.data_size), 'chisq': 0.125 * (i + 1), 'reduced_chisq': 1e-4 * (i + 1),
                    'p_value': 1.0, 'bestfit_marg_coeff': None if i else np.array([0.5, -1.5])}
             for i, (name, item) in enumerate(prob.items.items())}
    scan = [{'ap': 0.9 + 0.1 * i, 'bias_eta_LYA': -0.2, 'beta_LYA': 1.6, 'fval': 3.0 - i} for i in range(3)]
    out = Output({'filename': str(tmp_path / 'result')}, prob.items)
    out.analysis = type('A', (), {'grids': {'ap': np.linspace(0.9, 1.1, 3)}})()
    path = out.write_results(models, params, _Fit(), stats, scan)
    assert path.endswith('result.fits')
    with pytest.raises(OSError):
        out.write_results(models, params, _Fit(), stats, scan)          # (overwrite is off, as in the reference)

    hdus = fitslite.open(path)
    names = [h.header.get('EXTNAME') for h in hdus[1:]]
    assert names == ['MODEL_' + n.upper() for n in prob.items] + ['BESTFIT', 'SCAN']       # (upper case, as astropy stores hdu.name)
    for i, (name, item) in enumerate(prob.items.items()):
        h = hdus[1 + i]
        n = item.dist_grid.size
        np.testing.assert_array_equal(h.data[name + '_MODEL'], models[name])
        np.testing.assert_array_equal(np.asarray(h.data[name + '_MODEL_MASK'], dtype=bool), item.model_mask)
        np.testing.assert_array_equal(np.asarray(h.data[name + '_MASK'], dtype=bool)[:item.data_mask.size], item.data_mask)
        np.testing.assert_array_equal(h.data[name + '_DATA'][:item.data_vec.size], item.data_vec)
        assert np.isnan(h.data[name + '_DATA'][item.data_vec.size:]).all()
        np.testing.assert_array_equal(h.data[name + '_VAR'][:item.data_vec.size], np.diag(item.cov) if item.cov is not None
                                      else np.ones(item.data_vec.size))
        np.testing.assert_array_equal(h.data[name + '_RP'], item.dist_grid.rp)
        np.testing.assert_array_equal(h.data[name + '_RT'], item.dist_grid.rt)
        assert len(h.data[name + '_Z']) == n
        # parameters and statistics as HIERARCH cards, case kept (reference output.py:210-228)
        for par, val in params.items():
            assert h.header[par] == val
        assert h.header['masked_size'] == item.data_size and h.header['chisq'] == 0.125 * (i + 1)
        assert h.header['reduced_chisq'] == pytest.approx(1e-4 * (i + 1), rel=1e-15) and h.header['p_value'] == 1.0
        assert ('marg_coeff_0' in h.header) == (i == 0)
        if i == 0:
            assert (h.header['marg_coeff_0'], h.header['marg_coeff_1']) == (0.5, -1.5)
    best = hdus[len(prob.items) + 1]
    assert [s.decode().strip() if isinstance(s, bytes) else str(s).strip() for s in best.data['names']] == _Fit.names
    np.testing.assert_array_equal(best.data['values'], _Fit.values[0])
    np.testing.assert_array_equal(best.data['errors'], _Fit.errors[0])
    np.testing.assert_array_equal(np.asarray(best.data['covariance']).reshape(2, 2), _Fit.covariance[0])
    assert best.header['FVAL'] == 0.64 and best.header['VALID'] is True and best.header['ACCURATE'] is True
    sc = hdus[-1]
    # (four keys, three grid points: as long as the longer of the two, the shorter columns padded - what astropy's from_columns
    # makes of the reference's unequal columns, vega/output.py:319-331)
    np.testing.assert_allclose(sc.data['ap'], [0.9, 1.0, 1.1, 0.0])
    np.testing.assert_array_equal(sc.data['fval'], [3.0, 2.0, 1.0, 0.0])
    assert [str(s).strip() for s in sc.data['names']] == ['ap', 'bias_eta_LYA', 'beta_LYA', 'fval']
    assert sc.header['ap_min'] == 0.9 and sc.header['ap_max'] == 1.1 and sc.header['ap_num_bins'] == 3


def test_output_refuses_what_it_does_not_write(tmp_path):
    from vega_amd.output import Output
    prob = load_problem('full4')
    out = Output({'filename': str(tmp_path / 'x'), 'type': 'hdf'}, prob.items)
    with pytest.raises(NotImplementedError):
        out.write_results({}, {})
    with pytest.raises(ValueError):
        Output(None, prob.items).write_results({}, {})


def test_effective_data_size_counts_the_marginalised_modes():
    """reference vega/data.py:134, :825: fitted bins minus the modes the small-scale marginalisation removes"""
    prob = load_problem('full4')
    for item in prob.items.values():
        assert item.num_marg_modes == 0 and item.effective_data_size == item.data_size


def test_fit_results_reader_round_trip(tmp_path):
    """vega/postprocess/fit_results.py:33-141 over the written file: correlation names from the (upper-case) HDU names,
    columns matched without regard to case, `correlations` keyed by the lower-case name"""
    from vega_amd.fit_results import FitResults
    from vega_amd.output import Output
    prob = load_problem('full4')
    rng = np.random.default_rng(4)
    models = {name: rng.standard_normal(item.dist_grid.size) for name, item in prob.items.items()}
    stats = {name: {'masked_size': int(item.data_size), 'chisq': 1.5, 'reduced_chisq': 1e-3, 'p_value': 0.75,
                    'bestfit_marg_coeff': np.array([2.0]) if name == 'lyalya_qso' else None}
             for name, item in prob.items.items()}
    out = Output({'filename': str(tmp_path / 'r.fits')}, prob.items)
    out.write_results(models, {'ap': 1.0}, _Fit(), stats)
    res = FitResults(tmp_path / 'r.fits')
    assert res.chisq == 0.64 and res.valid is True and res.accurate is True and res.num_pars == 2
    assert list(res.names) == _Fit.names and res.params['beta_LYA'] == 1.67 and res.sigmas['bias_eta_LYA'] == 0.01
    np.testing.assert_array_equal(res.cov, _Fit.covariance[0])
    assert list(res.correlations) == list(prob.items)
    assert res.num_data_points == sum(item.data_size for item in prob.items.values())
    assert res.reduced_chisq == 0.64 / (res.num_data_points - 2)
    for name, item in prob.items.items():
        c = res.correlations[name]
        np.testing.assert_array_equal(c.model, models[name])
        np.testing.assert_array_equal(c.model_mask, item.model_mask)
        np.testing.assert_array_equal(c.data[c.data_mask], item.masked_data_vec)
        assert (c.size, c.chisq, c.reduced_chisq, c.p_value) == (item.data_size, 1.5, 1e-3, 0.75)
        np.testing.assert_array_equal(c.bestfit_marg_coeff, [2.0] if name == 'lyalya_qso' else [])
    assert FitResults(tmp_path / 'r.fits', results_only=True).marg_coeff == {}


def test_output_write_monte_carlo_paths(tmp_path):
    """reference vega/output.py:510-520: [output] mc_output, else monte_carlo/ next to the result file; one file per rank"""
    from types import SimpleNamespace
    from vega_amd import fitslite
    from vega_amd.output import Output
    prob = load_problem('full4')
    analysis = SimpleNamespace(has_monte_carlo=True, mc_mocks={'global': np.arange(12.).reshape(3, 4)}, mc_bestfits=None)
    out = Output({'filename': str(tmp_path / 'fits' / 'result.fits')}, prob.items, analysis)
    path = out.write_monte_carlo(cpu_id=2)
    assert path == tmp_path / 'fits' / 'monte_carlo' / 'monte_carlo_2.fits'
    np.testing.assert_array_equal(fitslite.open(path)[1].data['global'], analysis.mc_mocks['global'])
    out = Output({'filename': str(tmp_path / 'x.fits'), 'mc_output': str(tmp_path / 'elsewhere')}, prob.items, analysis)
    assert out.write_monte_carlo() == tmp_path / 'elsewhere' / 'monte_carlo.fits'
    with pytest.raises(ValueError):
        Output({'filename': 'x'}, prob.items).write_monte_carlo()


def test_marginalised_modes_reduce_the_effective_data_size(tmp_path):
    """reference vega/data.py:816-825, :134: the modes the SVD of the masked, prior-scaled templates keeps"""
    from conftest import marginalization_problem, MARGINALIZATION_CASES
    prob = marginalization_problem(tmp_path, MARGINALIZATION_CASES['rtmax'])
    item = prob.items['lyalya_lyalya']
    tm = item.marg_templates[item.model_mask, :]
    tm = tm.toarray() if hasattr(tm, 'toarray') else np.asarray(tm)
    sv = np.linalg.svd(tm * 5.0, compute_uv=False)
    modes = int((sv > 1e-8 * sv[0]).sum())
    assert 0 < modes <= tm.shape[1] and item.num_marg_modes == modes
    assert item.effective_data_size == item.data_size - modes
