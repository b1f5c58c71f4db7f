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
    vega.close()
