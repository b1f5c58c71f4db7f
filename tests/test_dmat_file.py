"""SURVEY 8(f)3, the branch DESI production files use: a separate `distortion-file` whose model grid is COEFMOD times
finer than the data grid, and a `covariance-file` (reference vega/data.py:441-473).  CPU part: set-up and oracle against
what the UNMODIFIED reference computed from the same files (tests/golden/make_golden.py::dump_dmat_file)."""
import numpy as np
import pytest

from conftest import GOLDEN, MARGINALIZATION_CASES, dmat_file_problem


@pytest.fixture(scope='module')
def problem(tmp_path_factory):
    return dmat_file_problem(tmp_path_factory.mktemp('dmat'))


def test_grids_of_the_distortion_file(problem):
    item = problem.items['lyalya_lyalya']
    assert item.model_grid.size == 10000 and item.dist_grid.size == 2500 and item.data_grid.size == 2500
    assert item.distortion.shape == (2500, 10000)
    assert (item.model_grid.n_rp, item.model_grid.n_rt) == (100, 100)
    # the distorted-model grid is the REGULAR data grid (no coordinates in the file for it: data.py:470-471)
    np.testing.assert_array_equal(item.dist_grid.rp, item.dist_grid.rp_regular)
    # model bin sizes stay the data's (reference vega/model.py:38-39)
    assert item.core.pk.bin_size_rp == 4.0 and item.core.pk.bin_size_rt == 4.0


def test_oracle_matches_the_reference(problem):
    from oracle import vega_cpu as oc
    exp = np.load(GOLDEN / 'expected_dmat_file.npz')
    assert oc.chi2(problem) == pytest.approx(float(exp['fid/chi2']), rel=1e-10)
    assert oc.log_lik(problem) == pytest.approx(float(exp['fid/log_lik']), rel=1e-10)
    model = oc.compute_model(problem)['lyalya_lyalya']
    np.testing.assert_allclose(model, exp['fid/model'], rtol=1e-9, atol=1e-12 * np.abs(exp['fid/model']).max())
    names = [str(n) for n in exp['param_names']]
    for i in (0, 3, 7):
        pars = dict(zip(names, exp['theta'][i]))
        assert oc.chi2(problem, pars) == pytest.approx(float(exp['walkers/chi2'][i]), rel=1e-10)
        np.testing.assert_allclose(oc.compute_model(problem, pars)['lyalya_lyalya'], exp['walkers/model'][i], rtol=1e-9,
                                   atol=1e-12 * np.abs(exp['fid/model']).max())


def test_marginalisation_on_the_finer_grid(tmp_path):
    from oracle import vega_cpu as oc
    exp = np.load(GOLDEN / 'expected_dmat_file.npz')
    prob = dmat_file_problem(tmp_path, marg_options=MARGINALIZATION_CASES['rtmax'])
    # (800 templates = the 2 x 4 x 100 model bins at rt < 16 on the FINER grid; the reference keeps 574 modes of them)
    assert prob.items['lyalya_lyalya'].marg_templates.shape == (2500, 800)
    assert oc.chi2(prob) == pytest.approx(float(exp['marg/chi2']), rel=1e-8)
    assert oc.log_lik(prob) == pytest.approx(float(exp['marg/log_lik']), rel=1e-8)
