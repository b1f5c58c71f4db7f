"""SURVEY 8(f)3 on the GPU: the `distortion-file` + COEFMOD = 2 case (reference vega/data.py:441-473; model grid 100 x 100,
distorted / data grid 50 x 50) through the C ABI against what the UNMODIFIED reference computed from the same files
(tests/golden/expected_dmat_file.npz) - full chain and chi2-only evaluations, B = 1, 8 and 256, the quadratic form in both of
its shapes: the half-form Q' (nq^2 flops per walker) and the factored form || U r0 - F dx ||^2 (2 n_masked nq), which the
engine picks by itself here (nq = 10 012 against n_masked = 1590).  Bars: chi2 1e-6, xi 1e-8 (BASELINE.json north_star)."""
import numpy as np
import pytest

from conftest import GOLDEN, MARGINALIZATION_CASES, dmat_file_problem, synth_joint_problem

pytestmark = pytest.mark.gpu

CHI2_RTOL = 1e-6
XI_RTOL = 1e-8


@pytest.fixture(scope='module')
def problem(tmp_path_factory):
    return dmat_file_problem(tmp_path_factory.mktemp('dmat'))


@pytest.fixture(scope='module')
def expected():
    exp = np.load(GOLDEN / 'expected_dmat_file.npz')
    return exp, [str(n) for n in exp['param_names']]


def _theta(vega, names, rows):
    return np.stack([vega.engine.theta_from_params(dict(zip(names, row))) for row in rows])


def _assert_xi(got, ref, what):
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= XI_RTOL * scale, what


def test_full_chain_against_the_reference(problem, expected):
    from vega_amd import VegaInterface
    exp, names = expected
    vega = VegaInterface(None, problem=problem, max_batch=256)
    assert vega.engine.model_size == 2500                       # models live on the distorted grid
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    assert vega.log_lik() == pytest.approx(float(exp['fid/log_lik']), rel=1e-9)
    _assert_xi(vega.compute_model()['lyalya_lyalya'], exp['fid/model'], 'fiducial model')
    theta = _theta(vega, names, exp['theta'])
    for batch in (1, 8, 256):
        rows = np.resize(np.arange(8), batch)
        chi2, status, model = vega.engine.eval(theta[rows], want_model=True)
        assert not status.any()
        np.testing.assert_allclose(chi2, exp['walkers/chi2'][rows], rtol=CHI2_RTOL)
        for b in {0, batch // 2, batch - 1}:
            _assert_xi(model[b], exp['walkers/model'][rows[b]], f'B = {batch}, walker {b}')
    pars = dict(zip(names, exp['theta'][2]))
    assert vega.log_lik(pars) == pytest.approx(float(exp['walkers/log_lik'][2]), rel=1e-9)
    vega.close()


@pytest.mark.parametrize('kind', ['auto', 'q', 'factored'])
def test_chi2_only_forms_against_the_reference(problem, expected, kind):
    import torch
    from vega_amd import VegaInterface
    exp, names = expected
    vega = VegaInterface(None, problem=problem, max_batch=256)
    eng = vega.engine
    assert eng.quadratic_form
    eng.set_quadratic_form_kind(kind)
    theta = _theta(vega, names, exp['theta'])
    want_form = 'q' if kind == 'q' else 'factored'              # nq = 10 012 against n_masked = 1590: auto takes the factored form
    for batch in (1, 8, 9, 256):
        rows = np.resize(np.arange(8), batch)
        chi2, status = eng.eval(theta[rows])[:2]
        assert eng.last_form() == want_form
        assert not status.any()
        np.testing.assert_allclose(chi2, exp['walkers/chi2'][rows], rtol=CHI2_RTOL)
    # device entry (two batches in flight), walkers that share their Arinyo / smoothing parameters
    from vega_amd import synthetic
    shared = synthetic.walkers(eng.low.theta0, eng.names, 256, varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'bias_hcd', 'L0_hcd'], seed=99)
    full = eng.eval(shared, want_model=True)[0]
    assert eng.last_form() == 'full'
    eng.set_constant_nl_hint(True, gaussian=True)
    eng.set_lanes(2)
    d_theta = torch.from_numpy(shared).cuda()
    outs = [torch.zeros(256, dtype=torch.float64, device='cuda') for _ in range(4)]
    for out in outs:
        eng.eval_device(d_theta.data_ptr(), 256, out.data_ptr())
    eng.sync()
    for out in outs:
        np.testing.assert_allclose(out.cpu().numpy(), full, rtol=1e-9)
        assert torch.equal(out, outs[0])                        # either lane, bit for bit
    eng.set_lanes(1)
    vega.close()


def test_the_two_forms_agree_on_the_standard_grids():
    """Joint auto + cross on model grid = data grid (auto keeps Q'); forcing the factored form gives the same chi2 to rounding
    (mocks as data, rescaled covariances and new data vectors: tests/test_quadratic_form_gpu.py runs both forms)."""
    from vega_amd import VegaInterface, synthetic
    vega = VegaInterface(None, problem=synth_joint_problem(), max_batch=64)
    eng = vega.engine
    theta = synthetic.walkers(eng.low.theta0, eng.names, 64, varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd'], seed=3)
    ref = eng.eval(theta)[0]
    assert eng.last_form() == 'q'
    full = eng.eval(theta, want_model=True)[0]
    eng.set_quadratic_form_kind('factored')
    for batch in (64, 8, 1):
        got = eng.eval(theta[:batch])[0]
        assert eng.last_form() == 'factored'
        np.testing.assert_allclose(got, ref[:batch], rtol=1e-11)
        np.testing.assert_allclose(got, full[:batch], rtol=1e-11)
    eng.set_quadratic_form_kind('auto')
    assert np.array_equal(eng.eval(theta)[0], ref)              # back on Q': the very same tape, bit for bit
    vega.close()


def test_marginalisation_on_the_finer_grid_through_the_engine(tmp_path, expected):
    from vega_amd import VegaInterface
    exp, names = expected
    prob = dmat_file_problem(tmp_path, marg_options=MARGINALIZATION_CASES['rtmax'])
    vega = VegaInterface(None, problem=prob, max_batch=2)
    assert vega.chi2() == pytest.approx(float(exp['marg/chi2']), rel=CHI2_RTOL)
    assert vega.log_lik() == pytest.approx(float(exp['marg/log_lik']), rel=1e-8)
    assert vega.chi2(dict(zip(names, exp['theta'][0]))) == pytest.approx(float(exp['marg/walker0/chi2']), rel=CHI2_RTOL)
    vega.close()
