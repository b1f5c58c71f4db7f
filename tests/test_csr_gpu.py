"""GPU tests of the CSR distortion path (``vmx_item_set_matrix_csr``): the reference keeps the distortion matrix as a
``scipy.sparse.csr_array`` (vega/data.py:342-346, product at vega/model.py:143-144).  Same matrix, both device
representations - CSR and dense - against each other, against the reference's fixture (which applied the CSR matrix) and
against the oracle, for every batch regime, full chain and quadratic form.
"""
import numpy as np
import pytest
from scipy import sparse

from conftest import GOLDEN, synth_joint_problem

pytestmark = pytest.mark.gpu

CHI2_RTOL = 1e-6


def _assert_xi(got, ref, mask, what):
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 1e-8 * scale, what
    np.testing.assert_allclose(got[mask], ref[mask], rtol=1e-8, atol=1e-12 * scale, err_msg=what)


@pytest.mark.parametrize('batch', [1, 3, 8, 40])
def test_csr_and_dense_representations_of_the_same_matrix(batch):
    from vega_amd import VegaInterface
    prob = synth_joint_problem()
    for item in prob.items.values():
        item.distortion = sparse.csr_array(item.distortion)             # what the reference holds
    exp = np.load(GOLDEN / 'expected_joint_synth.npz')
    dense = VegaInterface(None, problem=prob, max_batch=batch, csr_threshold=0.0)
    csr = VegaInterface(None, problem=prob, max_batch=batch, csr_threshold=1.1)
    assert dense.engine.csr_items == [] and csr.engine.csr_items == list(prob.items)
    names = [str(n) for n in exp['param_names']]
    base = np.stack([csr.engine.theta_from_params(dict(zip(names, row))) for row in exp['theta']])
    reps = -(-batch // 8)
    theta = np.tile(base, (reps, 1))[:batch]
    which = np.tile(np.arange(8), reps)[:batch]
    c_full, st, m_csr = csr.engine.eval(theta, want_model=True)
    d_full, _, m_dense = dense.engine.eval(theta, want_model=True)
    assert not st.any()
    np.testing.assert_allclose(c_full, d_full, rtol=1e-11)
    assert np.abs(m_csr - m_dense).max() <= 1e-12 * np.abs(m_dense).max()
    np.testing.assert_allclose(c_full, exp['chi2'][which], rtol=CHI2_RTOL)
    for b in range(batch):
        for name, sl in csr.engine.model_slices.items():
            _assert_xi(m_csr[b, sl], exp[f'walker{which[b]}/model/{name}'], prob.items[name].model_mask, f'csr {b} {name}')
    # chi2-only: the quadratic form built from the CSR matrix
    assert csr.engine.quadratic_form
    np.testing.assert_allclose(csr.engine.eval(theta)[0], c_full, rtol=1e-10)
    csr.close()
    dense.close()


def test_sparse_matrix_takes_the_csr_path_by_default():
    """A banded distortion matrix (3 % non-zeros) stays in CSR form without being asked to; against the oracle."""
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface
    prob = synth_joint_problem()
    for item in prob.items.values():
        dm = item.distortion.copy()
        rt = item.model_grid.rt
        dm[np.abs(rt[:, None] - rt[None, :]) > 6.0] = 0.0          # keep neighbouring transverse bins only
        np.fill_diagonal(dm, 1.0)
        item.distortion = sparse.csr_array(dm)
        assert item.distortion.nnz < 0.1 * dm.size
    vega = VegaInterface(None, problem=prob, max_batch=16)
    assert vega.engine.csr_items == list(prob.items)
    pars = {'ap': 1.02, 'at': 0.97, 'beta_LYA': 1.8, 'bias_hcd': -0.04}
    assert vega.chi2(pars) == pytest.approx(oc.chi2(prob, pars), rel=CHI2_RTOL)
    model = vega.compute_model(pars)
    ref = oc.compute_model(prob, pars)
    for name, item in prob.items.items():
        _assert_xi(model[name], ref[name], item.model_mask, name)
    theta = np.tile(vega.engine.theta_from_params(pars), (16, 1))
    theta[:, vega.engine.low.slot['ap']] = np.linspace(0.95, 1.05, 16)
    np.testing.assert_allclose(vega.engine.eval(theta)[0], vega.engine.eval(theta, want_model=True)[0], rtol=1e-10)
    vega.close()
