"""Every alternative path the library can be switched to (the VMX_* environment knobs read at vmx_finalize / at the first
chi2-only call) is run against the same references as the default path: a fallback that is never exercised rots.

The knobs are read once per engine, so they are exercised in three sets of knobs that act on different stages and do
not mask each other (a set per engine, two fixtures: six engines instead of one per knob and fixture; with
VEGA_TEST_SINGLE_KNOBS=1 every knob gets its own engine - 26 cases, ~10 s on an MI355X box since the synthetic matrices are
kept per session - the way to find the culprit when a set fails).  What masks what:
without the quadratic form (VMX_NO_QUAD) or without the mapped buffers (VMX_NO_ZERO_COPY) a single walker's chi2 is never
added up on the host, and the done word is only waited on when it is not - so VMX_NO_HOST_REDUCE runs once with the done
word and once without; VMX_NO_CINV_TAPE (the full chain's chi2 by the C^-1 products instead of the tape) sits with VMX_NO_QUAD,
where every chi2 of a large batch comes from the full chain.  Per set an engine is built with the variables set and checked on
  (a) the reference's 8 golden walkers of the dense-matrix joint fixture and of the joint + metals fixture, tiled to 64
      (walkers that differ in every parameter: the per-walker loops, the MFMA products, the streaming kernels at B = 1 / 8);
  (b) a batch that shares its Arinyo / smoothing parameters, as a sampler's does (the table levels, two walkers per thread,
      the shared-W kernel, the FFTLog ring at B = 256), against the default engine and - three walkers - the CPU oracle.
Bars: chi2 1e-6, xi 1e-8 of the vector's scale (BASELINE.json north_star).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_problem, synth_joint_problem

pytestmark = pytest.mark.gpu

XI_RTOL = 1e-8
CHI2_RTOL = 1e-6

KNOB_SETS = [
    ('VMX_NO_TAB2', 'VMX_NO_GRAPH', 'VMX_NO_HOST_REDUCE', 'VMX_NO_XI_LEAN'),
    ('VMX_EXACT_MU', 'VMX_NO_HOST_REDUCE', 'VMX_NO_DONE_WORD', 'VMX_NO_STATIC_POLY', 'VMX_NO_XI_SUMS'),
    ('VMX_NO_QUAD', 'VMX_NO_CINV_TAPE', 'VMX_NO_ZERO_COPY', 'VMX_NO_PK_W', 'VMX_NO_STATIC_BINS'),
]
if os.environ.get('VEGA_TEST_SINGLE_KNOBS', '0') not in ('', '0'):
    KNOB_SETS = [(k,) for k in sorted({k for ks in KNOB_SETS for k in ks})]
SHARED_VARIED = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO', 'drp_QSO', 'bias_hcd',
                 'beta_hcd', 'L0_hcd', 'bias_eta_SiII(1190)', 'bias_eta_SiII(1193)', 'bias_eta_SiIII(1207)',
                 'bias_eta_SiII(1260)', 'bias_eta_CIV(eff)', 'bao_amp']

_DEFAULT = {}


def _problem(tag):
    return synth_joint_problem() if tag == 'joint_synth' else load_problem(tag)


def _shared_walkers(vega, n):
    from vega_amd import synthetic
    eng = vega.engine
    return synthetic.walkers(eng.low.theta0, eng.names, n, varied=SHARED_VARIED, seed=4711)


def _default_results(tag, n):
    """chi2 [n] and the models of three walkers from an engine without any knob (once per session), plus the oracle's
    chi2 of those three."""
    if tag not in _DEFAULT:
        from oracle import vega_cpu as oc
        from vega_amd import VegaInterface
        vega = VegaInterface(None, problem=_problem(tag), max_batch=n)
        theta = _shared_walkers(vega, n)
        chi2 = vega.chi2_batch(theta)
        pick = [0, n // 2, n - 1]
        models = vega.compute_model_batch(theta[pick])
        oracle = [oc.chi2(vega.problem, dict(zip(vega.engine.names, theta[i]))) for i in pick]
        np.testing.assert_allclose(chi2[pick], oracle, rtol=CHI2_RTOL)
        vega.close()
        _DEFAULT[tag] = (theta, chi2, pick, models)
    return _DEFAULT[tag]


def _assert_xi(got, ref, what):
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= XI_RTOL * scale, what


@pytest.mark.parametrize('knobs', KNOB_SETS, ids=['+'.join(k[4:] for k in ks) for ks in KNOB_SETS])
@pytest.mark.parametrize('tag', ['joint_synth', 'joint_metals'])
def test_every_fallback_path_against_the_fixtures(monkeypatch, tag, knobs):
    from vega_amd import VegaInterface
    n_shared = 256 if tag == 'joint_synth' else 64
    theta_s, chi2_s, pick, models_s = _default_results(tag, n_shared)
    for k in knobs:
        monkeypatch.setenv(k, '1')
    knob = '+'.join(knobs)
    exp = np.load(GOLDEN / f'expected_{tag}.npz')
    vega = VegaInterface(None, problem=_problem(tag), max_batch=n_shared)
    eng = vega.engine
    names = [str(n) for n in exp['param_names']]
    base = np.stack([eng.theta_from_params(dict(zip(names, row))) for row in exp['theta']])
    # (a) the reference's golden walkers: one at a time (twice: the second call may run against kept tables), as a batch of
    # 8, tiled to 64
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    assert vega.chi2() == pytest.approx(float(exp['fid/chi2']), rel=CHI2_RTOL)
    for i in (0, 5):
        pars = dict(zip(names, exp['theta'][i]))
        for _ in range(3):
            assert vega.chi2(pars) == pytest.approx(float(exp['chi2'][i]), rel=CHI2_RTOL)
    np.testing.assert_allclose(vega.chi2_batch(base), exp['chi2'], rtol=CHI2_RTOL)
    theta = np.tile(base, (8, 1))
    which = np.tile(np.arange(8), 8)
    chi2, status, model = eng.eval(theta, want_model=True)
    assert not status.any()
    np.testing.assert_allclose(chi2, exp['chi2'][which], rtol=CHI2_RTOL)
    np.testing.assert_allclose(vega.chi2_batch(theta), exp['chi2'][which], rtol=CHI2_RTOL)       # chi2 only: the quadratic form
    for b in (0, 13, 63):
        for name, sl in eng.model_slices.items():
            _assert_xi(model[b, sl], exp[f'walker{which[b]}/model/{name}'], f'{knob} {tag} walker {b} {name}')
    # (b) a sampler-like batch (shared Arinyo / smoothing parameters): table levels, paired walkers, shared-W kernel
    np.testing.assert_allclose(vega.chi2_batch(theta_s), chi2_s, rtol=1e-9)
    models = vega.compute_model_batch(theta_s[pick])
    for name in models:
        for j in range(len(pick)):
            _assert_xi(models[name][j], models_s[name][j], f'{knob} {tag} shared walker {pick[j]} {name}')
    # ... and through the device entry with the caller's promise that the batch shares them (level-2 tables)
    import torch
    eng.set_constant_nl_hint(True, gaussian=True)
    got = vega.chi2_batch_device(torch.from_numpy(theta_s).to('cuda:0')).cpu().numpy()
    np.testing.assert_allclose(got, chi2_s, rtol=1e-9)
    vega.close()
