"""Correlations with different transform settings (num_bins_muk / old_fftlog / fht_lowring per `[model]` section, reference
vega/power_spectrum.py:52-58, vega/pktoxi.py:36-60): one engine per setting behind the same interface, against the oracle."""
import numpy as np
import pytest

from conftest import synth_joint_problem

pytestmark = pytest.mark.gpu

XI_RTOL = 1e-8
CHI2_RTOL = 1e-6


def _mixed(**changes):
    """The joint problem with the cross-correlation's pipelines on other settings than the auto-correlation's."""
    prob = synth_joint_problem()
    name = [n for n, it in prob.items.items() if it.tracer1.name != it.tracer2.name][0]
    item = prob.items[name]
    for pipe in [item.core] + [m.pipeline for m in item.metals]:
        for key, value in changes.items():
            setattr(pipe.pk if key == 'n_mu' else pipe.xi, key, value)
    return prob


@pytest.mark.parametrize('changes', [dict(fht_lowring=False), dict(old_fftlog=True), dict(n_mu=400),
                                     dict(n_mu=400, fht_lowring=False), dict(n_mu=2000)], ids=str)
def test_mixed_settings_match_the_oracle(changes):
    import torch
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface, synthetic
    from vega_amd.engine_group import EngineGroup
    prob = _mixed(**changes)
    vega = VegaInterface(None, problem=prob, max_batch=16)
    eng = vega.engine
    assert isinstance(eng, EngineGroup) and len(eng.children) == 2
    theta = np.vstack([eng.low.theta0[None, :], synthetic.walkers(eng.low.theta0, eng.names, 11, seed=29)])
    chi2, status = vega.chi2_batch(theta, return_status=True)
    assert not status.any()
    models = vega.compute_model_batch(theta[:2])
    for i in range(2):
        pars = dict(zip(eng.names, theta[i]))
        ref = oc.compute_model(prob, pars)
        for name in prob.items:
            assert np.abs(models[name][i] - ref[name]).max() <= XI_RTOL * np.abs(ref[name]).max(), (i, name)
        assert chi2[i] == pytest.approx(oc.chi2(prob, pars), rel=CHI2_RTOL)
    # the reference's scalar entry and the device entry agree with the batch
    assert vega.chi2(dict(zip(eng.names, theta[1]))) == pytest.approx(chi2[1], rel=1e-12)
    dev = vega.chi2_batch_device(torch.as_tensor(theta, device='cuda'))
    np.testing.assert_allclose(dev.cpu().numpy(), chi2, rtol=1e-12)
    vega.close()


def test_group_keeps_the_priors_once():
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface
    prob = _mixed(fht_lowring=False)
    prob.priors = {'beta_LYA': (prob.params['beta_LYA'] + 0.3, 0.1)}
    vega = VegaInterface(None, problem=prob, max_batch=4)
    pars = dict(prob.params)
    assert vega.chi2(pars) == pytest.approx(oc.chi2(prob, pars), rel=CHI2_RTOL)
    vega.close()


def test_two_voigt_tables():
    """`fvoigt_model` per correlation (reference vega/power_spectrum.py:310-340): two tables, two engines."""
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface
    from vega_amd.engine_group import EngineGroup
    from vega_amd.setup import load_fvoigt_table
    from conftest import GOLDEN
    prob = synth_joint_problem()
    table = load_fvoigt_table('fvoigt_models/Fvoigt_exp.txt', [GOLDEN / 'inputs'])
    other = table.copy()
    other[:, 1] = other[:, 1]**1.3
    for item, tab in zip(prob.items.values(), (table, other)):
        item.core.pk.hcd_model = 'fvoigt'
        item.core.pk.fvoigt_table = tab
    prob.params['L0_fvoigt'] = 0.8
    vega = VegaInterface(None, problem=prob, max_batch=4)
    assert isinstance(vega.engine, EngineGroup)
    pars = dict(prob.params)
    model = vega.compute_model(pars)
    ref = oc.compute_model(prob, pars)
    for name in prob.items:
        assert np.abs(model[name] - ref[name]).max() <= XI_RTOL * np.abs(ref[name]).max(), name
    assert vega.chi2(pars) == pytest.approx(oc.chi2(prob, pars), rel=CHI2_RTOL)
    vega.close()


def test_group_with_metals_freezes_static_metals_and_serves_device_walkers():
    """`freeze_static_metals` (reads the engine's metal plan) and `chi2_batch_device` in several chunks on a mixed-setting
    problem WITH metals: one engine per setting behind the same surface, against the oracle and the unfrozen group."""
    import torch
    from oracle import vega_cpu as oc
    from conftest import load_problem
    import copy
    from vega_amd import VegaInterface, synthetic
    from vega_amd.engine_group import EngineGroup
    prob = copy.deepcopy(load_problem('joint_metals'))
    name = [n for n, it in prob.items.items() if it.tracer1.name != it.tracer2.name][0]
    for pipe in [prob.items[name].core] + [m.pipeline for m in prob.items[name].metals]:
        pipe.xi.fht_lowring = False
    vega = VegaInterface(None, problem=prob, max_batch=8)
    assert isinstance(vega.engine, EngineGroup) and vega.engine.metal_plan == {}
    varied = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd', 'bias_eta_SiII(1190)', 'bias_eta_CIV(eff)']
    theta = synthetic.walkers(vega.engine.low.theta0, vega.engine.names, 20, varied=varied, seed=12)
    before = vega.chi2_batch(theta)
    for i in (0, 19):
        assert before[i] == pytest.approx(oc.chi2(prob, dict(zip(vega.engine.names, theta[i]))), rel=CHI2_RTOL)
    vega.freeze_static_metals()
    assert isinstance(vega.engine, EngineGroup)
    assert any(kind == 'basis' for entries in vega.engine.metal_plan.values() for kind, _ in entries)
    np.testing.assert_allclose(vega.chi2_batch(theta), before, rtol=1e-9)
    # device walkers, three chunks of max_batch: the chunks reuse the group's buffers, ordered by events only
    dev = vega.chi2_batch_device(torch.as_tensor(theta, device='cuda'))
    np.testing.assert_allclose(dev.cpu().numpy(), before, rtol=1e-9)
    vega.close()


def _chi2_models(vega, prob, theta):
    chi2, status = vega.chi2_batch(theta, return_status=True)
    assert not status.any()
    return chi2, vega.compute_model_batch(theta[:2])


def test_global_covariance_across_engines():
    """A global covariance (reference vega/vega_interface.py:295-304) over correlations that sit in different engines:
    r^T G r = sum_i r_i^T G_ii r_i + 2 sum_{i<j} r_i^T G_ij r_j - every engine with its diagonal block, the cross terms from
    the engines' models on the device.  (1) The same problem split by hand into two engines against ONE engine; (2) a
    mixed-setting problem (`fht_lowring = False` on the cross-correlation) against the oracle; device walkers; new data."""
    import torch
    from oracle import vega_cpu as oc
    from vega_amd import VegaInterface, synthetic
    from vega_amd.engine import Engine
    from vega_amd.engine_group import EngineGroup
    prob = synth_joint_problem(with_global_cov=True)
    names = list(prob.items)
    one = Engine(prob, max_batch=16)
    two = EngineGroup(prob, [[names[0]], [names[1]]], max_batch=16)
    theta = np.vstack([one.low.theta0[None, :], synthetic.walkers(one.low.theta0, one.names, 11, seed=29)])
    a, sa, ma = one.eval(theta, want_model=True)
    b, sb, mb = two.eval(theta, want_model=True)
    assert not sa.any() and not sb.any()
    np.testing.assert_allclose(b, a, rtol=1e-11)
    np.testing.assert_allclose(mb, ma, rtol=0, atol=1e-14 * np.abs(ma).max())        # (other launch groupings: other split-K sums)
    # mock pools with a per-walker row (Monte-Carlo fits in lock-step): the cross terms take the walker's row of every item
    rng = np.random.default_rng(3)
    for eng in (one, two):
        for name in names:
            d = np.asarray(prob.items[name].masked_data_vec)
            eng.set_mock_pool(name, d[None, :] * (1 + 0.01 * np.arange(1, 4)[:, None]))
        eng.set_mock_index(np.array([0, 2, -1, 1, 1, 0, 2, -1, 0, 1, 2, 0], dtype=np.int32))
    rng = None
    a2 = one.eval(theta)[0]
    b2 = two.eval(theta)[0]
    np.testing.assert_allclose(b2, a2, rtol=1e-11)
    assert a2[2] == pytest.approx(a[2], rel=1e-12) and abs(a2[0] - a[0]) > 1e-6 * abs(a[0])     # (row -1: the data)
    one.close(); two.close()

    prob = _mixed(fht_lowring=False)
    prob.global_cov = synth_joint_problem(with_global_cov=True).global_cov
    vega = VegaInterface(None, problem=prob, max_batch=8)
    assert isinstance(vega.engine, EngineGroup) and vega._use_global_cov
    theta = theta[:12]
    chi2, status = vega.chi2_batch(theta, return_status=True)          # (two chunks of max_batch)
    assert not status.any()
    for i in (0, 5, 11):
        pars = dict(zip(vega.engine.names, theta[i]))
        assert chi2[i] == pytest.approx(oc.chi2(prob, pars), rel=CHI2_RTOL)
    assert vega.chi2(dict(zip(vega.engine.names, theta[3]))) == pytest.approx(chi2[3], rel=1e-12)
    assert vega.log_lik(dict(zip(vega.engine.names, theta[3]))) == pytest.approx(oc.log_lik(prob, dict(zip(vega.engine.names, theta[3]))), rel=1e-6)
    dev = vega.chi2_batch_device(torch.as_tensor(theta, device='cuda'))
    np.testing.assert_allclose(dev.cpu().numpy(), chi2, rtol=1e-12)
    # other data (a Monte-Carlo mock installed as data: the cross terms read the engines' current data)
    import copy
    name = names[1]
    prob2 = copy.deepcopy(prob)
    prob2.items[name].data_vec = prob2.items[name].data_vec * 1.01
    vega.engine.set_data(name, prob2.items[name].masked_data_vec)
    moved = vega.chi2_batch(theta[:2])
    assert moved[1] == pytest.approx(oc.chi2(prob2, dict(zip(vega.engine.names, theta[1]))), rel=CHI2_RTOL)
    assert abs(moved[1] - chi2[1]) > 1e-5 * abs(chi2[1])             # (the data did change the answer)
    vega.close()
