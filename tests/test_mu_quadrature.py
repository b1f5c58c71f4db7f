"""The node rule of the mu sums (vega_amd/mu_quadrature.py, ``vmx_set_mu_quadrature``).

CPU: with the oracle's P(k, mu) (the restatement of reference vega/power_spectrum.py:87-196) evaluated on the rule's
nodes, the weighted sums reproduce the reference's 1000-point midpoint sums (vega/pktoxi.py:138) over the wide parameter
ranges of the engine's tests - far from the fiducial point, weak and strong smoothing in either direction, HCD scales from
1 to 40 Mpc/h, mu^bv exponents from 1 to 2.  The bar: 1e-12 of the largest k^3 M_n (k^3: how a multipole enters xi) for
wavenumbers up to the engine's switch-over, four even moments, every core component of the joint fit.

GPU (``-m gpu``): the engine's node table equals this module's; results with the rule on and off agree to 1e-11 of the
model's scale for the joint and the joint + metals configurations, and the rule is what chi2-evaluations run by default.
"""
import numpy as np
import pytest

from conftest import load_problem

RANGES = {'ap': (0.7, 1.3), 'at': (0.7, 1.3), 'bias_eta_LYA': (-0.5, -0.01), 'beta_LYA': (0.1, 4.0),
          'beta_QSO': (0.05, 1.0), 'bias_hcd': (-0.2, 0.0), 'beta_hcd': (0.0, 1.5), 'L0_hcd': (1.0, 40.0),
          'sigma_velo_disp_lorentz_QSO': (0.0, 15.0), 'bao_amp': (0.0, 2.0),
          'sigmaNL_par': (2.0, 12.0), 'sigmaNL_per': (1.0, 8.0), 'par_sigma_smooth': (0.5, 6.0),
          'per_sigma_smooth': (0.5, 6.0), 'dnl_arinyo_q1': (0.3, 1.5), 'dnl_arinyo_kv': (0.3, 3.0),
          'dnl_arinyo_av': (0.1, 0.9), 'dnl_arinyo_bv': (1.0, 2.0), 'dnl_arinyo_kp': (8.0, 40.0),
          'bias_gamma': (0.0, 0.3), 'lambda_uv': (100.0, 600.0)}
K_NODE_MAX = 6.0        # 24 / (4 Mpc/h bins): what the engine uses for these configurations


class _NodeGrid:
    """PkGrid of the oracle on arbitrary mu values."""

    def __init__(self, k, mu):
        self.k = np.asarray(k, dtype=float)
        self.mu = np.asarray(mu, dtype=float)[:, None]
        self.k_par = self.k * self.mu
        self.k_trans = self.k * np.sqrt(np.maximum(1 - self.mu**2, 0.))


def test_rule_weights_are_a_quadrature_of_the_midpoint_sum():
    """Polynomials: the rule returns the 1000-point midpoint sums of mu^p (not the integrals) to rounding."""
    from vega_amd.mu_quadrature import node_rule, N_MU
    mu, w = node_rule()
    assert mu.size == 48 + 48 + 82 and np.all((mu > 0) & (mu <= 1))
    mid = (np.arange(N_MU) + 0.5) / N_MU
    for p in (0, 1, 2, 5, 8, 14):
        want = np.sum(mid**p)
        assert np.sum(w * mu**p) == pytest.approx(want, rel=2e-14)
        assert abs(want - N_MU / (p + 1)) > 1e-9 * want or p < 2        # (the integral is something else)


def test_rule_reproduces_the_references_mu_sums_over_wide_parameter_ranges(monkeypatch):
    from oracle import vega_cpu as oc
    from vega_amd.mu_quadrature import node_rule
    monkeypatch.setattr(oc, 'sinc', lambda x: np.sinc(np.asarray(x) / np.pi))       # (a node sits at mu = 1: sinc(0))
    prob = load_problem('joint')
    mu, w = node_rule()
    full, nodes = oc.PkGrid(prob.k, 1000), _NodeGrid(prob.k, mu)
    weight = prob.k**3 * (prob.k <= K_NODE_MAX)
    rng = np.random.default_rng(20261004)
    worst = 0.0
    for trial in range(6):
        pars = oc.local_params(prob)
        if trial:
            for name, (lo, hi) in RANGES.items():
                if name in pars:
                    pars[name] = rng.uniform(lo, hi)
        for item in prob.items.values():
            for peak, pk_lin in ((False, prob.pk_smooth), (True, prob.pk_full - prob.pk_smooth)):
                pp = dict(pars, peak=peak)
                exact = oc.power_spectrum(item.core, full, pk_lin, prob.pk_fid, pp)
                at_nodes = oc.power_spectrum(item.core, nodes, pk_lin, prob.pk_fid, pp)
                for n in range(4):
                    ref = np.sum(full.mu**(2 * n) * exact, axis=0)
                    got = np.sum(w[:, None] * nodes.mu**(2 * n) * at_nodes, axis=0)
                    scale = np.abs(ref * weight).max()
                    worst = max(worst, np.abs((got - ref) * weight).max() / scale)
    assert worst <= 1e-12, worst


def _box_draws(prob, n_draws, rng):
    """Parameter points over the FULL box of the guard (vega_amd.mu_quadrature.RULE_BOX) and the reference's prior
    limits of every other sampled-able parameter: the fiducial point, corners (every watched parameter at one of its
    limits), then uniform draws."""
    from oracle import vega_cpu as oc
    from vega_amd.defaults import DEFAULT_VALUES
    from vega_amd.mu_quadrature import rule_box
    base = oc.local_params(prob)
    box = rule_box(base.keys())
    limits = {n: DEFAULT_VALUES[n][0] for n in base if n in DEFAULT_VALUES and n not in box}
    limits.update(box)
    draws = [dict(base)]
    for i in range(1, n_draws):
        pars = dict(base)
        corner = i <= n_draws // 4
        for name, (lo, hi) in limits.items():
            if corner and name in box:
                pars[name] = lo if rng.random() < 0.5 else hi
            else:
                pars[name] = rng.uniform(lo, hi)
        draws.append(pars)
    return draws, box


def test_rule_over_the_full_guard_box_with_corners(monkeypatch):
    """>= 100 draws over the whole box the engine trusts the rule on (the reference's default_values.txt limits, wider
    for L0_hcd and dnl_arinyo_kp), corners included: all four moments, both items, peak and smooth, bar 1e-12 of the
    largest k^3 M_n.  Outside this box the engine runs the plain loop (vmx_set_mu_rule_box)."""
    from oracle import vega_cpu as oc
    from vega_amd.mu_quadrature import node_rule
    monkeypatch.setattr(oc, 'sinc', lambda x: np.sinc(np.asarray(x) / np.pi))
    prob = load_problem('joint')
    mu, w = node_rule()
    full, nodes = oc.PkGrid(prob.k, 1000), _NodeGrid(prob.k, mu)
    weight = prob.k**3 * (prob.k <= K_NODE_MAX)
    draws, box = _box_draws(prob, 104, np.random.default_rng(20261005))
    assert {'L0_hcd', 'sigmaNL_par', 'sigmaNL_per', 'par_sigma_smooth', 'per_sigma_smooth', 'dnl_arinyo_bv',
            'sigma_velo_disp_lorentz_QSO'} <= set(box)
    worst = 0.0
    for pars in draws:
        for item in prob.items.values():
            for peak, pk_lin in ((False, prob.pk_smooth), (True, prob.pk_full - prob.pk_smooth)):
                pp = dict(pars, peak=peak)
                exact = oc.power_spectrum(item.core, full, pk_lin, prob.pk_fid, pp)
                at_nodes = oc.power_spectrum(item.core, nodes, pk_lin, prob.pk_fid, pp)
                for n in range(4):
                    ref = np.sum(full.mu**(2 * n) * exact, axis=0)
                    got = np.sum(w[:, None] * nodes.mu**(2 * n) * at_nodes, axis=0)
                    scale = np.abs(ref * weight).max()
                    if scale > 0:
                        worst = max(worst, np.abs((got - ref) * weight).max() / scale)
    assert worst <= 1e-12, worst


@pytest.mark.gpu
def test_rule_on_and_off_over_the_guard_box_on_the_gpu_and_the_fallback_outside_it():
    """32 walkers drawn over the guard's box: rule on vs off, xi <= 1e-10 of the vector's scale; a walker pushed outside
    the box is evaluated by the plain loop (bit-identical with the rule switched off), and counted."""
    from vega_amd import VegaInterface
    prob = load_problem('joint')
    vega = VegaInterface(None, problem=prob, max_batch=32)
    eng = vega.engine
    draws, box = _box_draws(prob, 32, np.random.default_rng(99))
    assert set(eng.mu_rule_box) == set(box)
    theta = np.stack([eng.theta_from_params({k: v for k, v in p.items() if k in eng.low.slot}) for p in draws])
    for sub in (theta, theta[:5]):
        assert eng.set_mu_quadrature(True)
        c_rule, s_rule, m_rule = eng.eval(sub, want_model=True)
        assert not eng.set_mu_quadrature(False)
        c_loop, s_loop, m_loop = eng.eval(sub, want_model=True)
        eng.set_mu_quadrature(True)
        np.testing.assert_array_equal(s_rule, s_loop)
        ok = s_rule == 0
        assert ok.sum() >= sub.shape[0] // 2
        for name, sl in eng.model_slices.items():
            scale = np.abs(m_loop[ok][:, sl]).max(axis=1, keepdims=True)
            assert (np.abs(m_rule[ok][:, sl] - m_loop[ok][:, sl]) <= 1e-10 * scale).all(), name
    assert eng.debug_read(4, 0, 8)[7] == 0          # nobody left the box so far
    outside = theta[:6].copy()
    outside[:, eng.low.slot['L0_hcd']] = 55.0       # beyond the validated 40 Mpc/h
    eng.set_mu_quadrature(True)
    c_guard, _, m_guard = eng.eval(outside, want_model=True)
    assert eng.debug_read(4, 0, 8)[7] == 6
    eng.set_mu_quadrature(False)
    c_loop, _, m_loop = eng.eval(outside, want_model=True)
    eng.set_mu_quadrature(True)
    np.testing.assert_array_equal(m_guard, m_loop)
    np.testing.assert_array_equal(c_guard, c_loop)
    vega.close()


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['joint', 'joint_metals'])
def test_a_walker_s_mu_rule_does_not_depend_on_its_neighbours(tag):
    """The rule's applicability guard is the WALKER's (include/vegamx.h: vmx_set_mu_rule_box; vega_amd/parallel.py: chi2 does
    not depend on how the walkers are sharded): in the two-walkers-per-thread kernels (level-2 tables: k_pk_tab2; the
    shared-W groups of the metals: k_pk_w) a block whose walkers disagree runs both ways.  In-box walkers interleaved with
    out-of-box ones return, bit for bit, what they return among in-box walkers only - and the out-of-box ones what the plain
    loop gives them."""
    import torch
    from vega_amd import VegaInterface, synthetic
    vega = VegaInterface(None, problem=load_problem(tag), max_batch=128)
    eng = vega.engine
    varied = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd', 'beta_hcd', 'L0_hcd']
    inside = synthetic.walkers(eng.low.theta0, eng.names, 128, varied=varied, seed=77)
    mixed = inside.copy()
    stray = np.arange(128) % 2 == 1                     # every second walker: each block of the paired kernels is mixed
    stray[64:] = False                                  # ... in the first half of the batch; the second half is untouched
    mixed[stray, eng.low.slot['L0_hcd']] = 55.0         # beyond the validated 40 Mpc/h
    eng.set_constant_nl_hint(True, gaussian=True)       # (shared Arinyo / smoothing parameters: the level-2 table kernels)

    def device_chi2(theta):
        out = torch.zeros(theta.shape[0], dtype=torch.float64, device='cuda')
        eng.eval_device(torch.from_numpy(theta).cuda().data_ptr(), theta.shape[0], out.data_ptr())
        eng.sync()
        return out.cpu().numpy()
    alone = device_chi2(inside)
    assert int(eng.debug_read(4, 0, 5)[4]) == 2
    both = device_chi2(mixed)
    np.testing.assert_array_equal(both[~stray], alone[~stray])
    eng.set_mu_quadrature(False)
    loop = device_chi2(mixed)
    eng.set_mu_quadrature(True)
    # (switching the rule off re-evaluates the quadratic form's expansion point: the last bit of chi2 may move)
    np.testing.assert_allclose(both[stray], loop[stray], rtol=1e-13)
    assert np.all(both[stray] != alone[stray])
    vega.close()


@pytest.mark.gpu
def test_engine_nodes_and_the_rule_against_the_plain_loop():
    from vega_amd import VegaInterface, synthetic
    from vega_amd.mu_quadrature import extra_nodes
    for tag, batch in (('joint', 24), ('joint_metals', 12)):
        vega = VegaInterface(None, problem=load_problem(tag), max_batch=batch)
        eng = vega.engine
        mu, w = eng.mu_nodes()
        mu_ref, w_ref = extra_nodes()
        np.testing.assert_allclose(mu, mu_ref, rtol=0, atol=2e-16)
        np.testing.assert_allclose(w, w_ref, rtol=2e-12)
        varied = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO', 'bias_hcd', 'beta_hcd',
                  'L0_hcd', 'sigmaNL_par', 'sigmaNL_per', 'par_sigma_smooth', 'per_sigma_smooth', 'dnl_arinyo_q1',
                  'dnl_arinyo_bv', 'bias_eta_SiII(1260)', 'bias_eta_CIV(eff)']
        theta = synthetic.walkers(eng.low.theta0, eng.names, batch, varied=varied, seed=31, scale=0.15)
        for sub in (theta, theta[:3]):           # table mode / large-batch shape, and the small-batch shape
            assert eng.set_mu_quadrature(True)
            c_rule, s_rule, m_rule = eng.eval(sub, want_model=True)
            assert not eng.set_mu_quadrature(False)
            c_loop, s_loop, m_loop = eng.eval(sub, want_model=True)
            np.testing.assert_array_equal(s_rule, s_loop)
            ok = s_rule == 0
            assert ok.sum() >= sub.shape[0] // 2
            assert np.abs(m_rule[ok] - m_loop[ok]).max() <= 1e-11 * np.abs(m_loop[ok]).max()
            np.testing.assert_allclose(c_rule[ok], c_loop[ok], rtol=1e-9)
            eng.set_mu_quadrature(True)
        vega.close()
