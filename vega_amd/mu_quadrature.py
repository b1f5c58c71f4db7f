"""The node rule of the mu sums, restated in NumPy (documentation and tests; the engine builds the same table in
``vmx_set_template``, csrc/vegamx.hip, and reports it through ``vmx_get_mu_nodes``).

The reference projects P(k, mu) on Legendre polynomials with a midpoint sum over ``n_mu = 1000`` values of mu
(vega/power_spectrum.py:76-77, vega/pktoxi.py:138).  The engine has to return THAT sum - not the integral it approximates:
the two differ by ~1e-7, an order of magnitude above the parity bar.  For an integrand f that is smooth on
[a, b] = [lo, n_mu - hi] / n_mu the Euler-Maclaurin formula gives the midpoint sum over that range as

    sum_j f(mu_j) = (1/h) int_a^b f  -  (h/24) [f'(b) - f'(a)]  +  (7 h^3 / 5760) [f'''(b) - f'''(a)]  -  O(h^5 f^(5)),

with h = 1 / n_mu.  The rule keeps the first ``lo`` and the last ``hi`` midpoints as they are (the sharp features of the
model sit at the ends of the mu range), integrates the middle with ``panels`` Gauss-Legendre panels of ``n_gl`` nodes, and
forms the two derivative terms with one-sided five-point finite-difference stencils - every piece is a fixed set of nodes
with fixed weights: ``sum_j W_j mu_j^(2n) P(k, mu_j)`` over 96 + 96 + 84 nodes instead of 1000.
"""
import numpy as np

N_MU, LO, HI, PANELS, N_GL, EPS1, EPS3 = 1000, 96, 96, 2, 32, 1e-3, 2e-3
# five-point one-sided stencils: f'(x) ~ sum c1_i f(x + i e) / e,  f'''(x) ~ sum c3_i f(x + i e) / e^3
C1 = np.array([-25., 48., -36., 16., -3.]) / 12.
C3 = np.array([-5., 18., -24., 14., -3.]) / 2.


def extra_nodes(n_mu=N_MU, lo=LO, hi=HI, panels=PANELS, n_gl=N_GL, eps1=EPS1, eps3=EPS3):
    """(mu, w) of the nodes that replace the midpoints lo .. n_mu - hi - 1, in the engine's order: the Gauss-Legendre
    panels, then per stencil point i the four entries (first derivative at b, at a, third derivative at b, at a)."""
    h = 1.0 / n_mu
    a, b = lo * h, (n_mu - hi) * h
    x, wx = np.polynomial.legendre.leggauss(n_gl)
    x, wx = x[::-1], wx[::-1]                   # (descending, as Newton's iteration from cos(...) produces them)
    mu, w = [], []
    for p in range(panels):
        pa, pb = a + (b - a) * p / panels, a + (b - a) * (p + 1) / panels
        mu += list(0.5 * (pb - pa) * x + 0.5 * (pa + pb))
        w += list(0.5 * (pb - pa) * wx / h)
    t1, t3 = h / 24., 7. * h**3 / 5760.
    for i in range(5):
        mu += [b - i * eps1, a + i * eps1, b - i * eps3, a + i * eps3]
        w += [-t1 * (-C1[i] / eps1), t1 * (C1[i] / eps1), t3 * (-C3[i] / eps3**3), -t3 * (C3[i] / eps3**3)]
    return np.array(mu), np.array(w)


def node_rule(n_mu=N_MU, lo=LO, hi=HI, **kw):
    """(mu, w) of the whole rule: the kept midpoints with unit weight, then the extra nodes."""
    kept = np.concatenate([np.arange(lo), np.arange(n_mu - hi, n_mu)])
    mu_x, w_x = extra_nodes(n_mu, lo, hi, **kw)
    return np.concatenate([(kept + 0.5) / n_mu, mu_x]), np.concatenate([np.ones(kept.size), w_x])


# ---- applicability guard ---------------------------------------------------------------------------------------------
# Parameters that shape P(k, mu) other than through polynomials in mu^2 (the Kaiser / HCD amplitudes, which the panels
# integrate exactly): name or name prefix -> the interval the rule is validated on.  The limits are the reference's prior
# limits (vega/parameters/default_values.txt, restated in vega_amd/defaults.py), widened where the tests go further
# (tests/test_mu_quadrature.py draws over exactly this table, corners included).  A walker outside takes the plain
# 1000-point loop (include/vegamx.h: vmx_set_mu_rule_box).
RULE_BOX = {
    'L0_hcd': (0.0, 40.0),
    'sigmaNL_par': (0.0, 15.0), 'sigmaNL_per': (0.0, 15.0),
    'par_sigma_smooth': (0.0, 10.0), 'per_sigma_smooth': (0.0, 10.0),
    'sigma_velo_disp_gauss': (0.0, 15.0), 'sigma_velo_disp_lorentz': (0.0, 15.0),
    'growth_rate': (0.0, 2.0),               # (sigmaNL_per follows from sigmaNL_par / (1 + f) when only one is given)
    'dnl_arinyo_q1': (0.0, 2.0), 'dnl_arinyo_q2': (-1.0, 1.0), 'dnl_arinyo_kv': (0.1, 4.0), 'dnl_arinyo_av': (0.1, 1.0),
    'dnl_arinyo_bv': (1.0, 2.0), 'dnl_arinyo_kp': (7.0, 40.0),
}


def rule_box(names):
    """{parameter name: (lo, hi)} for the names of ``names`` the guard watches (exact name, or prefix + '_<tracer>')."""
    out = {}
    for name in names:
        for key, limits in RULE_BOX.items():
            if name == key or name.startswith(key + '_'):
                out[name] = limits
    return out
