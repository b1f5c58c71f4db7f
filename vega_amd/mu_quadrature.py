"""The node rule of the mu sums, restated in NumPy (documentation and tests; the engine builds the same table in
``vmx_set_template``, csrc/vegamx.hip, and reports it through ``vmx_get_mu_nodes``).

The reference projects P(k, mu) on Legendre polynomials with a midpoint sum over ``n_mu = 1000`` values of mu
(vega/power_spectrum.py:76-77, vega/pktoxi.py:138).  The engine has to return THAT sum - not the integral it approximates:
the two differ by ~1e-7, an order of magnitude above the parity bar.  For an integrand f that is smooth on
[a, b] = [lo, n_mu - hi] / n_mu the Euler-Maclaurin formula gives the midpoint sum over that range as

    sum_j f(mu_j) = (1/h) int_a^b f  -  (h/24) [f'(b) - f'(a)]  +  (7 h^3 / 5760) [f'''(b) - f'''(a)]  -  O(h^5 f^(5)),

with h = 1 / n_mu.  The rule keeps the first ``lo`` and the last ``hi`` midpoints as they are (the sharp features of the
model sit at the ends of the mu range), integrates the middle with ``panels`` Gauss-Legendre panels of ``n_gl`` nodes, and
forms three derivative terms (the formula's next one is - (31 h^5 / 967680) [D^5 f]) with ONE one-sided nine-point
finite-difference stencil per end - every piece is a fixed set of nodes
with fixed weights: ``sum_j W_j mu_j^(2n) P(k, mu_j)`` over 48 + 48 + 82 nodes instead of 1000 (round 2: 96 + 96 + 84 with
five-point stencils and two derivative terms).
"""
import numpy as np

N_MU, LO, HI, PANELS, N_GL, EPS, N_STENCIL = 1000, 48, 48, 2, 32, 1e-3, 9
# midpoint sum = (1/h) int f - (h/24) [Df] + (7 h^3/5760) [D^3 f] - (31 h^5/967680) [D^5 f] + O(h^7 D^7 f),  [g] = g(b) - g(a)
EM_ORDERS = (1, 3, 5)


def em_coefficient(order, h):
    return {1: -h / 24., 3: 7. * h**3 / 5760., 5: -31. * h**5 / 967680.}[order]


def fd_weights(n_points, max_order=5):
    """c[i][k]: D^k f(x) ~ sum_i c[i][k] f(x + i e) / e^k on the points x, x + e, .., x + (n_points - 1) e (Fornberg's
    recursion, the same sequence of operations as the engine's host code)."""
    c = np.zeros((n_points, max_order + 1))
    c1, c4 = 1.0, 0.0
    c[0, 0] = 1.0
    for i in range(1, n_points):
        mn = min(i, max_order)
        c2, c5, c4 = 1.0, c4, float(i)
        for j in range(i):
            c3 = float(i - j)
            c2 *= c3
            if j == i - 1:
                for k in range(mn, 0, -1):
                    c[i, k] = c1 * (k * c[i - 1, k - 1] - c5 * c[i - 1, k]) / c2
                c[i, 0] = -c1 * c5 * c[i - 1, 0] / c2
            for k in range(mn, 0, -1):
                c[j, k] = (c4 * c[j, k] - k * c[j, k - 1]) / c3
            c[j, 0] = c4 * c[j, 0] / c3
        c1 = c2
    return c


def extra_nodes(n_mu=N_MU, lo=LO, hi=HI, panels=PANELS, n_gl=N_GL, eps=EPS, n_stencil=N_STENCIL):
    """(mu, w) of the nodes that replace the midpoints lo .. n_mu - hi - 1, in the engine's order: the Gauss-Legendre
    panels, then per stencil point i the two entries (b - i eps, a + i eps) - ONE one-sided stencil per end carries all the
    derivative terms."""
    h = 1.0 / n_mu
    a, b = lo * h, (n_mu - hi) * h
    x, wx = np.polynomial.legendre.leggauss(n_gl)
    x, wx = x[::-1], wx[::-1]                   # (descending, as Newton's iteration from cos(...) produces them)
    mu, w = [], []
    for p in range(panels):
        pa, pb = a + (b - a) * p / panels, a + (b - a) * (p + 1) / panels
        mu += list(0.5 * (pb - pa) * x + 0.5 * (pa + pb))
        w += list(0.5 * (pb - pa) * wx / h)
    c = fd_weights(n_stencil)
    for i in range(n_stencil):
        wb = wa = 0.0
        for order in EM_ORDERS:
            scaled = c[i, order] / eps**order
            wb += em_coefficient(order, h) * (-scaled)          # backward stencil at b (odd derivative: minus)
            wa += -em_coefficient(order, h) * scaled
        mu += [b - i * eps, a + i * eps]
        w += [wb, wa]
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
