"""Static tables of the additive xi terms (built once on the host).

* UV-background shot noise: A(tau) of Gontcho A Gontcho et al. 2014 eq. 19, tabulated as the reference does in
  ``CorrelationFunction.compute_shotnoise_A`` (reference vega/correlation_func.py:597-626) and interpolated
  linearly with ``left = A[0], right = 0`` (:628-647);
* DESI instrumental systematics: ``b * interp(rt)`` on the bins with 0 < rp < rp_binsize
  (reference vega/correlation_func.py:553-595), added to the non-peak component (vega/model.py:133-135).
"""
from functools import lru_cache

import numpy as np
from scipy.special import expn

DESI_INST_SYS_DEFAULT_AMP = 0.0003189935987295203     # reference correlation_func.py:577


@lru_cache(maxsize=2)
def shotnoise_a(n_tau=100, n_rho=10000):
    tau = np.linspace(0.01, 5, n_tau)
    rho = np.linspace(0.0001, 10, n_rho)
    d_rho = rho[1] - rho[0]
    weight = d_rho * np.exp(-rho) / rho
    a = np.array([-np.sum(weight * (expn(1, rho * np.sqrt(1 + (t / rho)**2)) - expn(1, rho * np.abs(1 - t / rho))))
                  for t in tau])
    return tau, a


def instrumental_systematics_template(item):
    """Per-bin template (amplitude 1) on the item's model grid."""
    pipe = item.core
    rp = pipe.r * pipe.mu
    rt = pipe.r * np.sqrt(1 - pipe.mu**2)
    table = item.inst_sys_table
    out = np.zeros(rt.shape)
    w = (rp > 0) & (rp < item.rp_binsize)
    x = rt[w]
    if np.any(x < table[0, 0]) or np.any(x > table[-1, 0]):
        raise ValueError('rt outside the instrumental-systematics table')
    out[w] = np.interp(x, table[:, 0], table[:, 1])
    return out
