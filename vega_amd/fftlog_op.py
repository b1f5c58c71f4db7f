"""Static linear operators of the xi stage, built once on the host.

The reference turns each multipole P_ell(k) into xi_ell(r) with ``mcfit.P2xi`` (an FFTLog:
reference vega/pktoxi.py:53,141) and then interpolates xi_ell in ln r with a not-a-knot cubic
spline (``scipy.interpolate.interp1d(kind='cubic')``, reference vega/pktoxi.py:144).  Both steps
are linear in P_ell, and their grids depend only on the template's k grid, so the engine folds
them into ONE dense operator per multipole,

    c_ell = OP_ell . P_ell        OP_ell = S . H_ell      (n_knots + 2) x n_k

that maps P_ell(k) straight to the uniform cubic B-spline coefficients of xi_ell on the FFTLog
output knots.  On the GPU this is a batched fp64 MFMA product over all walkers and pipelines; the
bins then evaluate four B-spline taps per multipole.

H_ell is written down from the FFTLog's circulant structure: with the zero-padded length N, the
low-ringing offset ln(xy) and the kernel's Mellin transform U_ell, the transform of a unit impulse
is one row of ``hfft(u)/N``.
"""
import functools

import numpy as np
from scipy.special import loggamma


def _mellin_u(ell, z):
    # Mellin transform of the spherical Bessel kernel j_ell: 2^(z-3/2) G((ell+z)/2) / G((3+ell-z)/2)
    return np.exp(np.log(2.0) * (z - 1.5) + loggamma(0.5 * (ell + z)) - loggamma(0.5 * (3 + ell - z)))


def fftlog_matrix(k, ell, q=1.5, lowring=True, pads=False):
    """(H, ln r): xi_ell(r_n) = sum_j H[n, j] P_ell(k_j) for mcfit.P2xi(k, l=ell, lowring=lowring)
    called with extrap=False (zero padding).  ``lowring=False`` (`fht_lowring = False`, reference pktoxi.py:42,53):
    x y = 1 and a real Nyquist term.

    ``pads=True`` (`fht_extrap = True`, reference pktoxi.py:41,141) returns (H, HL, HR, ln r): the operator's columns over
    the padded input as well - HL[:, t - 1] multiplies the left pad sample F[0] (F[1] / F[0])**(-t), HR[:, t - 1] the right
    pad sample F[-1] (F[-1] / F[-2])**t (t = 1 .. pad length), prefactor on the extended grid included."""
    k = np.asarray(k, dtype=float)
    n = k.size
    delta = np.log(k[-1] / k[0]) / (n - 1)
    N = 2 ** int(np.ceil(np.log2(2 * n)))
    lnxy = delta / np.pi * np.angle(_mellin_u(ell, q + 1j * np.pi / delta)) if lowring else 0.0
    m = np.arange(N // 2 + 1)
    u = _mellin_u(ell, q + 2j * np.pi * m / (N * delta)) * np.exp(-2j * np.pi * lnxy * m / (N * delta))
    if not lowring:
        u[N // 2] = u[N // 2].real

    # g = hfft(rfft(f) u)/N uses the forward sign twice, so it is a circular CORRELATION of the padded
    # input with c = hfft(u)/N: g[i] = sum_t f[t] c[(i + t) mod N]  (hence the reversed output grid)
    impulse = np.fft.hfft(u, n=N) / N
    pad = N - n
    pad_in, pad_out = pad // 2, pad - pad // 2
    rows = np.arange(n)[:, None] + pad_out
    cols = np.arange(n)[None, :] + pad_in
    circ = impulse[(rows + cols) % N]

    ln_r = lnxy - delta - np.log(k[::-1])
    pre = k ** (3 - q) / (2 * np.pi) ** 1.5
    post = (-1.0) ** (ell // 2) * np.exp(-q * ln_r)
    H = post[:, None] * circ * pre[None, :]
    if not pads:
        return H, ln_r
    t_left = np.arange(1, pad_in + 1)                   # padded index pad_in - t
    t_right = np.arange(1, N - pad_in - n + 1)          # padded index pad_in + n + t - 1
    pre_left = (k[0] * np.exp(-delta * t_left)) ** (3 - q) / (2 * np.pi) ** 1.5
    pre_right = (k[-1] * np.exp(delta * t_right)) ** (3 - q) / (2 * np.pi) ** 1.5
    HL = post[:, None] * impulse[(rows + (pad_in - t_left)[None, :]) % N] * pre_left[None, :]
    HR = post[:, None] * impulse[(rows + (pad_in + n + t_right - 1)[None, :]) % N] * pre_right[None, :]
    return H, HL, HR, ln_r


@functools.lru_cache(maxsize=4)
def notaknot_bspline_matrix(n):
    """S ((n+2) x n): knot values on a uniform grid -> uniform cubic B-spline coefficients of the
    not-a-knot interpolating spline.  Coefficient i multiplies the B-spline centred on knot i-1.
    (A function of the knot count alone - the four multipoles of an engine share it; cached, read-only.)"""
    M = np.zeros((n + 2, n + 2))
    idx = np.arange(n)
    M[idx, idx] = 1.0 / 6.0
    M[idx, idx + 1] = 4.0 / 6.0
    M[idx, idx + 2] = 1.0 / 6.0
    stencil = np.array([-1.0, 4.0, -6.0, 4.0, -1.0])
    M[n, 0:5] = stencil                 # third derivative continuous at knot 1
    M[n + 1, n - 3:n + 2] = stencil     # ... and at knot n-2
    rhs = np.zeros((n + 2, n))
    rhs[idx, idx] = 1.0
    S = np.linalg.solve(M, rhs)
    S.setflags(write=False)
    return S


def xi_operator(k, ell, lowring=True, extrap=False):
    """(OP, x0, h, n_knots): B-spline coefficients of xi_ell(ln r) from P_ell(k).  ``extrap``: OP has the columns of the
    power-law pads behind those of the samples, [n_k | left pads t = 1.. | right pads t = 1..] (:func:`fftlog_matrix`)."""
    if extrap:
        H, HL, HR, ln_r = fftlog_matrix(k, ell, lowring=lowring, pads=True)
        H = np.hstack([H, HL, HR])
    else:
        H, ln_r = fftlog_matrix(k, ell, lowring=lowring)
    n = ln_r.size
    h = (ln_r[-1] - ln_r[0]) / (n - 1)
    if np.max(np.abs(np.diff(ln_r) - h)) > 1e-10 * h:
        raise ValueError('the template k grid is not log-uniform; the engine needs uniform ln r knots')
    return notaknot_bspline_matrix(n) @ H, float(ln_r[0]), float(h), n


def hamilton_spline(k, pk, ell, kind):
    """B-spline coefficients of the odd-multipole terms of the cross-correlation.

    The reference computes the relativistic (``kind='rel'``, ell = 1, 3) and standard-asymmetry
    (``kind='asy'``, ell = 0, 2) contributions with its legacy in-repo FFTLog (Hamilton 2000) applied to the
    isotropic linear spectrum, followed by a cubic interpolating spline in ln r on knots shifted by half a step
    (reference vega/pktoxi.py:230-279, :321-382).  The linear spectrum is static, so the whole chain collapses
    to one coefficient vector per (component, ell).  Returns (coef [n + 2], x0, h).
    """
    k = np.asarray(k, dtype=float)
    pk = np.asarray(pk, dtype=float)
    n_pts = k.size
    span = np.log(k.max() / k[0])
    m = n_pts * np.fft.fftfreq(n_pts)
    n_pow = 1.0 if kind == 'rel' else 2.0
    q = 2 - n_pow - 0.5
    z = q + 2j * np.pi * m / span
    mu = ell + 0.5
    um = k[0] ** (-2j * np.pi * m / span) * 2 ** z * np.exp(loggamma((mu + 1 + z) / 2) - loggamma((mu + 1 - z) / 2))
    um[0] = um[0].real
    spec = np.fft.ifft(np.fft.fft(pk * k ** n_pow * np.sqrt(np.pi / 2)) * um)
    r = np.exp(-m * span / n_pts)
    order = np.argsort(r)
    r = r[order]
    xi = (spec[order] / r ** (3 - n_pow)).real
    xi[-1] = 0.0                                    # reference pktoxi.py:275
    h = span / n_pts
    x0 = np.log(r[0]) - h / 2                       # knots at ln r - dr/2 (reference pktoxi.py:276)
    return notaknot_bspline_matrix(n_pts) @ xi, float(x0), float(h)


def hamilton_spline_operator(k, ell, kind):
    """:func:`hamilton_spline` as a matrix: ``coef = OP @ pk`` ([n + 2] x [n]) - the chain is linear in the spectrum, which a
    caller-supplied spectrum per walker (`direct_pk`) needs; same operations, applied to the columns of the identity."""
    k = np.asarray(k, dtype=float)
    n_pts = k.size
    span = np.log(k.max() / k[0])
    m = n_pts * np.fft.fftfreq(n_pts)
    n_pow = 1.0 if kind == 'rel' else 2.0
    q = 2 - n_pow - 0.5
    z = q + 2j * np.pi * m / span
    mu = ell + 0.5
    um = k[0] ** (-2j * np.pi * m / span) * 2 ** z * np.exp(loggamma((mu + 1 + z) / 2) - loggamma((mu + 1 - z) / 2))
    um[0] = um[0].real
    spec = np.fft.ifft(np.fft.fft(np.diag(k ** n_pow * np.sqrt(np.pi / 2)), axis=0) * um[:, None], axis=0)
    r = np.exp(-m * span / n_pts)
    order = np.argsort(r)
    r = r[order]
    xi = (spec[order] / r[:, None] ** (3 - n_pow)).real
    xi[-1, :] = 0.0
    return notaknot_bspline_matrix(n_pts) @ xi


def hamilton_xi_operator(k, ell):
    """(OP, x0, h, n_knots) for the reference's legacy transform (``old_fftlog = True``): the Hamilton FFTLog of
    ``PktoXi.Pk2Mp`` (reference vega/pktoxi.py:230-279) applied to a multipole P_ell(k), followed by its cubic
    spline on the half-step-shifted ln r knots.  Like :func:`xi_operator` it is linear in P_ell; unlike the
    mcfit path its spline is evaluated outside the knot range by polynomial extension (``splev``), without error.
    """
    k = np.asarray(k, dtype=float)
    n_pts = k.size
    span = np.log(k.max() / k[0])
    m = n_pts * np.fft.fftfreq(n_pts)
    z = -0.5 + 2j * np.pi * m / span                 # q = 2 - n - 1/2 with n = 2
    mu = ell + 0.5
    um = k[0] ** (-2j * np.pi * m / span) * 2 ** z * np.exp(loggamma((mu + 1 + z) / 2) - loggamma((mu + 1 - z) / 2))
    um[0] = um[0].real
    pre = (-1.0) ** (ell // 2) / 2 / np.pi ** 2 * k ** 2 * np.sqrt(np.pi / 2)
    spec = np.fft.ifft(np.fft.fft(np.diag(pre), axis=0) * um[:, None], axis=0)
    r = np.exp(-m * span / n_pts)
    order = np.argsort(r)
    r = r[order]
    H = (spec[order] / r[:, None]).real              # / r^(3 - n)
    H[-1, :] = 0.0                                   # xi_loc[-1] = 0 (reference pktoxi.py:275)
    h = span / n_pts
    return notaknot_bspline_matrix(n_pts) @ H, float(np.log(r[0]) - h / 2), float(h), n_pts
