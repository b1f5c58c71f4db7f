"""Independent pin of the FFTLog stage (CPU): direct quadrature of the Hankel integral.

``mcfit`` is absent from /root/reference and from the image, so every reference fixture on the FFTLog path was
generated with ``oracle/fftlog.py`` standing in for it (tools/refshim/mcfit).  This file pins that restatement -
and the product's operator form ``vega_amd/fftlog_op.fftlog_matrix`` - against something that does not go through
either: the defining integral of ``mcfit.P2xi`` (reference call sites vega/pktoxi.py:53, :138-152),

    xi_ell(r) = i^ell  int k^2 dk / (2 pi^2)  P_ell(k) j_ell(k r),

evaluated with ``scipy.special.spherical_jn`` on a trapezoid in ln k that is 16 times finer than the template grid,
for ell = 0, 2, 4, 6 at every FFTLog output radius in [2, 280] Mpc/h.

* analytic spectrum (smooth broadband x BAO-like wiggles x Gaussian cut-off, evaluated exactly on both grids): the
  FFTLog must agree to 5e-11 of the vector's scale - measured 1.5e-12, the rounding of the quadrature sum;
* the PlanckDR16 template x Gaussian cut-off: the quadrature has to interpolate the 814 samples (cubic spline in
  ln k) while the FFTLog implies the trigonometric interpolant, and the BAO wiggles at k ~ 0.3 h/Mpc are sampled with
  ~7 points per period; the two readings of the samples differ by 5e-9 absolute (4e-9), bar 1e-8.
"""
import numpy as np
import pytest
from scipy.interpolate import CubicSpline
from scipy.special import spherical_jn

from conftest import GOLDEN

ELLS = (0, 2, 4, 6)
R_MIN, R_MAX = 2.0, 280.0
FINE = 16


def _template():
    from vega_amd.tables import read_tables
    table = read_tables(GOLDEN / 'inputs' / 'PlanckDR16.npz')[0]
    return np.asarray(table.data['K'], dtype=float), np.asarray(table.data['PK'], dtype=float)


def _analytic_pk(k, sigma):
    x = k / 0.02
    wiggles = 1 + 0.06 * np.sin(105. * k) * np.exp(-(8. * k)**2)
    return 2.5e4 * x / (1 + x**2.6) * wiggles * np.exp(-(k * sigma)**2)


def _quadrature(k_fine, pk_fine, r, ell):
    """Trapezoid in ln k of k^3 P(k) j_ell(kr) / (2 pi^2), with the i^ell sign of the even multipoles."""
    lnk = np.log(k_fine)
    integrand = k_fine[None, :]**3 * pk_fine[None, :] * spherical_jn(ell, k_fine[None, :] * r[:, None])
    integrand /= 2 * np.pi**2
    total = integrand.sum(axis=1) - 0.5 * (integrand[:, 0] + integrand[:, -1])
    return (-1)**(ell // 2) * total * (lnk[1] - lnk[0])


def _transforms(k, pk, ell):
    """(r, xi) from the oracle's P2xi and xi from the product's explicit operator, same radii."""
    from oracle.fftlog import P2xi
    from vega_amd.fftlog_op import fftlog_matrix
    r, xi_oracle = P2xi(k, l=ell)(pk, extrap=False)
    H, ln_r = fftlog_matrix(k, ell)
    np.testing.assert_allclose(np.exp(ln_r), r, rtol=1e-13)
    return r, xi_oracle, H @ pk


@pytest.mark.parametrize('sigma', [1.0, 3.0])
def test_fftlog_equals_the_hankel_integral_of_an_analytic_spectrum(sigma):
    k, _ = _template()
    lnk = np.log(k)
    step = (lnk[-1] - lnk[0]) / (k.size - 1) / FINE
    # the analytic spectrum continues below the template's first wavenumber: integrate it there too (the FFTLog pads
    # with zeros; the tail below k = 1e-4 h/Mpc contributes ~1e-17)
    k_fine = np.exp(np.arange(lnk[0] - 8.0, lnk[-1] + step / 2, step))
    pk_fine = _analytic_pk(k_fine, sigma)
    for ell in ELLS:
        r, xi_oracle, xi_op = _transforms(k, _analytic_pk(k, sigma), ell)
        sel = (r >= R_MIN) & (r <= R_MAX)
        ref = _quadrature(k_fine, pk_fine, r[sel], ell)
        scale = np.abs(ref).max()
        assert sel.sum() > 200
        assert np.abs(xi_oracle[sel] - ref).max() <= 5e-11 * scale, (ell, np.abs(xi_oracle[sel] - ref).max() / scale)
        assert np.abs(xi_op[sel] - ref).max() <= 5e-11 * scale, (ell, np.abs(xi_op[sel] - ref).max() / scale)


@pytest.mark.parametrize('sigma', [1.0, 2.0, 4.0])
def test_fftlog_of_the_template_against_quadrature(sigma):
    k, pk = _template()
    lnk = np.log(k)
    lnk_fine = np.linspace(lnk[0], lnk[-1], (k.size - 1) * FINE + 1)
    k_fine = np.exp(lnk_fine)
    pk_fine = CubicSpline(lnk, pk)(lnk_fine) * np.exp(-(k_fine * sigma)**2)
    for ell in ELLS:
        r, xi_oracle, xi_op = _transforms(k, pk * np.exp(-(k * sigma)**2), ell)
        sel = (r >= R_MIN) & (r <= R_MAX)
        ref = _quadrature(k_fine, pk_fine, r[sel], ell)
        # absolute: xi_0 is 0.1 - 0.35 at r = 2 Mpc/h here, i.e. <= 3e-8 of the ell = 0 scale; interpolation-limited
        assert np.abs(xi_oracle[sel] - ref).max() <= 1e-8, (ell, np.abs(xi_oracle[sel] - ref).max())
        assert np.abs(xi_op[sel] - ref).max() <= 1e-8, (ell, np.abs(xi_op[sel] - ref).max())
        # the operator form is the same arithmetic as the FFT form (the r^-1.5 tilt amplifies rounding below 1 Mpc/h)
        assert np.abs(xi_op[sel] - xi_oracle[sel]).max() <= 1e-12 * np.abs(xi_oracle[sel]).max()
        assert np.abs(xi_op - xi_oracle).max() <= 2e-9 * np.abs(xi_oracle).max()


def test_padded_length_is_immaterial():
    """N = 2048 is a reading of mcfit's default ``N = 2j`` that no pinned number can distinguish from
    4096 or 8192 - the transforms agree to rounding on the radii the bins use."""
    from oracle.fftlog import P2xi
    k, pk = _template()
    f = pk * np.exp(-(k * 1.0)**2)
    for ell in ELLS:
        r, base = P2xi(k, l=ell, N=2048)(f)
        sel = (r >= R_MIN) & (r <= R_MAX)
        for n in (4096, 8192):
            other = P2xi(k, l=ell, N=n)(f)[1]
            assert np.abs(other[sel] - base[sel]).max() <= 1e-11 * np.abs(base[sel]).max()


def test_power_law_padding_against_the_hankel_integral_of_the_extended_spectrum():
    """`fht_extrap = True` (reference vega/pktoxi.py:41,141): ``mcfit`` pads P_ell with power laws through its end
    segments.  No fixture of the reference exercises the option, so the restatement (``oracle/fftlog.P2xi(extrap=True)``) and
    the product's padded operator (``fftlog_matrix(pads=True)``) are held against the defining integral of exactly that
    extended function over the padded range: the quadrature is good to ~1e-5 there (the integrand still oscillates at the
    far end), the zero-padded transform is 15 - 50 times further from it.  Wrong exponents or pad order would be off by
    orders of magnitude."""
    from oracle.fftlog import P2xi
    from vega_amd.fftlog_op import fftlog_matrix
    # (a grid that ends at 50 h/Mpc, where this spectrum is still 1e-5 of its peak: on the template's own grid the pads are
    # below the quadrature's resolution)
    k = np.exp(np.linspace(np.log(1e-4), np.log(50.0), 814))
    lnk = np.log(k)
    d = (lnk[-1] - lnk[0]) / (k.size - 1)
    npad = 2 ** int(np.ceil(np.log2(2 * k.size))) - k.size
    lo, hi = npad // 2, npad - npad // 2

    def spectrum(kk):
        x = kk / 0.02
        return 2.5e4 * x / (1 + x**4)

    pk = spectrum(k)
    step = d / FINE
    k_fine = np.exp(np.arange(lnk[0] - lo * d, lnk[-1] + hi * d + step / 2, step))
    slope_lo, slope_hi = np.log(pk[1] / pk[0]) / d, np.log(pk[-1] / pk[-2]) / d
    pk_fine = np.where(k_fine < k[0], pk[0] * (k_fine / k[0])**slope_lo,
                       np.where(k_fine > k[-1], pk[-1] * (k_fine / k[-1])**slope_hi, spectrum(k_fine)))
    for ell in ELLS:
        r, xi = P2xi(k, l=ell)(pk, extrap=True)
        xi_zero = P2xi(k, l=ell)(pk, extrap=False)[1]
        H, HL, HR, ln_r = fftlog_matrix(k, ell, pads=True)
        assert HL.shape == (k.size, lo) and HR.shape == (k.size, hi)
        t_lo, t_hi = np.arange(1, lo + 1.0), np.arange(1, hi + 1.0)
        xi_op = H @ pk + HL @ (pk[0] * (pk[1] / pk[0])**(-t_lo)) + HR @ (pk[-1] * (pk[-1] / pk[-2])**t_hi)
        sel = (r >= R_MIN) & (r <= R_MAX)
        ref = _quadrature(k_fine, pk_fine, r[sel], ell)
        scale = np.abs(ref).max()
        err, err_zero = np.abs(xi[sel] - ref).max() / scale, np.abs(xi_zero[sel] - ref).max() / scale
        assert err <= 1e-4 and err <= 0.2 * err_zero, (ell, err, err_zero)
        assert np.abs(xi_op[sel] - xi[sel]).max() <= 1e-12 * np.abs(xi[sel]).max()
    # 0 / 0 end segments (a spectrum smoothed to exact zeros): NaN, as numpy's - and the reference's - arithmetic gives
    dead = pk * (k < 5.0)
    assert np.isnan(P2xi(k, l=0)(dead, extrap=True)[1]).all()
