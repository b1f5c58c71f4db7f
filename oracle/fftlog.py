"""TEST INFRASTRUCTURE (oracle) - CPU restatement of the FFTLog Hankel transform.

The reference transforms P_ell(k) -> xi_ell(r) with the third-party package
``mcfit`` (``mcfit.P2xi``; call sites: reference vega/pktoxi.py:5, :53, :141).
``mcfit`` is an unpinned dependency (reference pyproject.toml:36) whose source
is not under /root/reference and which is not installed in this image, so this
file restates its published algorithm (Hamilton 2000 FFTLog with the "low-ringing"
choice of the x*y product) from the specification in SURVEY.md Appendix B.

Pinning: the restatement is pinned through the reference's own call sites - with
this module standing in for ``mcfit`` the unmodified reference reproduces its
pinned log-likelihood (reference tests/test_vega.py:14) to 5e-10 relative; see
tests/golden/make_golden.py and tests/test_oracle.py.

Nothing in the product path (vega_amd/) imports this file.
"""
import numpy as np
from scipy.special import loggamma


def mellin_sph_bessel(ell):
    """Mellin transform U_ell(z) of the spherical-Bessel kernel used by P2xi.

    U_ell(z) = 2^(z-3/2) Gamma((ell+z)/2) / Gamma((3+ell-z)/2)
    """
    def mk(z):
        return np.exp(np.log(2) * (z - 1.5) + loggamma(0.5 * (ell + z))
                      - loggamma(0.5 * (3 + ell - z)))
    return mk


class P2xi:
    """Drop-in restatement of ``mcfit.P2xi(k, l=ell, lowring=True)``.

    ``r, xi = P2xi(k, l)(pk_ell, extrap=False)`` with
    xi_ell(r) = i^ell int k^2 dk / (2 pi^2) P_ell(k) j_ell(kr).
    """

    def __init__(self, k, l=0, q=1.5, N=2j, lowring=True):  # noqa: E741
        self.x = np.asarray(k, dtype=float)
        self.ell = int(l)
        self.q = q
        self.Nin = len(self.x)
        self.delta = np.log(self.x[-1] / self.x[0]) / (self.Nin - 1)

        if isinstance(N, complex):
            folds = int(np.ceil(np.log2(self.Nin * N.imag)))
            N = 2**folds
        if N < self.Nin:
            raise ValueError('N must be at least the input size')
        self.N = int(N)

        mk = mellin_sph_bessel(self.ell)
        if lowring and self.N % 2 == 0:
            self.lnxy = self.delta / np.pi * np.angle(mk(q + 1j * np.pi / self.delta))
        else:
            self.lnxy = 0.0

        self.y = np.exp(self.lnxy - self.delta) / self.x[::-1]

        m = np.arange(0, self.N // 2 + 1)
        self.u = mk(q + 2j * np.pi / self.N / self.delta * m)
        self.u = self.u * np.exp(-2j * np.pi * self.lnxy / self.N / self.delta * m)
        if not lowring and self.N % 2 == 0:
            self.u[self.N // 2] = self.u[self.N // 2].real

        # P2xi prefactor x^3/(2 pi)^1.5 and the generic x^-q tilt
        self.xfac = self.x**(3 - q) / (2 * np.pi)**1.5
        # (-1)^(ell/2) phase for even ell and the generic y^-q tilt
        phase = (-1)**(self.ell // 2) if self.ell % 2 == 0 else np.nan
        self.yfac = phase * self.y**(-q)

        npad = self.N - self.Nin
        self.pad_in = (npad // 2, npad - npad // 2)
        self.pad_out = (npad - npad // 2, npad // 2)

    def __call__(self, F, extrap=False):
        F = np.asarray(F, dtype=float)
        f = np.zeros(self.N)
        f[self.pad_in[0]:self.pad_in[0] + self.Nin] = self.xfac * F
        if extrap:
            # `fht_extrap = True` (reference vega/pktoxi.py:41,141): mcfit pads the INPUT by power-law extrapolation of its end
            # segments instead of zeros - left pad end * ratio**(-Npad .. -1) with ratio = F[1] / F[0], right pad
            # end * ratio**(1 .. Npad) with ratio = F[-1] / F[-2] (`mcfit.mcfit._pad(..., extrap=True, out=False)`) - and
            # applies its prefactor on the equally extended (geometric) x grid.  0 / 0 end segments give NaN, as there.
            # PARITY UNPINNED for this option: the reference holds no fixture or test with fht_extrap on; the restatement is
            # checked against the Hankel integral of a power-law-extended spectrum (tests/test_fftlog_quadrature.py).
            lo, hi = self.pad_in
            with np.errstate(all='ignore'):
                left = F[0] * (F[1] / F[0]) ** np.arange(-lo, 0)
                right = F[-1] * (F[-1] / F[-2]) ** np.arange(1, hi + 1)
            x_left = self.x[0] * np.exp(self.delta * np.arange(-lo, 0))
            x_right = self.x[-1] * np.exp(self.delta * np.arange(1, hi + 1))
            f[:lo] = x_left**(3 - self.q) / (2 * np.pi)**1.5 * left
            f[lo + self.Nin:] = x_right**(3 - self.q) / (2 * np.pi)**1.5 * right
        g = np.fft.hfft(np.fft.rfft(f) * self.u, n=self.N) / self.N
        g = g[self.pad_out[0]:self.pad_out[0] + self.Nin]
        return self.y, self.yfac * g

    def matrix(self):
        """The transform as an explicit (Nin x Nin) linear operator (rows = r)."""
        eye = np.eye(self.Nin)
        return np.stack([self(eye[j])[1] for j in range(self.Nin)], axis=1)
