"""Set-up time construction of the metal matrices (`new_metals = True`).

The reference builds, for every metal pair of a correlation, the matrix that moves the pair's correlation function
from the separations computed with the *true* absorber wavelengths to the ones the analysis *assumed* (every forest
pixel taken as the main absorber), from the stacked-delta weights of the forests and the redshift distribution of the
discrete tracer (reference vega/metals.py:389-752), instead of reading matrices picca computed (vega/data.py:556-687).
Nothing here runs per likelihood call: the result is a `(matrix, rp, rt, z)` tuple per pair, consumed exactly like a
matrix read from a metal file (the engine uploads it with `vmx_item_set_matrix(..., VMX_MAT_METAL, ...)`).

Two pieces come from picca, which is neither in the reference tree nor in this image: the rest wavelengths of the
absorbers (`picca.constants.ABSORBER_IGM`) and the comoving distance of `picca.constants.Cosmo`.  Both are restated
here from their published definitions (`ABSORBER_IGM`, `PiccaCosmo`); the histogram arithmetic is pinned against the
unmodified reference running on these same two stand-ins (tests/golden/expected_new_metals.npz), picca itself is not.
"""
from dataclasses import dataclass

import numpy as np
from scipy import sparse

from .tables import read_tables

# Rest wavelengths [Angstrom] (picca.constants.ABSORBER_IGM; a config `[metal-matrix] wavelength_<name>` overrides)
ABSORBER_IGM = {
    'MgI(2853)': 2852.96, 'MgII(2804)': 2803.5324, 'MgII(2796)': 2796.3511, 'FeII(2600)': 2600.1724835,
    'FeII(2587)': 2586.6495659, 'MnII(2577)': 2576.877, 'FeII(2383)': 2382.7641781, 'FeII(2374)': 2374.4603294,
    'FeII(2344)': 2344.2129601, 'AlIII(1863)': 1862.79113, 'AlIII(1855)': 1854.71829, 'AlII(1671)': 1670.7886,
    'FeII(1608)': 1608.4511, 'CIV(1551)': 1550.77845, 'CIV(eff)': 1549.06, 'CIV(1548)': 1548.2049,
    'SiII(1527)': 1526.70698, 'SiIV(1403)': 1402.77291, 'SiIV(1394)': 1393.76018, 'CII(1335)': 1334.5323,
    'SiII(1304)': 1304.3702, 'OI(1302)': 1302.1685, 'SiII(1260)': 1260.4221, 'NV(1243)': 1242.804,
    'NV(1239)': 1238.821, 'LYA': 1215.67, 'SiIII(1207)': 1206.500, 'NI(1200)': 1200., 'SiII(1193)': 1193.2897,
    'SiII(1190)': 1190.4158, 'OI(1039)': 1039.230, 'OVI(1038)': 1037.613, 'OVI(1032)': 1031.912, 'LYB': 1025.72,
}

SPEED_LIGHT = 299792.458    # km / s


class PiccaCosmo:
    """Comoving distance [Mpc/h] of picca.constants.Cosmo: H(z) on a 10000-point grid up to z = 10, its inverse
    integrated with the trapezoid rule, linear interpolation in between."""

    def __init__(self, Om, Ok=0., Or=0., wl=-1., H0=100., **_):
        n, z_max = 10000, 10.
        dz = z_max / n
        self.z = np.arange(n) * dz
        ol = 1. - Ok - Om - Or
        hubble = H0 * np.sqrt(ol * (1. + self.z)**(3. * (1. + wl)) + Ok * (1. + self.z)**2
                              + Om * (1. + self.z)**3 + Or * (1. + self.z)**4)
        inv = SPEED_LIGHT / hubble
        self.chi = np.concatenate([[0.], np.cumsum(0.5 * (inv[1:] + inv[:-1]) * dz)])
        self.hubble = hubble
        self.dist_hubble = inv

    def get_r_comov(self, z):
        z = np.asarray(z, dtype=float)
        if np.any(z < self.z[0]) or np.any(z > self.z[-1]):
            raise ValueError('redshift outside the tabulated range of the cosmology')
        return np.interp(z, self.z, self.chi)

    def get_dist_hubble(self, z):
        """c / H(z) [Mpc/h], linear interpolation on the same grid."""
        return np.interp(np.asarray(z, dtype=float), self.z, self.dist_hubble)


def _block_mean(vec, factor):
    """Mean over consecutive blocks of ``factor`` samples, the ragged tail dropped (reference metals.py:369-387)."""
    n = vec.size // factor
    return vec[:n * factor].reshape(n, factor).mean(axis=1)


@dataclass
class TracerSample:
    """Redshifts and weights one tracer contributes to the pair sums."""
    true_z: np.ndarray
    assumed_z: np.ndarray
    weights: np.ndarray


class MetalMatrixBuilder:
    """Everything of one correlation that its metal matrices share.

    tracers        (name, type) of the two main tracers
    weight_paths   stacked-delta file (LOGLAM, WEIGHT) of a forest / catalogue (Z) of a discrete tracer, per tracer
    grid           the model grid of the correlation (rp_min, rp_max, rt_max, n_rp, n_rt)
    config         the `[metal-matrix]` section (dict-like with string values)
    cosmo          object with get_r_comov(z)
    """

    def __init__(self, tracers, weight_paths, grid, config, cosmo, zmin=0., zmax=10.):
        self.tracers, self.paths = tracers, weight_paths
        self.grid, self.cfg, self.cosmo = grid, config, cosmo
        self.zmin, self.zmax = zmin, zmax
        self.n_rp, self.n_rt = grid.n_rp, grid.n_rt
        self.any_discrete = 'discrete' in (tracers[0][1], tracers[1][1])
        self._raw = {}
        self.last_factors = None

    # ---- inputs -----------------------------------------------------------------------------------------------
    def _getfloat(self, key, default=None):
        val = self.cfg.get(key, None)
        if val is None:
            if default is None:
                raise KeyError(f'[metal-matrix] needs {key}')
            return default
        return float(val)

    def _wavelength(self, name):
        val = self.cfg.get(f'wavelength_{name}', None)
        if val is not None:
            return float(val)
        if name not in ABSORBER_IGM:
            raise KeyError(f'no rest wavelength known for absorber {name}: set [metal-matrix] wavelength_{name}')
        return ABSORBER_IGM[name]

    def _forest(self, which):
        """(wavelength, weight) of the stacked deltas of forest tracer ``which`` (reference metals.py:389-416)."""
        key = ('forest', self.paths[which])
        if key not in self._raw:
            tab = read_tables(self.paths[which])[0]
            wave = 10**np.asarray(tab.data['LOGLAM'], dtype=float)
            weight = np.asarray(tab.data['WEIGHT'], dtype=float)
            rebin = self.cfg.get('rebin_factor', None)
            if rebin is not None:
                wave, weight = _block_mean(wave, int(rebin)), _block_mean(weight, int(rebin))
            self._raw[key] = (wave, weight)
        return self._raw[key]

    def _objects(self, which):
        """Weighted mean redshift and total weight of the occupied redshift bins of a catalogue
        (reference metals.py:418-449)."""
        key = ('objects', self.paths[which])
        if key not in self._raw:
            z_cat = np.asarray(read_tables(self.paths[which])[0].data['Z'], dtype=float)
            z_ref = self._getfloat('z_ref_objects', 2.25)
            z_evol = self._getfloat('z_evol_objects', 1.44)
            n_bins = int(self.cfg.get('z_bins_objects', 1000))
            w_cat = ((1. + z_cat) / (1. + z_ref))**(z_evol - 1.)
            w_sum, edges = np.histogram(z_cat, bins=n_bins, weights=w_cat)
            wz_sum, _ = np.histogram(z_cat, bins=edges, weights=w_cat * z_cat)
            keep = w_sum > 0
            self._raw[key] = (wz_sum[keep] / w_sum[keep], w_sum[keep])
        return self._raw[key]

    def _sample(self, which, true_absorber):
        name, kind = self.tracers[which]
        if kind != 'continuous':
            z, w = self._objects(which)
            return TracerSample(z, z, w)
        wave, w = self._forest(which)
        true_z = wave / self._wavelength(true_absorber) - 1.
        assumed_z = wave / self._wavelength(name) - 1.
        # weights were estimated with the assumed absorber's redshift evolution (reference metals.py:477-499)
        alpha_true = self._getfloat(f'alpha_{true_absorber}')
        alpha_assumed = self._getfloat(f'alpha_{name}', 2.9)
        return TracerSample(true_z, assumed_z, w * (1 + true_z)**(alpha_true + alpha_assumed - 2))

    # ---- pair sums --------------------------------------------------------------------------------------------
    def _separations(self, z1, z2):
        """Line-of-sight separations and mean distances of all (z1, z2) pairs (reference metals.py:451-475)."""
        if np.any(z1 < 0) or np.any(z2 < 0):
            raise ValueError('Attempting to compute distance to a negative redshift')
        d1, d2 = self.cosmo.get_r_comov(z1), self.cosmo.get_r_comov(z2)
        rp = np.subtract.outer(d1, d2).ravel()
        if not self.any_discrete:
            rp = np.abs(rp)
        return rp, (np.add.outer(d1, d2) / 2).ravel()

    def _pairs(self, absorber1, absorber2):
        s1, s2 = self._sample(0, absorber1), self._sample(1, absorber2)
        true_rp, true_dist = self._separations(s1.true_z, s2.true_z)
        assumed_rp, assumed_dist = self._separations(s1.assumed_z, s2.assumed_z)
        weights = np.multiply.outer(s1.weights, s2.weights).ravel()
        z_pair = np.add.outer(s1.assumed_z, s2.assumed_z) / 2.
        weights *= ((z_pair >= self.zmin) & (z_pair <= self.zmax)).ravel()
        z_true_pair = (np.add.outer(s1.true_z, s2.true_z) / 2.).ravel()
        return true_rp, true_dist, assumed_rp, assumed_dist, weights, z_true_pair

    def _rp_edges(self):
        return np.linspace(self.grid.rp_min, self.grid.rp_max, self.n_rp + 1)

    def _effective_rp_z(self, assumed_rp, weights, z_true_pair, edges):
        """Weighted mean assumed separation and true-absorber redshift per rp bin."""
        w_sum, _ = np.histogram(assumed_rp, bins=edges, weights=weights)
        w_rp, _ = np.histogram(assumed_rp, bins=edges, weights=weights * assumed_rp)
        w_z, _ = np.histogram(assumed_rp, bins=edges, weights=weights * z_true_pair)
        norm = w_sum + (w_sum == 0)
        return w_rp / norm, w_z / norm

    # ---- matrices ---------------------------------------------------------------------------------------------
    def rp_rt_matrix(self, absorber1, absorber2):
        """Full (rp, rt) matrix of a pair as the outer product of a line-of-sight and a transverse migration matrix
        (reference metals.py:501-655).  Returns (csr matrix [n_rp n_rt]^2, rp_eff, rt_eff, z_eff)."""
        true_rp, true_dist, assumed_rp, assumed_dist, weights, z_true = self._pairs(absorber1, absorber2)
        rp_edges = self._rp_edges()
        m_rp, _, _ = np.histogram2d(assumed_rp, true_rp, bins=(rp_edges, rp_edges), weights=weights)
        col = m_rp.sum(axis=0)
        m_rp /= col + (col == 0)

        # transverse migration: r_t scales with the ratio of assumed to true distance; pairs close along the line of
        # sight in truth dominate where it matters, each weighted by the solid angle ~ 1 / distance^2
        rt_edges = np.linspace(0, self.grid.rt_max, self.n_rt + 1)
        ratio_w, ratio_edges = np.histogram(assumed_dist / true_dist, bins=4 * rt_edges.size,
                                            weights=weights / true_dist**2 * (np.abs(true_rp) < 20.))
        ratio = (ratio_edges[1:] + ratio_edges[:-1]) / 2
        centres = (rt_edges[:-1] + rt_edges[1:]) / 2
        half = self.grid.rt_binsize / 2
        oversample = 7      # sub-bin offsets, evenly spaced over the bin
        offsets = np.linspace(-half, half * (1 - 2 / oversample), oversample)
        m_rt = np.zeros((self.n_rt, self.n_rt))
        for i, rt in enumerate(centres):
            sub = rt + offsets
            m_rt[:, i], _ = np.histogram(np.outer(ratio, sub).ravel(), bins=rt_edges,
                                         weights=np.outer(ratio_w, sub).ravel())
        col = m_rt.sum(axis=0)
        m_rt /= col + (col == 0)

        n = self.n_rp * self.n_rt
        # bin index = rt index + n_rt * rp index on both sides
        full = sparse.csr_matrix(np.einsum('ij,kl->ikjl', m_rp, m_rt).reshape(n, n))
        self.last_factors = (m_rp, m_rt)        # the engine applies the two factors instead of their Kronecker product

        rp_eff, z_eff = self._effective_rp_z(assumed_rp, weights, z_true, rp_edges)
        lo = np.arange(self.n_rt) * self.grid.rt_max / self.n_rt
        hi = (1 + np.arange(self.n_rt)) * self.grid.rt_max / self.n_rt
        rt_eff = (2 * (hi**3 - lo**3)) / (3 * (hi**2 - lo**2))        # area-weighted mean radius of the annulus
        idx = np.arange(n)
        return full, rp_eff[idx // self.n_rt], rt_eff[idx % self.n_rt], z_eff[idx // self.n_rt]

    def rp_matrix(self, absorber1, absorber2):
        """Line-of-sight-only matrix (`rp_only_metal_mats`; reference metals.py:657-752): [n_rp, n_rp], applied to every
        rt column alike.  Returns (dense matrix, rp_eff, rt_eff, z_eff) with the coordinates on the full grid."""
        true_rp, _, assumed_rp, _, weights, z_true = self._pairs(absorber1, absorber2)
        rp_edges = self._rp_edges()
        m_rp, _, _ = np.histogram2d(assumed_rp, true_rp, bins=(rp_edges, rp_edges), weights=weights)
        w_true, _ = np.histogram(true_rp, bins=rp_edges, weights=weights)
        m_rp *= ((w_true > 0) / (w_true + (w_true == 0)))[None, :]
        rp_eff, z_eff = self._effective_rp_z(assumed_rp, weights, z_true, rp_edges)
        rt_centres = np.arange(self.grid.rt_binsize / 2, self.grid.rt_max, self.grid.rt_binsize)
        idx = np.arange(self.n_rp * self.n_rt)
        return m_rp, rp_eff[idx // self.n_rt], rt_centres[idx % self.n_rt], z_eff[idx // self.n_rt]

    def expand_rp_matrix(self, m_rp):
        """The [n_rp n_rt]^2 operator equivalent to applying ``m_rp`` along rp for every rt
        (reference metals.py:354-358): kron(m_rp, I_rt) in the rt-fastest bin order."""
        return sparse.kron(sparse.csr_matrix(m_rp), sparse.identity(self.n_rt), format='csr')
