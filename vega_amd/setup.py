"""Host-side static state: configs -> a :class:`Problem` the engine (and the oracle) consume.

This is the init-time glue the reference spreads over ``VegaInterface.__init__``
(reference vega/vega_interface.py:36-206), ``CorrelationItem`` (vega/correlation_item.py),
``Data`` (vega/data.py:30-134, :285-473, :475-687), ``Coordinates`` (vega/coordinates.py) and
the constructors of ``PowerSpectrum`` / ``PktoXi`` / ``CorrelationFunction`` / ``Metals`` /
``BroadbandPolynomials``.  It reads the same ``.ini`` keys and produces plain arrays and option
records; nothing here is evaluated per likelihood call.

Options outside the hot-path scope (new_metals matrix construction, a parameter-dependent mock binning
kernel, ...) raise ``NotImplementedError`` instead of being silently ignored.
"""
import configparser
import os
from dataclasses import dataclass, field
from functools import lru_cache
from pathlib import Path

import numpy as np
from scipy import sparse
from scipy.integrate import quad
from scipy.interpolate import interp1d

from .tables import find_file, read_tables

TRUE_WORDS = ('True', 'true', 't', 'y', 'yes')

# blinding (reference vega/data.py:9, vega/utils.py:16-25)
BLINDING_STRATEGIES = ('desi_dr3',)
BLIND_FIXED_PARS = ('ap_full', 'at_full', 'aiso_full', 'epsilon_full', 'phi_full')
VEGA_BLINDED_PARS = {'phi_smooth': ('all',), 'growth_rate': ('all',)}


def init_blinding(items, sample_params):
    """Parameter-level blinding checks of the reference (vega/vega_interface.py:853-886): returns (blind, names of the
    sampled parameters that must carry an offset).  The offsets themselves live in collaboration files the
    reference names per strategy (vega/utils.py:320-372); it has none for `desi_dr3`, so sampling a blinded parameter
    on such data is the reference's ValueError."""
    strat = None
    blind = False
    for item in items.values():
        if getattr(item, 'blind', False):
            blind = True
            if strat is None:
                strat = item.blinding_strat
            elif strat != item.blinding_strat:
                raise ValueError('Different blinding strategies found in the data sets.')
    if not blind:
        return False, []
    sampled = sample_params.get('limits', {})
    blind_pars = []
    for par in sampled:
        if par in BLIND_FIXED_PARS:
            raise ValueError(f'Running on blind data, parameter {par} must be fixed.')
        if par not in VEGA_BLINDED_PARS:
            continue
        tracers = VEGA_BLINDED_PARS[par]
        # reference correlation_item.py:153-173
        if any('all' in tracers or any(t in item.tracer1.name or t in item.tracer2.name for t in tracers)
               for item in items.values()):
            blind_pars.append(par)
    if blind_pars:
        # reference utils.get_blinding: only desi_y1 / desi_y3 are known there, both without an offsets file
        if strat not in ('desi_y1', 'desi_y3'):
            raise ValueError(f'Unknown blinding version: {strat}.')
        blind_pars = []
    if 'bias_QSO' in sampled and 'beta_QSO' in sampled:
        raise ValueError('Running on blind data and sampling bias_QSO and beta_QSO.')
    return True, blind_pars


def blinding_transform(names, offsets):
    """(scale, shift) per parameter column for blinding offsets {name: v}: p += pi - exp(v^2) on the blinded names
    and the full-shape scale parameters pinned to 1 (reference vega/utils.py:375-393, vega_interface.py:413-419)."""
    scale, shift = np.ones(len(names)), np.zeros(len(names))
    for i, name in enumerate(names):
        if name in offsets:
            shift[i] = np.pi - np.exp(offsets[name]**2)
        if name in BLIND_FIXED_PARS:
            scale[i], shift[i] = 0.0, 1.0
    unknown = set(offsets) - set(names)
    if unknown:
        raise KeyError(f'blinding offsets for parameters the model does not have: {sorted(unknown)}')
    return scale, shift


# --------------------------------------------------------------------------------------
# coordinate grids and scale-cut masks (reference vega/coordinates.py:8-73, :146-182)
# --------------------------------------------------------------------------------------
class Grid:
    """(rp, rt, z) bin centres of one correlation plus the regular grid used for cuts."""

    def __init__(self, rp_min, rp_max, rt_max, n_rp, n_rt, rp=None, rt=None, z=None,
                 r=None, mu=None):
        self.rp_min, self.rp_max, self.rt_max = float(rp_min), float(rp_max), float(rt_max)
        self.n_rp, self.n_rt = int(n_rp), int(n_rt)
        self.rp_binsize = (self.rp_max - self.rp_min) / self.n_rp
        self.rt_binsize = self.rt_max / self.n_rt

        rp_c = np.arange(self.rp_min + self.rp_binsize / 2, self.rp_max, self.rp_binsize)
        rt_c = np.arange(self.rt_binsize / 2, self.rt_max, self.rt_binsize)
        self.rp_regular = np.repeat(rp_c, rt_c.size)
        self.rt_regular = np.tile(rt_c, rp_c.size)
        self.r_regular = np.sqrt(self.rp_regular**2 + self.rt_regular**2)
        self.mu_regular = np.zeros_like(self.r_regular)
        pos = self.r_regular > 0
        self.mu_regular[pos] = self.rp_regular[pos] / self.r_regular[pos]

        self.rp = self.rp_regular if rp is None else np.asarray(rp, dtype=float)
        self.rt = self.rt_regular if rt is None else np.asarray(rt, dtype=float)
        self.r = np.sqrt(self.rp**2 + self.rt**2) if r is None else np.asarray(r, dtype=float)
        if mu is None:
            self.mu = np.zeros_like(self.r)
            pos = self.r > 0
            self.mu[pos] = self.rp[pos] / self.r[pos]
        else:
            self.mu = np.asarray(mu, dtype=float)
        self.z = None if z is None else np.asarray(z, dtype=float)

    @property
    def size(self):
        return self.rp.size

    def scale_cut_mask(self, cuts):
        """Strict-inequality cuts on the regular grid (reference vega/coordinates.py:146-182)."""
        def get(key, default):
            return float(cuts.get(key, default)) if cuts is not None else default
        m = (self.rp_regular > get('rp-min', 0.)) & (self.rt_regular > get('rt-min', 0.))
        m &= self.r_regular > get('r-min', 10.)
        m &= (self.rp_regular < get('rp-max', 300.)) & (self.rt_regular < get('rt-max', 300.))
        m &= self.r_regular < get('r-max', 180.)
        m &= (self.mu_regular > get('mu-min', -1.)) & (self.mu_regular < get('mu-max', 1.))
        return m


# --------------------------------------------------------------------------------------
# growth factor (reference vega/utils.py:128-227, vega/correlation_func.py:372-403)
# --------------------------------------------------------------------------------------
def _hubble(z, om, ode):
    return np.sqrt(om * (1 + z)**3 + ode + (1 - om - ode) * (1 + z)**2)


@lru_cache(maxsize=32)
def _growth_interp(om, ode):
    z_grid = np.linspace(0, 10, 1000)
    growth = np.zeros(z_grid.size)
    for i, z in enumerate(z_grid):
        integral = quad(lambda a: 1. / (a * _hubble(1 / a - 1, om, ode))**3, 0, 1 / (1 + z))[0]
        growth[i] = 2.5 * om * _hubble(z, om, ode) * integral
    return interp1d(z_grid, growth, kind='cubic')


def growth_squared(z, z_fid, om, ode):
    """[D(z)/D(z_fid)]^2, or the matter-dominated ((1+z_fid)/(1+z))^2 when Omega_de is None."""
    if ode is None:
        return ((1 + z_fid) / (1. + z))**2
    g = _growth_interp(om, ode)
    return (g(z) / g(z_fid))**2


@lru_cache(maxsize=32)
def _growth_interp_old(om, ode):
    z_grid = 5. * np.arange(100, dtype=float) / 99
    growth = np.zeros(z_grid.size)
    for i, z in enumerate(z_grid):
        integral = quad(lambda a: 1. / (a * _hubble(1 / a - 1, om, ode))**3, 0, 1 / (1 + z))[0]
        growth[i] = 2.5 * om * _hubble(z, om, ode) * integral
    return interp1d(z_grid, growth)


def growth_squared_old(z, z_fid, om, ode):
    """`old_growth_func = True` (reference vega/correlation_func.py:405-444, deprecated there): the growth factor
    tabulated at 100 redshifts up to 5 and interpolated linearly."""
    g = _growth_interp_old(om, ode)
    return (g(z) / g(z_fid))**2


# --------------------------------------------------------------------------------------
# option records
# --------------------------------------------------------------------------------------
@dataclass
class Tracer:
    name: str
    type: str  # 'continuous' | 'discrete'


@dataclass
class PkOptions:
    """What ``PowerSpectrum`` reads from a [model] / [metals] section
    (reference vega/power_spectrum.py:43-68,76,128-194)."""
    bin_size_rp: float = 4.0
    bin_size_rt: float = 4.0
    use_gk: bool = True
    skip_nl_in_peak: bool = False
    damping_scale: float = None
    damping_power: int = 2
    hcd_model: str = None        # None | 'Rogers' | 'fvoigt' | 'sinc'
    fvoigt_table: np.ndarray = None
    uvb: bool = False
    heii: bool = False
    small_scale_nl: str = None   # None | 'arinyo' | 'mcdonald'
    fullshape_smoothing: str = None  # None | 'gauss' | 'exp'
    velocity_dispersion: str = None  # None | 'gauss' | 'lorentz'
    mock_bin_size: float = None      # 'mock-bin-size' (reference power_spectrum.py:143-160)
    mock_los_smoothing: str = None   # None | 'growth' | 'amplitude' | 'only-los'
    n_mu: int = 1000


@dataclass
class XiOptions:
    """What ``PktoXi`` and ``CorrelationFunction`` read
    (reference vega/pktoxi.py:39-45, vega/correlation_func.py:49-111,316-321)."""
    ell_max: int = 6
    old_fftlog: bool = False
    single_multipole: int = -1
    new_bias_evol: bool = False      # separate redshifts of the two tracers in a cross-correlation
    evol_model: dict = field(default_factory=dict)   # tracer name -> 'standard' | 'croom'
    radiation: bool = False
    relativistic: bool = False
    asymmetry: bool = False
    uv_shotnoise: bool = False
    rescale_coords_systematics: bool = False
    fht_lowring: bool = True         # [model] fht_lowring (reference pktoxi.py:42,53)
    fht_extrap: bool = False         # [model] fht_extrap (reference pktoxi.py:41,141): power-law pads of the FFTLog input
    old_growth: bool = False         # [model] old_growth_func (reference correlation_func.py:75-80)


@dataclass
class Pipeline:
    """One P(k,mu) -> xi(r,mu) chain: a core correlation or one metal pair."""
    tracer1: Tracer
    tracer2: Tracer
    dataset: str                 # correlation-item name (keys 'par binsize <name>')
    pk: PkOptions
    xi: XiOptions
    metal_corr: bool
    r: np.ndarray
    mu: np.ndarray
    z: np.ndarray
    xi_growth: np.ndarray
    rel_z_evol: np.ndarray
    rel_z_evol_1: np.ndarray = None      # per-tracer (1 + z_tracer) / (1 + z_eff) with new-bias-evolution
    rel_z_evol_2: np.ndarray = None
    delta_rp_name: str = None

    @property
    def corr_name(self):
        return f'{self.tracer1.name}x{self.tracer2.name}'


@dataclass
class MetalPair:
    names: tuple                 # (name1, name2) canonical order
    pipeline: Pipeline
    matrix: object               # scipy.sparse matrix / ndarray, or None for identity
    double_count: bool           # xi *= 2 (reference vega/metals.py:238-239)
    cross_with_main: bool
    auto_bias_names: tuple = None  # candidate 'bias_<m1>_<m2>' names (separate-metal-auto-biases)
    kron: tuple = None           # (A [n_rp, n_rp], B [n_rt, n_rt] or None): matrix = kron(A, B), as new_metals builds it


@dataclass
class BroadbandTerm:
    name: str
    func: str                    # 'broadband' | 'broadband_sky'
    kind: str                    # 'add' | 'mul'
    pos: str                     # 'pre' | 'post'
    coords: str                  # 'r,mu' | 'rp,rt'
    r1: tuple
    r2: tuple


@dataclass
class CorrItem:
    name: str
    tracer1: Tracer
    tracer2: Tracer
    core: Pipeline
    model_grid: Grid
    dist_grid: Grid
    data_grid: Grid
    metals: list
    metal_opts: dict
    broadband: list
    distortion: object           # ndarray / scipy.sparse, or None (identity)
    data_vec: np.ndarray
    data_mask: np.ndarray
    model_mask: np.ndarray
    cov: np.ndarray              # full covariance or None (identity)
    rp_binsize: float = 4.0
    inst_sys_table: np.ndarray = None   # (rt, xi) table of the DESI instrumental-systematics model, or None
    cov_rescale: float = None           # [data] cov_rescale: also the default mock scale (reference analysis.py:147-150)
    cholesky_masked_cov: bool = True    # [data] cholesky-masked-cov (reference data.py:46, :727-757)

    _inv_masked_cov: np.ndarray = None
    _log_cov_det: float = None

    @property
    def masked_data_vec(self):
        return self.data_vec[self.data_mask]

    @property
    def data_size(self):
        return int(self.data_mask.sum())

    @property
    def inv_masked_cov(self):
        """Inverse of the masked covariance (reference vega/utils.py:271-298)."""
        if self._inv_masked_cov is None:
            n = self.data_size
            if self.cov is None:
                self._inv_masked_cov = np.eye(n)
            else:
                masked = self.cov[:, self.data_mask][self.data_mask, :]
                self._inv_masked_cov = np.linalg.inv(masked)
        return self._inv_masked_cov

    # marginalize-in-fit (reference vega_interface.py:282-292, :546-579): the best-fit template coefficients are a
    # linear map of the residual, coeff = M diff, and the templates T are added to the model before chi2, so
    # diff' = (I - T_masked M) diff =: P diff with a static projector P
    marg_templates = None        # distorted templates [n_dist, n_templates] (sparse)
    marg_diff2coeff = None       # M [n_templates, n_masked]
    marginalize_in_fit = False
    num_marg_modes = 0           # modes the marginalisation removes (reference data.py:92,825)
    variance = None              # diagonal of the covariance as read (reference data.py:82-85); None: taken from `cov`
    has_data = True              # False: a correlation without a data file - model only, on the caller's coordinates
    nb = None                    # pair counts of the data file (column NB), if any

    @property
    def effective_data_size(self):
        """Fitted bins minus the marginalised modes (reference vega/data.py:134)."""
        return self.data_size - self.num_marg_modes

    def marg_projector(self):
        tm = self.marg_templates[self.model_mask, :]
        tm = tm.toarray() if hasattr(tm, 'toarray') else np.asarray(tm)
        return np.eye(self.data_size) - tm.dot(self.marg_diff2coeff)

    @property
    def chi2_matrix(self):
        """The matrix Q of chi2 = diff^T Q diff: the inverse masked covariance, or P^T C^-1 P when the
        marginalisation templates are fitted on the fly."""
        if not (self.marginalize_in_fit and self.marg_diff2coeff is not None):
            return self.inv_masked_cov
        if getattr(self, '_chi2_matrix', None) is None:
            P = self.marg_projector()
            self._chi2_matrix = P.T.dot(self.inv_masked_cov).dot(P)
        return self._chi2_matrix

    @property
    def log_cov_det(self):
        if self._log_cov_det is None:
            if self.cov is None:
                self._log_cov_det = 0.0
            else:
                masked = self.cov[:, self.data_mask][self.data_mask, :]
                self._log_cov_det = float(np.linalg.slogdet(masked)[1])
        return self._log_cov_det

    def set_covariance(self, cov, inv_masked_cov=None):
        """`inv_masked_cov`: the inverse of the masked covariance when the caller holds it already (it is computed at first use
        otherwise - a second and a half at 3180 bins)."""
        self.cov = None if cov is None else np.asarray(cov, dtype=float)
        if inv_masked_cov is not None:
            inv_masked_cov = np.asarray(inv_masked_cov, dtype=float)
            if cov is None or inv_masked_cov.shape != (self.data_size, self.data_size):
                raise ValueError('inv_masked_cov must be the [data_size, data_size] inverse of the masked covariance')
        self._inv_masked_cov = inv_masked_cov
        self._log_cov_det = None
        self._chi2_matrix = None


@dataclass
class ScaleOptions:
    """[cosmo-fit type] (reference vega/scale_parameters.py:12-36)."""
    parametrisation: str = 'ap_at'
    full_shape: bool = False
    full_shape_alpha: bool = False
    smooth_scaling: bool = False
    metal_scaling: bool = False
    two_alpha_smooth: bool = False


@dataclass
class Problem:
    k: np.ndarray
    pk_full: np.ndarray
    pk_smooth: np.ndarray
    z_fid: float
    z_eff: float
    omega_m: float
    omega_de: float
    growth_rate: float
    scale: ScaleOptions
    params: dict
    sample_params: dict
    priors: dict
    items: dict
    global_cov: np.ndarray = None
    main_config: object = None
    mc_config: dict = None

    _global = None
    search_dirs = ()

    @property
    def pk_fid(self):
        """Fiducial P(k) scaled to z_eff for the Arinyo term (reference power_spectrum.py:72-73)."""
        return self.pk_full * ((1 + self.z_fid) / (1. + self.z_eff))**2

    def global_masks(self):
        """Concatenated masks and masked inverse of the global covariance
        (reference vega/vega_interface.py:907-954)."""
        if self.global_cov is None:
            return None
        if self._global is None:
            data_mask = np.concatenate([it.data_mask for it in self.items.values()])
            model_mask = np.concatenate([it.model_mask for it in self.items.values()])
            masked = self.global_cov[:, data_mask][data_mask, :]
            invcov = np.linalg.inv(masked)
            chi2_matrix = invcov
            if any(it.marginalize_in_fit and it.marg_diff2coeff is not None for it in self.items.values()):
                # block-diagonal projector of the items that fit their templates (reference :282-292: the
                # coefficients ignore the global covariance)
                P = np.eye(masked.shape[0])
                j = 0
                for it in self.items.values():
                    n = it.data_size
                    if it.marginalize_in_fit and it.marg_diff2coeff is not None:
                        P[j:j + n, j:j + n] = it.marg_projector()
                    j += n
                chi2_matrix = P.T.dot(invcov).dot(P)
            self._global = dict(data_mask=data_mask, model_mask=model_mask, invcov=invcov, chi2_matrix=chi2_matrix,
                                log_det=float(np.linalg.slogdet(masked)[1]))
        return self._global


# --------------------------------------------------------------------------------------
# parsing helpers
# --------------------------------------------------------------------------------------
def _parser(path):
    cfg = configparser.ConfigParser()
    cfg.optionxform = lambda option: option   # case-preserving, as the reference
    if not cfg.read(str(path)):
        raise RuntimeError(f'Could not read config {path}')
    return cfg


def load_fvoigt_table(path, search_dirs=()):
    """(k, F) table of the Voigt-profile HCD model: the reference's text file or its ``.npy`` twin."""
    try:
        return np.loadtxt(find_file(path, search_dirs))
    except RuntimeError:
        alt = str(path)[:-4] + '.npy' if str(path).endswith('.txt') else str(path) + '.npy'
        return np.load(find_file(alt, [*search_dirs, *[Path(d) / 'inputs' for d in search_dirs]]))


def _pk_options(section, bin_size_rp, bin_size_rt, search_dirs):
    opts = PkOptions(bin_size_rp=bin_size_rp, bin_size_rt=bin_size_rt)
    opts.use_gk = section.getboolean('model binning', True)
    opts.skip_nl_in_peak = section.getboolean('skip-nl-model-in-peak', False)
    opts.damping_scale = section.getfloat('pk-damping-scale', None)
    opts.damping_power = section.getint('pk-damping-power', 2)
    opts.uvb = section.getboolean('UVB-fluctuations', False)
    opts.heii = section.getboolean('HeII-reionization', False)
    opts.n_mu = section.getint('num_bins_muk', 1000)

    hcd = section.get('model-hcd', None)
    if hcd is not None:
        # substring precedence as in reference power_spectrum.py:291-303
        if 'Rogers' in hcd:
            opts.hcd_model = 'Rogers'
        elif 'fvoigt' in hcd:
            opts.hcd_model = 'fvoigt'
            model = section.get('fvoigt_model')
            if model is None:
                raise ValueError('No fvoigt_model specified in config')
            path = model if '/' in model else f'fvoigt_models/Fvoigt_{model}.txt'
            opts.fvoigt_table = load_fvoigt_table(path, search_dirs)
        elif 'sinc' in hcd:
            opts.hcd_model = 'sinc'
        else:
            raise ValueError(f"Unknown hcd model {hcd}. Choose from ['Rogers', 'fvoigt', 'sinc']")

    nl = section.get('small scale nl', None)
    if nl is not None:
        if 'arinyo' in nl:
            opts.small_scale_nl = 'arinyo'
        elif 'mcdonald' in nl:
            opts.small_scale_nl = 'mcdonald'
        else:
            raise ValueError("Incorrect 'small scale nl' specified")

    sm = section.get('fullshape smoothing', None)
    if sm is not None:
        if 'gauss' in sm:
            opts.fullshape_smoothing = 'gauss'
        elif 'exp' in sm:
            opts.fullshape_smoothing = 'exp'
        else:
            raise ValueError('"fullshape smoothing" must be of type "gauss" or "exp".')

    vd = section.get('velocity dispersion', None)
    if vd is not None:
        # 'gauss' is tested first, so 'lorentz_gauss' resolves to gauss
        # (reference power_spectrum.py:180-186)
        if 'gauss' in vd:
            opts.velocity_dispersion = 'gauss'
        elif 'lorentz' in vd:
            opts.velocity_dispersion = 'lorentz'
        else:
            raise ValueError('"velocity dispersion" must be of type "gauss" or "lorentz".')

    if 'mock-bin-size' in section:
        opts.mock_bin_size = section.getfloat('mock-bin-size')
        opts.mock_los_smoothing = section.get('mock-los-smoothing', None)
        if opts.mock_los_smoothing not in (None, 'growth', 'amplitude', 'only-los'):
            raise ValueError(f'Unknown mock LOS smoothing option {opts.mock_los_smoothing}.')
    return opts


def _xi_options(model_section, xi_section, tracers):
    """``model_section`` feeds PktoXi (always the item's [model]; reference metals.py:131-132),
    ``xi_section`` feeds CorrelationFunction ([model] for the core, [metals] for metal pairs)."""
    opts = XiOptions()
    opts.ell_max = model_section.getint('ell_max', 6)
    opts.old_fftlog = model_section.getboolean('old_fftlog', False)
    opts.fht_extrap = model_section.getboolean('fht_extrap', False)       # (the legacy transform ignores it)
    opts.fht_lowring = model_section.getboolean('fht_lowring', True)
    opts.single_multipole = xi_section.getint('single_multipole', -1)
    opts.rescale_coords_systematics = xi_section.getboolean('rescale-coords-systematics', False)
    for tr in tracers:
        handle = f'z evol {tr.name}'
        model = xi_section.get(handle, 'standard') if handle in xi_section \
            else xi_section.get('z evol', 'standard')
        opts.evol_model[tr.name] = 'croom' if 'croom' in model else 'standard'
    opts.radiation = xi_section.getboolean('radiation effects', False) \
        if 'radiation effects' in xi_section else False
    opts.relativistic = xi_section.getboolean('relativistic correction', False) \
        if 'relativistic correction' in xi_section else False
    opts.asymmetry = xi_section.getboolean('standard asymmetry', False) \
        if 'standard asymmetry' in xi_section else False
    opts.uv_shotnoise = xi_section.getboolean('UVB-shotnoise', False) if 'UVB-shotnoise' in xi_section else False
    opts.new_bias_evol = xi_section.getboolean('new-bias-evolution', False)
    opts.old_growth = xi_section.getboolean('old_growth_func', False)
    return opts


def picca_dist_hubble(z, cosmo):
    """D_H(z) = c / H(z) in Mpc/h of the picca cosmology a data file carries (OMEGAM, OMEGAK, OMEGAR, WL header
    keywords; reference vega/data.py:360-366, vega/correlation_item.py:138-151).  picca (picca.constants.Cosmo) is
    absent from the reference tree and from this image: this restates its published construction - H on a 10000-point
    grid up to z = 10, linearly interpolated - and is NOT pinned against picca itself."""
    om, ok, orad, wl = cosmo['Omega_m'], cosmo['Omega_k'], cosmo['Omega_r'], cosmo['wl']
    ol = 1. - ok - om - orad
    zg = np.arange(10000) * (10. / 10000)
    hubble = 100. * np.sqrt(ol * (1. + zg)**(3. * (1. + wl)) + ok * (1. + zg)**2 + om * (1. + zg)**3 + orad * (1. + zg)**4)
    return np.interp(z, zg, 299792.458 / hubble)


def _make_pipeline(tr1, tr2, dataset, pk_opts, xi_opts, grid, problem_consts, metal_corr, cosmo=None):
    z_fid, z_eff, om, ode = problem_consts
    if xi_opts.radiation:
        names = [tr1.name, tr2.name]
        if not ('QSO' in names and 'LYA' in names):
            raise ValueError('You asked for QSO radiation effects, but it can only be applied '
                             'to the cross (QSOxLya)')
    if xi_opts.relativistic or xi_opts.asymmetry:
        types = [tr1.type, tr2.type]
        if ('continuous' not in types) or (types[0] == types[1]):
            raise ValueError('You asked for relativistic effects or standard asymmetry, '
                             'but they only work for the cross')
    delta_rp_name = None
    if tr1.type == 'discrete' and tr2.type != 'discrete':
        delta_rp_name = 'drp_' + tr1.name
    elif tr2.type == 'discrete' and tr1.type != 'discrete':
        delta_rp_name = 'drp_' + tr2.name
    z = grid.z
    pipe = Pipeline(
        tracer1=tr1, tracer2=tr2, dataset=dataset, pk=pk_opts, xi=xi_opts,
        metal_corr=metal_corr, r=grid.r, mu=grid.mu, z=z,
        xi_growth=np.asarray((growth_squared_old if xi_opts.old_growth else growth_squared)(z, z_fid, om, ode), dtype=float),
        rel_z_evol=(1. + z) / (1 + z_eff), delta_rp_name=delta_rp_name)
    # new-bias-evolution (reference vega/correlation_func.py:238-274): in a cross-correlation the two tracers sit
    # at z -/+ rp / (2 D_H(z)); auto-correlations and files without a cosmology keep the mean redshift
    if xi_opts.new_bias_evol and tr1.type != tr2.type and cosmo is not None:
        if 'croom' in xi_opts.evol_model.values():
            raise ValueError('Croom model is not supported with new bias evol')
        shift = (grid.r * grid.mu) / (2 * picca_dist_hubble(z, cosmo))
        rel_q, rel_f = (1. + z - shift) / (1 + z_eff), (1. + z + shift) / (1 + z_eff)
        pipe.rel_z_evol_1 = rel_q if tr1.type == 'discrete' else rel_f
        pipe.rel_z_evol_2 = rel_q if tr2.type == 'discrete' else rel_f
    return pipe


def _use_metal_correlation(name1, name2, use_metal_autos):
    """reference vega/data.py:632-653"""
    if name1 == 'CIV(eff)' or name2 == 'CIV(eff)':
        return name1 == name2
    if 'SiII' in name1 and 'SiII' in name2 and not use_metal_autos:
        return False
    return True


def _canonical_pair(corr, tr1_name, tr2_name):
    """reference vega/correlation_item.py:91-100"""
    pair = tuple(sorted([corr[0], corr[1]]))
    if pair[0] == tr2_name or pair[1] == tr1_name:
        pair = (pair[1], pair[0])
    return pair


def _parse_broadband(section, item_name):
    """reference vega/broadband_poly.py:30-72"""
    terms = []
    for i, raw in enumerate(section.values()):
        bb = raw.split()
        if len(bb) not in (5, 6):
            raise ValueError(f'Broadband setup must have 5 or 6 elements. Got {len(bb)} elements')
        if bb[0] not in ('add', 'mul'):
            raise ValueError(f'Broadband type must be either "add" or "mul". Got {bb[0]}')
        if bb[1] not in ('pre', 'post'):
            raise ValueError(f'Broadband position must be either "pre" or "post". Got {bb[1]}')
        if bb[2] not in ('rp,rt', 'r,mu'):
            raise ValueError(f'Broadband coordinates must be either "rp,rt" or "r,mu". Got {bb[2]}')
        r1 = tuple(int(x) for x in bb[3].split(':'))
        r2 = tuple(int(x) for x in bb[4].split(':'))
        if len(r1) != 3 or len(r2) != 3:
            raise ValueError('Broadband coordinates must be in the format "min:max:step".')
        if len(bb) == 6:
            if bb[5] != 'broadband_sky':
                raise ValueError('The sixth broadband element must be "broadband_sky".')
            name, func = f'BB-{item_name}-{i}-{bb[5]}', 'broadband_sky'
        else:
            name, func = f'BB-{item_name}-{i} {bb[0]} {bb[1]} {bb[2]}', 'broadband'
        terms.append(BroadbandTerm(name, func, bb[0], bb[1], bb[2], r1, r2))
    return terms


# --------------------------------------------------------------------------------------
# small-scale marginalisation: a set-up time covariance update
# (reference vega/correlation_item.py:175-268, vega/data.py:96-109, :762-828)
# --------------------------------------------------------------------------------------
def marginalization_templates(model_grid, dist_grid, cuts, marg, match_data_bins=False):
    """Undistorted templates [n_model, n_templates]: one indicator per marginalised model bin (or per nearest
    distorted-grid bin with ``match_data_bins``)."""
    if 'all-rmin' not in marg:
        sets = []
        if 'rtmax' in marg:
            sets.append(np.nonzero(model_grid.rt_regular < marg['rtmax'])[0])
        if 'rtmin' in marg:
            sets.append(np.nonzero(model_grid.rt_regular > marg['rtmin'])[0])
        if 'rpmax' in marg:
            sets.append(np.nonzero(np.abs(model_grid.rp_regular) < marg['rpmax'])[0])
        if 'rpmin' in marg:
            sets.append(np.nonzero(np.abs(model_grid.rp_regular) > marg['rpmin'])[0])
        common = sets[0]
        for other in sets[1:]:
            common = np.intersect1d(common, other)
        if common.size == 0:
            raise ValueError('No common indices found for small-scale marginalization templates.')
    else:
        # every model bin under a distorted-grid bin that the small-scale cuts remove
        def get(key, default):
            return float(cuts.get(key, default)) if cuts is not None else default
        keep = (dist_grid.rp_regular > get('rp-min', 0.)) & (dist_grid.rt_regular > get('rt-min', 0.))
        keep &= dist_grid.r_regular > get('r-min', 10.)
        keep = keep.reshape(dist_grid.n_rp, dist_grid.n_rt)
        cb = model_grid.n_rp // dist_grid.n_rp
        mask_model = np.zeros((model_grid.n_rp, model_grid.n_rt))
        for i in range(dist_grid.n_rp):
            for j in range(dist_grid.n_rt):
                mask_model[i * cb:i * cb + cb, j * cb:j * cb + cb] = keep[i, j]
        common = np.nonzero(~mask_model.reshape(-1).astype(bool))[0]
    n = model_grid.rt_regular.size
    ones = np.ones(common.size)
    if match_data_bins:
        rp, rt = model_grid.rp[common], model_grid.rt[common]
        nearest = ((dist_grid.rp[None, :] - rp[:, None])**2 + (dist_grid.rt[None, :] - rt[:, None])**2).argmin(axis=1)
        unique = np.unique(nearest)
        rows = np.searchsorted(unique, nearest)
        return sparse.coo_array((ones, (rows, common)), shape=(unique.size, n)).tocsr().T
    return sparse.coo_array((ones, (np.arange(common.size), common)), shape=(common.size, n)).tocsr().T


def marginalization_scale_mask(grid, cuts, marg):
    """Bins that are marginalised (reference vega/coordinates.py:184-217)."""
    mask = np.ones_like(grid.rp_regular, dtype=bool)
    if 'rtmax' in marg:
        mask &= grid.rt_regular < marg['rtmax']
    if 'rtmin' in marg:
        mask &= grid.rt_regular > marg['rtmin']
    if 'rpmax' in marg:
        mask &= np.abs(grid.rp_regular) < marg['rpmax']
    if 'rpmin' in marg:
        mask &= np.abs(grid.rp_regular) > marg['rpmin']
    if 'all-rmin' in marg:
        def get(key, default):
            return float(cuts.get(key, default)) if cuts is not None else default
        keep = (grid.rp_regular > get('rp-min', 0.)) & (grid.rt_regular > get('rt-min', 0.))
        keep &= grid.r_regular > get('r-min', 10.)
        mask = ~keep
    return mask


def marginalization_modes(distortion, model_grid, dist_grid, cuts, marg, model_mask, prior_sigma=10.0,
                          match_data_bins=False, factor=1e-8):
    """(A A^T, number of modes) of the distorted, masked, prior-scaled templates with degenerate modes removed by an SVD
    (reference vega/data.py:762-828): the matrix added to the masked block of the covariance, and the count the reference
    keeps as ``Data.num_marg_modes`` (:825) for the effective data size."""
    templates = distortion.dot(marginalization_templates(model_grid, dist_grid, cuts, marg, match_data_bins))
    t = (templates * prior_sigma)[model_mask, :].toarray()
    u, sv, _ = np.linalg.svd(t, full_matrices=False)
    w = sv > factor * sv[0]
    u, sv = u[:, w], sv[w]
    return np.dot(u * sv**2, u.T), int(w.sum())


def marginalization_cov_update(*args, **kw):
    """The covariance update alone (see :func:`marginalization_modes`)."""
    return marginalization_modes(*args, **kw)[0]


# --------------------------------------------------------------------------------------
# one correlation item
# --------------------------------------------------------------------------------------
def _build_item(cfg, consts, search_dirs, marginalize_in_fit=False, coordinates=None):
    d = cfg['data']
    name = d.get('name')
    tr1 = Tracer(d.get('tracer1'), d.get('tracer1-type'))
    tr2 = Tracer(d.get('tracer2', tr1.name), d.get('tracer2-type', tr1.type))
    model_sec = cfg['model']

    # small-scale marginalisation (reference vega/correlation_item.py:53-72)
    marg = {}
    for key, short in (('marginalize-below-rtmax', 'rtmax'), ('marginalize-above-rtmin', 'rtmin'),
                       ('marginalize-below-rpmax', 'rpmax'), ('marginalize-above-rpmin', 'rpmin')):
        if model_sec.getfloat(key, 0) > 0:
            marg[short] = model_sec.getfloat(key, 0)
    if model_sec.getboolean('marginalize-all-rmin-cuts', False):
        marg['all-rmin'] = True
    fit_marg_scales = bool(marg) and model_sec.getboolean('fit-marginalized-scales', False)
    new_metals = model_sec.getboolean('new_metals', False)
    has_data = 'filename' in d and d.getboolean('has_datafile', True)
    if not has_data:
        # A correlation without a data file (reference vega/correlation_item.py:40-42): the model alone, on coordinates the caller
        # hands in (reference: `corr_item.init_coordinates(Coordinates(...))` before `compute_model`, correlation_item.py:120-136,
        # vega_interface.py:110-137 - `data[name] = None`, no distortion matrix, model on the model coordinates, chi2 refused).
        # Here the coordinates come with the constructor: VegaInterface(main, coordinates={name: Grid(...)}).
        if coordinates is None:
            raise NotImplementedError(f'{name}: a correlation without a data file computes its model on coordinates the caller '
                                      'hands in - VegaInterface(main, coordinates={name: vega_amd.Coordinates(rp_min, rp_max, rt_max, '
                                      'rp_nbins, rt_nbins)})')
        if 'metals' in cfg and not new_metals:
            raise NotImplementedError(f'{name}: metal matrices come with a data file; a model-only correlation has no metal terms')
        g = coordinates
        z = np.full(g.rp.size, consts[1]) if g.z is None else np.broadcast_to(np.asarray(g.z, dtype=float), g.rp.shape)
        from .tables import Table
        tabs = [Table({'RPMIN': g.rp_min, 'RPMAX': g.rp_max, 'RTMAX': g.rt_max, 'NP': g.n_rp, 'NT': g.n_rt, 'BLINDING': 'none'},
                      {'DA': np.zeros(g.rp.size), 'RP': g.rp, 'RT': g.rt, 'Z': z})]
    else:
        # ---- data file (reference vega/data.py:285-421)
        tabs = read_tables(find_file(d.get('filename'), search_dirs))
    t1 = tabs[0]
    hdr = t1.header
    # blinding strategy of the file (reference vega/data.py:305-339): `desi_dr3` data are blinded (DA_BLIND must be
    # there), the earlier DESI tags and none / None are not, anything else is an error
    blinding = hdr.get('BLINDING', None)
    if blinding in ('none', 'None'):
        blinding = None
    blind = False
    if blinding in BLINDING_STRATEGIES:
        blind = True
        if blinding == 'desi_dr3' and 'DA_BLIND' not in t1.data:
            raise AssertionError('Blinding failed, do not run!!!')
        if 'DA_BLIND' in t1.data:
            data_vec = np.asarray(t1.data['DA_BLIND'], dtype=float)
        elif 'DA' in t1.data:
            data_vec = np.asarray(t1.data['DA'], dtype=float)
        else:
            raise ValueError('No DA or DA_BLIND column found in data file.')
    elif blinding is None or blinding in ('desi_m2', 'desi_y1', 'desi_y3'):
        data_vec = np.asarray(t1.data['DA'], dtype=float)
    else:
        raise ValueError(f'Unknown blinding strategy {blinding}.')
    data_grid = Grid(hdr['RPMIN'], hdr['RPMAX'], hdr['RTMAX'], hdr['NP'], hdr['NT'],
                     rp=t1.data['RP'], rt=t1.data['RT'], z=t1.data['Z'])

    distortion = None
    cov = None
    model_grid = None
    dist_grid = None
    dmat_path = d.get('distortion-file', None)
    cov_path = d.get('covariance-file', None)
    has_distortion_flag = d.getboolean('distortion', True)
    if dmat_path is None:
        for col in ('DM_BLIND', 'DM'):
            if t1.has(col):
                distortion = sparse.csr_array(np.asarray(t1.data[col], dtype=float))
                break
        if len(tabs) > 1 and tabs[1].has('DMRP'):
            t2 = tabs[1]
            model_grid = Grid(hdr['RPMIN'], hdr['RPMAX'], hdr['RTMAX'], hdr['NP'], hdr['NT'],
                              rp=t2.data['DMRP'], rt=t2.data['DMRT'], z=t2.data['DMZ'])
    else:
        dt = read_tables(find_file(dmat_path, search_dirs))
        dh = dt[0].header
        col = 'DM' if dt[0].has('DM') else 'DM_BLIND'
        distortion = sparse.csr_array(np.asarray(dt[0].data[col], dtype=float))
        coef = int(dh['COEFMOD'])
        model_grid = Grid(dh['RPMIN'], dh['RPMAX'], dh['RTMAX'], dh['NP'] * coef, dh['NT'] * coef,
                          rp=dt[1].data['RP'], rt=dt[1].data['RT'], z=dt[1].data['Z'])
        dist_grid = Grid(dh['RPMIN'], dh['RPMAX'], dh['RTMAX'], dh['NP'], dh['NT'])

    if cov_path is not None:
        cov = np.asarray(read_tables(find_file(cov_path, search_dirs))[0].data['CO'], dtype=float)
    elif t1.has('CO'):
        cov = np.asarray(t1.data['CO'], dtype=float)
    rescale = d.getfloat('cov_rescale', None)
    if cov is not None and rescale is not None:
        cov = cov * rescale
    # what the fit output writes next to the data (reference vega/data.py:82-85, :370): the covariance's diagonal as read,
    # the pair counts if the file has them
    variance = np.ones(data_vec.size) if cov is None else np.array(np.diag(cov), dtype=float)
    nb = np.asarray(t1.data['NB']) if t1.has('NB') else None
    cosmo = None
    if 'OMEGAM' in hdr:     # the picca cosmology of the file: only new-bias-evolution (and new_metals) use it
        cosmo = {'Omega_m': float(hdr['OMEGAM']), 'Omega_k': float(hdr.get('OMEGAK', 0.)),
                 'Omega_r': float(hdr.get('OMEGAR', 0.)), 'wl': float(hdr.get('WL', -1.))}

    if model_grid is None:
        model_grid = data_grid
    if dist_grid is None:
        dist_grid = model_grid
    if not has_distortion_flag:
        distortion = None

    cuts = cfg['cuts'] if 'cuts' in cfg else None
    data_mask = data_grid.scale_cut_mask(cuts)
    model_mask = dist_grid.scale_cut_mask(cuts)

    if marg:
        if distortion is None:
            raise ValueError('Distortion matrix required for marginalization')
        if fit_marg_scales:
            # the marginalised scales join the fitted bins (reference vega/data.py:793-812)
            data_mask = data_mask | marginalization_scale_mask(data_grid, cuts, marg)
            model_mask = model_mask | marginalization_scale_mask(dist_grid, cuts, marg)
            if data_mask.sum() != model_mask.sum():
                raise ValueError('Data and model masks should be the same after marginalization scale cuts.')
        if cov is None:
            cov = np.eye(data_vec.size)
        cov = np.array(cov, dtype=float)
        prior_sigma = model_sec.getfloat('marginalize-prior-sigma', 10.0)
        match_bins = model_sec.getboolean('marginalize-match-data-bins', False)
        marg_templates = distortion.dot(marginalization_templates(model_grid, dist_grid, cuts, marg, match_bins))
        # coefficient solve of the templates against the covariance before any update (reference data.py:101-128)
        inv = np.linalg.inv(cov[:, data_mask][data_mask, :])
        tm = marg_templates[model_mask, :]
        G = tm.T.dot(inv)
        A = tm.T.dot(G.T).T
        if not (fit_marg_scales and match_bins):
            A = A + np.diag(np.full(marg_templates.shape[1], prior_sigma**-2))
        marg_diff2coeff = np.linalg.inv(A).dot(G)
        update, num_marg_modes = marginalization_modes(distortion, model_grid, dist_grid, cuts, marg, model_mask,
                                                       prior_sigma=prior_sigma, match_data_bins=match_bins)
        if not marginalize_in_fit:
            cov[np.ix_(data_mask, data_mask)] += update
            # (the reference's `variance` is a live view of the covariance's diagonal, vega/data.py:85: the in-place update of
            # data.py:107 shows in what the result file writes as <name>_VAR)
            variance = np.array(np.diag(cov), dtype=float)

    # the reference injects the data bin sizes into the [model] / [metals] sections
    # (reference vega/model.py:38-39, vega/metals.py:119-123)
    bs_rp, bs_rt = data_grid.rp_binsize, data_grid.rt_binsize
    core = _make_pipeline(tr1, tr2, name, _pk_options(model_sec, bs_rp, bs_rt, search_dirs),
                          _xi_options(model_sec, model_sec, (tr1, tr2)), model_grid, consts, False, cosmo=cosmo)

    # ---- metals (reference vega/data.py:475-687, vega/metals.py:43-142)
    metals = []
    metal_opts = {}
    if 'metals' in cfg:
        msec = cfg['metals']
        use_autos = model_sec.getboolean('use_metal_autos', True)
        test_flag = d.getboolean('test', False)
        in1 = msec.get('in tracer1').split() if 'in tracer1' in msec else None
        in2 = msec.get('in tracer2').split() if 'in tracer2' in msec else None
        if in1 is None and in2 is None:
            raise ValueError("The metals config must specify 'in tracer1' and/or 'in tracer2'")
        catalog = {tr1.name: tr1, tr2.name: tr2}
        for m in (in1 or []) + (in2 or []):
            catalog[m] = Tracer(m, 'continuous')

        mtabs = mh = prefix = None
        if not new_metals:
            mtabs = read_tables(find_file(msec.get('filename'), search_dirs))
            mh = mtabs[0].header
            prefix = 'DM_BLIND_' if mh.get('BLINDING', 'none') != 'none' else 'DM_'
        pairs = []          # (tracers tuple as listed, column-name stem)

        def _stem(a, b):
            if new_metals:
                return None
            stem = f'{a}_{b}'
            return stem if mtabs[1].has('RP_' + stem) else f'{b}_{a}'

        if in2 is not None:
            for m in in2:
                if _use_metal_correlation(tr1.name, m, use_autos):
                    pairs.append(((tr1.name, m), _stem(tr1.name, m)))
        if in1 is not None:
            for m in in1:
                if _use_metal_correlation(m, tr2.name, use_autos):
                    pairs.append(((m, tr2.name), _stem(m, tr2.name)))
        if in1 is not None and in2 is not None:
            for i, m1 in enumerate(in1):
                j0 = i if (tr1.name == tr2.name and tr1.type == tr2.type) else 0
                for m2 in in2[j0:]:
                    if _use_metal_correlation(m1, m2, use_autos):
                        pairs.append(((m1, m2), _stem(m1, m2)))

        metal_opts = dict(
            separate_metal_auto_biases=model_sec.getboolean('separate-metal-auto-biases', False),
            single_metal_beta=model_sec.getboolean('single-metal-beta', False),
            fast_metals=model_sec.getboolean('fast_metals', False),
            fast_metal_bias=model_sec.getboolean('fast_metal_bias', True),
            no_metal_decomp=model_sec.getboolean('no-metal-decomp', True),
        )
        if metal_opts['fast_metals'] or metal_opts['separate_metal_auto_biases']:
            metal_opts['fast_metal_bias'] = True
        rp_only = model_sec.getboolean('rp_only_metal_mats', False)
        if rp_only and not new_metals:
            raise ValueError('rp_only_metal_mats needs new_metals = True (matrices read from a file are full ones)')

        stored = {}
        builder = None
        if new_metals:
            # matrices built here from the stacked-delta weights (reference vega/metals.py:83-112, :389-752)
            from .metal_matrices import MetalMatrixBuilder, PiccaCosmo
            if cosmo is None:
                raise ValueError('new_metals needs the cosmology keywords (OMEGAM, ...) in the data file header')
            if 'metal-matrix' not in cfg:
                raise ValueError('new_metals needs a [metal-matrix] section')
            w1 = d.get('weights-tracer1', None)
            if w1 is None:
                raise ValueError('new_metals needs [data] weights-tracer1')
            w2 = d.get('weights-tracer2', None) or w1
            builder = MetalMatrixBuilder(
                ((tr1.name, tr1.type), (tr2.name, tr2.type)),
                (find_file(w1, search_dirs), find_file(w2, search_dirs)), model_grid, dict(cfg['metal-matrix']),
                PiccaCosmo(Om=cosmo['Omega_m'], Ok=cosmo['Omega_k'], Or=cosmo['Omega_r'], wl=cosmo['wl']),
                zmin=d.getfloat('zmin', 0.0), zmax=d.getfloat('zmax', 10.0))
        for tracers, stem in pairs:
            if new_metals:
                continue
            grid = Grid(mh['RPMIN'], mh['RPMAX'], mh['RTMAX'], mh['NP'], mh['NT'],
                        rp=mtabs[1].data['RP_' + stem], rt=mtabs[1].data['RT_' + stem],
                        z=mtabs[1].data['Z_' + stem])
            col = prefix + stem
            if mtabs[1].has(col):
                mat = sparse.csr_array(np.asarray(mtabs[1].data[col], dtype=float))
            elif len(mtabs) > 2 and mtabs[2].has(col):
                mat = sparse.csr_array(np.asarray(mtabs[2].data[col], dtype=float))
            elif test_flag:
                mat = None    # identity (reference vega/data.py:683-684)
            else:
                raise ValueError('Cannot find correct metal matrices. Check that blinding is '
                                 'consistent between cf and metal files.')
            stored[tracers] = (grid, mat)

        is_auto = tr1.name == tr2.name
        seen = []
        metal_pk = _pk_options(msec, bs_rp, bs_rt, search_dirs)
        for tracers, _ in pairs:
            pair = _canonical_pair(tracers, tr1.name, tr2.name)
            if pair in seen:
                continue
            seen.append(pair)
            kron = None
            if builder is not None:
                # the canonical order puts each absorber on the side of the tracer whose forest holds it
                if rp_only:
                    m_rp, rp_eff, rt_eff, z_eff = builder.rp_matrix(*pair)
                    mat = sparse.csr_array(builder.expand_rp_matrix(m_rp))
                    kron = (m_rp, None)
                else:
                    mat, rp_eff, rt_eff, z_eff = builder.rp_rt_matrix(*pair)
                    mat = sparse.csr_array(mat)
                    kron = builder.last_factors
                grid = Grid(model_grid.rp_min, model_grid.rp_max, model_grid.rt_max, model_grid.n_rp, model_grid.n_rt,
                            rp=rp_eff, rt=rt_eff, z=z_eff)
            else:
                grid, mat = stored[pair] if pair in stored else stored[pair[::-1]]
            t_a, t_b = catalog[pair[0]], catalog[pair[1]]
            pipe = _make_pipeline(t_a, t_b, name, metal_pk,
                                  _xi_options(model_sec, msec, (t_a, t_b)), grid, consts, True, cosmo=cosmo)
            main = (tr1.name, tr2.name)
            metals.append(MetalPair(
                names=pair, pipeline=pipe, matrix=mat, kron=kron,
                double_count=is_auto and pair[0] != pair[1],
                cross_with_main=(pair[0] in main or pair[1] in main),
                auto_bias_names=(f'bias_{pair[0]}_{pair[1]}', f'bias_{pair[1]}_{pair[0]}')))

    broadband = _parse_broadband(cfg['broadband'], name) if 'broadband' in cfg else []

    inst_sys_table = None
    if model_sec.getboolean('desi-instrumental-systematics', False):
        # reference correlation_func.py:569-591: auto-correlations only; (RT, XI) table, linear interpolation
        if tr1.type != tr2.type:
            raise ValueError('DESI instrumental systematics model only applies to auto-correlation functions.')
        path = find_file('instrumental_systematics/desi-instrument-syst-for-forest-auto-correlation.csv',
                         [*search_dirs, *[Path(d) / 'inputs' for d in search_dirs]])
        inst_sys_table = np.loadtxt(path, delimiter=',', skiprows=1)

    item = CorrItem(name=name, tracer1=tr1, tracer2=tr2, core=core, model_grid=model_grid,
                    dist_grid=dist_grid, data_grid=data_grid, metals=metals, metal_opts=metal_opts,
                    broadband=broadband, distortion=distortion, data_vec=data_vec,
                    data_mask=data_mask, model_mask=model_mask, cov=cov,
                    rp_binsize=data_grid.rp_binsize, inst_sys_table=inst_sys_table)
    # also added to a global covariance (build_problem)
    item.cov_marg_update = update if (marg and not marginalize_in_fit) else None
    if marg:
        item.marg_templates, item.marg_diff2coeff = marg_templates, marg_diff2coeff
        item.marginalize_in_fit = bool(marginalize_in_fit)
        item.num_marg_modes = num_marg_modes
    item.blind, item.blinding_strat = blind, (blinding if blind else None)
    item.variance, item.nb = variance, nb
    item.cov_rescale = rescale
    item.cholesky_masked_cov = d.getboolean('cholesky-masked-cov', True)
    item.has_data = has_data
    return item


# --------------------------------------------------------------------------------------
# default sampling table (reference vega/parameters/default_values.txt is data the engine
# does not ship; callers that rely on "param = True" must give explicit limits)
# --------------------------------------------------------------------------------------
def _read_sample(section, params):
    """reference vega/vega_interface.py:738-816"""
    from .defaults import DEFAULT_VALUES
    out = {'limits': {}, 'values': {}, 'errors': {}, 'fix': {}}

    def default(param):
        if param not in DEFAULT_VALUES:
            raise ValueError(f'Default values not found for: {param}. Please provide the full sampling '
                             'specification.')
        return DEFAULT_VALUES[param]

    for param, values in section.items():
        if param not in params:
            print(f'Warning: You tried sampling the parameter: {param}. As this parameter was not specified '
                  'under [parameters], it will be skipped.')
            continue
        vals = values.split()
        if len(vals) > 1:
            lo = None if vals[0] == 'None' else float(vals[0])
            hi = None if vals[1] == 'None' else float(vals[1])
            out['limits'][param] = (lo, hi)
        else:
            if vals[0] not in TRUE_WORDS:
                continue
            out['limits'][param] = default(param)[0]
        out['values'][param] = float(vals[2]) if len(vals) > 2 else params[param]
        if len(vals) > 3:
            assert len(vals) == 4
            out['errors'][param] = float(vals[3])
        else:
            out['errors'][param] = default(param)[1]
        out['fix'][param] = False
    return out


def build_problem(main_path, search_dirs=(), fiducial_overrides=None, coordinates=None):
    """Parse ``main.ini`` and everything it names into a :class:`Problem`.

    ``fiducial_overrides`` may replace ``Omega_m`` / ``Omega_de`` read from the template
    (the reference's tests do this by mutating ``vega.fiducial`` before ``compute_model``;
    reference tests/test_vega.py:30,35).
    """
    main_path = Path(find_file(main_path, search_dirs))
    dirs = [Path(p) for p in search_dirs] + [main_path.parent, main_path.parent.parent]
    main = _parser(main_path)

    ftabs = read_tables(find_file(os.path.expandvars(main['fiducial'].get('filename')), dirs))
    fh = ftabs[0].header
    k = np.asarray(ftabs[0].data['K'], dtype=float)
    pk_full = np.asarray(ftabs[0].data['PK'], dtype=float)
    pk_smooth = np.asarray(ftabs[0].data['PKSB'], dtype=float)
    z_fid, om, ode = float(fh['ZREF']), float(fh['OM']), float(fh['OL'])
    if fiducial_overrides:
        fiducial_overrides = dict(fiducial_overrides)
        om = fiducial_overrides.get('Omega_m', om)
        ode = fiducial_overrides.get('Omega_de', ode)
    z_eff = main['data sets'].getfloat('zeff')
    consts = (z_fid, z_eff, om, ode)

    control = main['control'] if 'control' in main else {}
    marginalize_in_fit = False
    if 'control' in main:
        marginalize_in_fit = main['control'].getboolean('marginalize-in-fit', False)

    items = {}
    cfgs = {}
    for path in main['data sets'].get('ini files').split():
        cfg = _parser(find_file(os.path.expandvars(path), dirs))
        cfgs[cfg['data'].get('name')] = cfg
    for name, cfg in cfgs.items():
        items[name] = _build_item(cfg, consts, dirs, marginalize_in_fit, coordinates=(coordinates or {}).get(name))

    # parameters: component configs first, main config wins (reference :705-736)
    params = {}
    for cfg in cfgs.values():
        if 'parameters' in cfg:
            for p, v in cfg.items('parameters'):
                params[p] = float(v)
    for p, v in main['parameters'].items():
        params[p] = float(v)

    growth_rate = None
    use_template = control.getboolean('use_template_growth_rate', True) \
        if 'control' in main else True
    if 'F_ZREF' in fh:
        growth_rate = float(fh['F_ZREF'])
        if use_template:
            params['growth_rate'] = growth_rate
    elif 'growth_rate' in params:
        growth_rate = params['growth_rate']

    sample = _read_sample(main['sample'], params) if 'sample' in main else \
        {'limits': {}, 'values': {}, 'errors': {}, 'fix': {}}

    priors = {}
    if 'priors' in main:
        for p, spec in main['priors'].items():
            parts = spec.split()
            if len(parts) != 3:
                raise ValueError('Prior configuration must have the format: '
                                 '"<param> = gaussian <mean> <sigma>"')
            if parts[0] not in ('gaussian', 'Gaussian'):
                raise ValueError('Only gaussian priors are supported.')
            priors[p] = np.array(parts[1:], dtype=float)
            if p not in sample['limits']:
                raise ValueError(f'Prior specified for a parameter that is not sampled: {p}')

    sc = main['cosmo-fit type']
    scale = ScaleOptions(
        parametrisation=sc.get('cosmo fit func', 'ap_at'),
        full_shape=sc.getboolean('full-shape', False),
        full_shape_alpha=sc.getboolean('full-shape-alpha', False),
        smooth_scaling=sc.getboolean('smooth-scaling', False),
        metal_scaling=sc.getboolean('metal-scaling', False),
        two_alpha_smooth=sc.getboolean('two-alpha-smooth', False))
    if scale.parametrisation not in ('ap_at', 'aiso_epsilon', 'phi_alpha'):
        raise ValueError(f'Unknown parametrisation {scale.parametrisation}.')
    if scale.full_shape_alpha and scale.two_alpha_smooth:
        raise ValueError('The "full-shape-alpha" and "two-alpha-smooth" options are incompatible.')
    if scale.metal_scaling and scale.two_alpha_smooth:
        raise ValueError('The "metal-scaling" and "two-alpha-smooth" options are incompatible.')

    mc_config = None
    if 'monte carlo' in main:
        mc_config = {'params': {p: float(v) for p, v in main['mc parameters'].items()} if 'mc parameters' in main else {},
                     'sample': _read_sample(main['monte carlo'], params)}

    global_cov = None
    gc_file = main['data sets'].get('global-cov-file', None)
    if gc_file is not None:
        global_cov = np.asarray(read_tables(find_file(gc_file, dirs))[0].data['COV'], dtype=float)
        cov_scale = control.getfloat('cov_scale', None) if 'control' in main else None
        if cov_scale is not None:
            global_cov = global_cov * cov_scale
        # marginalisation templates update the items' diagonal blocks (reference vega_interface.py:918-938)
        j = 0
        for item in items.values():
            n = item.data_vec.size
            if getattr(item, 'cov_marg_update', None) is not None:
                block = global_cov[j:j + n, j:j + n]
                block[np.ix_(item.data_mask, item.data_mask)] += item.cov_marg_update
            j += n

    prob = Problem(k=k, pk_full=pk_full, pk_smooth=pk_smooth, z_fid=z_fid, z_eff=z_eff,
                   omega_m=om, omega_de=ode, growth_rate=growth_rate, scale=scale,
                   params=params, sample_params=sample, priors=priors, items=items,
                   global_cov=global_cov, main_config=main, mc_config=mc_config)
    prob.search_dirs = dirs
    return prob
