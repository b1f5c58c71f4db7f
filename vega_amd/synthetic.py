"""Deterministic synthetic tensors for the BASELINE configs.

The reference ships no distortion matrix, covariance or metal matrices with its test data
(the reference substitutes ``np.eye``: vega/data.py:77-80).  The
benchmark configs therefore use the seeded generators below; the same arrays are injected into
the reference when the golden fixtures are made (tests/golden/make_golden.py) so that engine,
oracle and reference all see identical inputs.
"""
import hashlib

import numpy as np

SEED = 20260803

# The dense generators are pure functions of their grids and cost seconds at 5000 bins; bench.py and the test suites build the same
# problems dozens of times per process.  Results are kept per (function, grid, arguments) and handed out READ-ONLY (a caller that
# wants to change one copies it); a few hundred MB per process at the benchmark's sizes.
_KEPT = {}


def _kept(tag, arrays, args, make):
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a, dtype=float)
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    key = (tag, h.hexdigest(), args)
    if key not in _KEPT:
        out = make()
        out.flags.writeable = False
        _KEPT[key] = out
    return _KEPT[key]


def forget():
    """Drops the kept arrays (a process that has built many different grids and wants the memory back)."""
    _KEPT.clear()


def distortion_matrix(rp, rt, seed=SEED, dense_fraction=0.65):
    """``DM = I - W``: W couples bins at similar rt, falls off with |d rp|; rows of W sum to 0.3;
    35 % of the far-off-band (|d rt| > 24) entries are zeroed so a CSR twin is meaningfully
    sparse.  (Kept per grid; the array is read-only.)"""
    return _kept('dm', (rp, rt), (seed, dense_fraction), lambda: _distortion_matrix(rp, rt, seed, dense_fraction))


def _distortion_matrix(rp, rt, seed, dense_fraction):
    rp = np.asarray(rp, dtype=float)
    rt = np.asarray(rt, dtype=float)
    n = rp.size
    rng = np.random.default_rng(seed + n)
    w_col = rng.uniform(0.5, 1.5, size=n)
    d_rt = rt[:, None] - rt[None, :]
    W = np.exp(-d_rt**2 / (2 * 8.**2))
    off_band = np.abs(d_rt) > 24
    del d_rt
    W /= 1 + np.abs(rp[:, None] - rp[None, :]) / 20.
    W *= w_col[None, :]
    drop = off_band & (rng.random((n, n)) > dense_fraction)
    W[drop] = 0.
    del drop, off_band
    W *= 0.3 / W.sum(axis=1, keepdims=True)
    dm = -W
    dm[np.arange(n), np.arange(n)] += 1.
    return dm


def fine_model_grid(rp_min, rp_max, rt_max, n_p, n_t, coef, z_ref=2.3):
    """Bin centres of a model grid `coef` times finer than the data grid (what a DESI distortion file's HDU 2 carries:
    reference vega/data.py:463-468), rp-major like the reference's regular grids, with a slowly varying redshift."""
    n_p, n_t = int(n_p) * int(coef), int(n_t) * int(coef)
    dp, dt = (rp_max - rp_min) / n_p, rt_max / n_t
    rp = np.repeat(rp_min + dp * (np.arange(n_p) + 0.5), n_t)
    rt = np.tile(dt * (np.arange(n_t) + 0.5), n_p)
    # (bin centres jittered like measured pair-weighted coordinates are: the grids of a real file are not the regular ones)
    rng = np.random.default_rng(SEED + n_p * n_t)
    rp = rp + 0.1 * dp * rng.uniform(-1, 1, rp.size)
    rt = rt + 0.1 * dt * rng.uniform(-1, 1, rt.size)
    z = z_ref + 4e-4 * (np.abs(rp) - 100.) - 2e-4 * (rt - 100.)
    return rp, rt, z


def distortion_matrix_rect(rp_data, rt_data, rp_model, rt_model, coef, seed=SEED, dense_fraction=0.65):
    """Rectangular distortion matrix [n_data][n_model] for a model grid `coef` times finer than the data grid (the
    `distortion-file` + COEFMOD case, reference vega/data.py:441-473): ``DM = A - W`` with A the average over the coef^2
    model bins inside each data bin (both grids rp-major) and W as in :func:`distortion_matrix` between data-bin and
    model-bin centres (rows of W sum to 0.3, 35 % of the far-off-band entries zeroed)."""
    rp_d, rt_d = np.asarray(rp_data, dtype=float), np.asarray(rt_data, dtype=float)
    rp_m, rt_m = np.asarray(rp_model, dtype=float), np.asarray(rt_model, dtype=float)
    nd, nm = rp_d.size, rp_m.size
    rng = np.random.default_rng(seed + nd + nm)
    w_col = rng.uniform(0.5, 1.5, size=nm)
    d_rt = rt_d[:, None] - rt_m[None, :]
    W = np.exp(-d_rt**2 / (2 * 8.**2))
    off_band = np.abs(d_rt) > 24
    del d_rt
    W /= 1 + np.abs(rp_d[:, None] - rp_m[None, :]) / 20.
    W *= w_col[None, :]
    W[off_band & (rng.random((nd, nm)) > dense_fraction)] = 0.
    del off_band
    W *= 0.3 / W.sum(axis=1, keepdims=True)
    dm = -W
    return dm


def add_bin_average(dm, n_p, n_t, coef):
    """dm[n_p n_t][coef n_p coef n_t] += the average of the coef^2 model bins inside each data bin."""
    n_tm = n_t * coef
    rows = np.arange(n_p * n_t)
    ip, it = rows // n_t, rows % n_t
    for a in range(coef):
        for b in range(coef):
            dm[rows, (coef * ip + a) * n_tm + coef * it + b] += 1. / coef**2
    return dm


def write_distortion_file(path, grid_header, coef, dm, rp, rt, z, blinding=None):
    """A separate distortion-matrix file as the reference reads it (vega/data.py:441-473): HDU 1 = the vector column DM
    [n_data rows][n_model] with RPMIN / RPMAX / RTMAX / NP / NT of the DATA grid and COEFMOD, HDU 2 = RP, RT, Z of the
    model grid (NP COEFMOD x NT COEFMOD bins)."""
    from . import fitslite
    dm = np.asarray(dm, dtype=float)
    hdr = {k: grid_header[k] for k in ('RPMIN', 'RPMAX', 'RTMAX', 'NP', 'NT')}
    hdr['COEFMOD'] = int(coef)
    if blinding is not None:
        hdr['BLINDING'] = blinding
    fitslite.write_tables(str(path), [
        ('DMAT', [('DM', f'{dm.shape[1]}D', dm)], hdr),
        ('ATTRI', [('RP', 'D', rp), ('RT', 'D', rt), ('Z', 'D', z)])], overwrite=True)
    return path


def write_covariance_file(path, cov):
    """A separate `covariance-file`: HDU 1 with the vector column CO (reference vega/data.py:349-352)."""
    from . import fitslite
    cov = np.asarray(cov, dtype=float)
    fitslite.write_tables(str(path), [('COV', [('CO', f'{cov.shape[1]}D', cov)])], overwrite=True)
    return path


def write_dmat_file_case(directory, source, coef=2):
    """The three files of the `distortion-file` case for one correlation: returns (distortion path, covariance path) for
    the data grid of ``source`` (a reference-format table list); the data vector stays in the item's own data file."""
    from pathlib import Path
    t1 = source[0]
    h = t1.header
    rp_d, rt_d = np.asarray(t1.data['RP'], dtype=float), np.asarray(t1.data['RT'], dtype=float)
    rp, rt, z = fine_model_grid(h['RPMIN'], h['RPMAX'], h['RTMAX'], h['NP'], h['NT'], coef,
                                z_ref=float(np.mean(t1.data['Z'])))
    dm = add_bin_average(distortion_matrix_rect(rp_d, rt_d, rp, rt, coef), int(h['NP']), int(h['NT']), coef)
    directory = Path(directory)
    dmat = write_distortion_file(directory / f'dmat_coef{coef}_{rp_d.size}.fits', h, coef, dm, rp, rt, z)
    cov = write_covariance_file(directory / f'cov_{rp_d.size}.fits', covariance(rp_d, rt_d))
    return dmat, cov


def dmat_file_configs(directory, golden, config='auto', coef=2, item_options=None):
    """Configuration files of the `distortion-file` case: the golden config ``config`` (tests/golden/configs/<config>) with
    every correlation pointed at its own distortion file (model grid ``coef`` times finer than its data grid) and covariance
    file, written under ``directory`` by :func:`write_dmat_file_case`.  ``item_options``: text added to every [model]
    section.  Returns the main file's path relative to ``directory`` (a search directory for build_problem)."""
    import re
    from pathlib import Path
    from .tables import read_tables
    directory, golden = Path(directory), Path(golden)
    cfg = directory / 'configs' / f'dmatfile_{config}'
    cfg.mkdir(parents=True, exist_ok=True)
    main = (golden / 'configs' / config / 'main.ini').read_text()
    items = re.search(r'ini files = (.*)', main).group(1).split()
    names = [Path(i).name for i in items]
    main = re.sub(r'ini files = .*', 'ini files = ' + ' '.join(f'configs/dmatfile_{config}/{n}' for n in names), main)
    (cfg / 'main.ini').write_text(main)
    for name in names:
        text = (golden / 'configs' / config / name).read_text()
        data_file = re.search(r'filename = (.*)', text).group(1).strip()
        dmat, cov = write_dmat_file_case(directory, read_tables(golden / data_file), coef=coef)
        text = text.replace('[data]', f'[data]\ndistortion-file = {dmat}\ncovariance-file = {cov}', 1)
        if item_options:
            text = text.replace('[model]', '[model]\n' + item_options)
        (cfg / name).write_text(text)
    return f'configs/dmatfile_{config}/main.ini'


def covariance(rp, rt):
    """SPD covariance: variance ~ 1/r^2, Kronecker-exponential correlations in (rp, rt).  (Kept per grid; read-only.)"""
    return _kept('cov', (rp, rt), (), lambda: _covariance(rp, rt))


def inverse_masked_covariance(rp, rt, mask):
    """``inv(covariance(rp, rt)[mask][:, mask])`` - what `CorrItem.inv_masked_cov` computes (reference vega/utils.py:271-298) -
    kept per (grid, mask), read-only: `item.set_covariance(cov, inv_masked_cov=...)`."""
    mask = np.asarray(mask, dtype=bool)
    return _kept('icov', (rp, rt, mask), (), lambda: np.linalg.inv(covariance(rp, rt)[:, mask][mask, :]))


def _covariance(rp, rt):
    rp = np.asarray(rp, dtype=float)
    rt = np.asarray(rt, dtype=float)
    r = np.sqrt(rp**2 + rt**2)
    sigma = 2e-4 * 50. / np.maximum(r, 5.)
    corr = 0.25**(np.abs(rp[:, None] - rp[None, :]) / 4.)
    corr *= 0.15**(np.abs(rt[:, None] - rt[None, :]) / 4.)
    corr *= sigma[:, None]
    corr *= sigma[None, :]
    return corr


def global_covariance(grids, coupling=0.08):
    """SPD covariance of the concatenated data vectors of several correlations (the `global-cov-file` case,
    reference vega/vega_interface.py:888-954): the items' own :func:`covariance` blocks on the diagonal and, between
    two items, ``coupling`` times the geometric mean of the variances with the same exponential fall-off in the
    separation of the bin centres - a cross-covariance a joint analysis really has.  ``grids``: [(rp, rt), ...] in
    item order.  Diagonally dominant enough to stay positive definite (checked by the Cholesky factor at use)."""
    blocks = [covariance(rp, rt) for rp, rt in grids]
    sizes = [b.shape[0] for b in blocks]
    off = np.concatenate([[0], np.cumsum(sizes)])
    cov = np.zeros((off[-1], off[-1]))
    for i, b in enumerate(blocks):
        cov[off[i]:off[i + 1], off[i]:off[i + 1]] = b
    for i in range(len(blocks)):
        for j in range(i + 1, len(blocks)):
            rp_i, rt_i = (np.asarray(a, dtype=float) for a in grids[i])
            rp_j, rt_j = (np.asarray(a, dtype=float) for a in grids[j])
            sig = np.sqrt(np.outer(np.diag(blocks[i]), np.diag(blocks[j])))
            cross = coupling * sig * 0.25**(np.abs(np.abs(rp_i[:, None]) - np.abs(rp_j[None, :])) / 4.)
            cross *= 0.15**(np.abs(rt_i[:, None] - rt_j[None, :]) / 4.)
            cov[off[i]:off[i + 1], off[j]:off[j + 1]] = cross
            cov[off[j]:off[j + 1], off[i]:off[i + 1]] = cross.T
    return cov


def write_global_covariance(path, cov):
    """A `global-cov-file`: HDU 1 with the vector column COV (reference vega/vega_interface.py:899-900)."""
    from . import fitslite
    cov = np.asarray(cov, dtype=float)
    fitslite.write_tables(str(path), [('GLOBALCOV', [('COV', f'{cov.shape[1]}D', cov)])], overwrite=True)
    return path


def walkers(theta_fid, names, n, varied=None, seed=SEED, scale=0.02, limits=None):
    """``theta_b = theta_fid + scale * |theta_fid| * N(0, 1)`` on the ``varied`` names
    (all non-sentinel parameters when None), clipped to ``limits`` = {name: (lo, hi)}."""
    rng = np.random.default_rng(seed)
    theta_fid = np.asarray(theta_fid, dtype=float)
    theta = np.tile(theta_fid, (n, 1))
    for j, name in enumerate(names):
        if varied is not None and name not in varied:
            continue
        if name == 'qso_rad_lifetime':
            continue
        g = rng.standard_normal(n)
        if theta_fid[j] == 0:
            theta[:, j] = 0.01 * scale / 0.02 * g
        else:
            theta[:, j] = theta_fid[j] + scale * abs(theta_fid[j]) * g
        if limits is not None and name in limits:
            lo, hi = limits[name]
            theta[:, j] = np.clip(theta[:, j], lo, hi)
    return theta


def blinded_data_vector(data_vec):
    """Deterministic stand-in for a blinded data vector (the `DA_BLIND` column of a `desi_dr3` file)."""
    data_vec = np.asarray(data_vec, dtype=float)
    return 1.02 * data_vec + 1e-6 * np.sin(0.37 * np.arange(data_vec.size))


BLINDING_SHIFTS = {'growth_rate': 0.03, 'ap': -0.01, 'bias_hcd': 0.004, 'beta_LYA': 0.02}


def blinding_offsets(shifts=None):
    """Offsets v whose blinding p += pi - exp(v^2) (reference vega/utils.py:375-393) moves each parameter by the
    wanted amount."""
    shifts = BLINDING_SHIFTS if shifts is None else shifts
    return {name: float(np.sqrt(np.log(np.pi - d))) for name, d in shifts.items()}


def write_data_file(path, source, with_distortion=True, with_covariance=True, extra_header=None, blind_data=None):
    """A correlation data file in the layout the reference reads (vega/data.py:285-421): HDU 1 = RP, RT, Z, DA (+ the
    synthetic distortion matrix `DM` and covariance `CO` of this module as vector columns) with the grid keywords,
    HDU 2 = the model-grid coordinates DMRP, DMRT, DMZ.  ``source`` is a reference-format table list
    (vega_amd.tables.read_tables) whose grids and data vector are reused."""
    from . import fitslite
    t1, t2 = source[0], source[1]
    rp, rt = np.asarray(t1.data['RP'], dtype=float), np.asarray(t1.data['RT'], dtype=float)
    cols = [('RP', 'D', rp), ('RT', 'D', rt), ('Z', 'D', t1.data['Z']), ('DA', 'D', t1.data['DA'])]
    n = rp.size
    if blind_data is not None:      # a blinded data vector next to DA (reference vega/data.py:313-327)
        cols.append(('DA_BLIND', 'D', np.asarray(blind_data, dtype=float)))
    if with_distortion:
        cols.append(('DM', f'{n}D', distortion_matrix(np.asarray(t2.data['DMRP'], dtype=float),
                                                     np.asarray(t2.data['DMRT'], dtype=float))))
    if with_covariance:
        cols.append(('CO', f'{n}D', covariance(rp, rt)))
    hdr = {k: t1.header[k] for k in ('RPMIN', 'RPMAX', 'RTMAX', 'NP', 'NT')}
    hdr.update(extra_header or {})
    fitslite.write_tables(str(path), [
        ('COR', cols, hdr),
        ('DMATTRI', [('DMRP', 'D', t2.data['DMRP']), ('DMRT', 'D', t2.data['DMRT']), ('DMZ', 'D', t2.data['DMZ'])])],
        overwrite=True)
    return path


PICCA_COSMOLOGY_HEADER = {'OMEGAM': 0.315, 'OMEGAK': 0., 'OMEGAR': 7.9e-5, 'WL': -1.}

METAL_MATRIX_SECTION = """[metal-matrix]
rebin_factor = 2
alpha_LYA = 2.9
alpha_SiII(1260) = 1.
alpha_SiIII(1207) = 1.
alpha_SiII(1193) = 1.
alpha_SiII(1190) = 1.
alpha_CIV(eff) = 1.
z_ref_objects = 2.25
z_evol_objects = 1.44
z_bins_objects = 200
"""


def write_stacked_deltas(path, n_pix=1200):
    """A stacked-delta file in the layout `new_metals` reads (reference vega/metals.py:405-409): LOGLAM, WEIGHT of
    the forest pixels between 3600 and 5500 Angstrom, weights peaking mid-forest."""
    from . import fitslite
    loglam = np.linspace(np.log10(3600.), np.log10(5500.), n_pix)
    lam = 10**loglam
    weight = 0.1 + np.exp(-((lam - 4300.) / 600.)**2) * (1 + 0.2 * np.sin(lam / 37.))
    fitslite.write_tables(str(path), [('STACK', [('LOGLAM', 'D', loglam), ('WEIGHT', 'D', weight)])], overwrite=True)
    return path


def write_object_catalog(path, n_obj=4000, seed=SEED):
    """A quasar catalogue with a Z column (reference vega/metals.py:434-435)."""
    from . import fitslite
    z = np.clip(np.random.default_rng(seed + 17).normal(2.4, 0.3, n_obj), 1.8, 3.6)
    fitslite.write_tables(str(path), [('CAT', [('Z', 'D', z)])], overwrite=True)
    return path
