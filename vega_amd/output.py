"""Result files in the reference's FITS layout (reference vega/output.py), written with `fitslite`.

* a fit (`Output.write_results`, reference :37-123): one HDU ``MODEL_<correlation>`` per correlation (columns
  ``<name>_MODEL / _MODEL_MASK / _MASK / _DATA / _VAR / _RP / _RT / _Z / _NB``, the parameters and the correlation's fit
  statistics as HIERARCH cards, reference :144-235), HDU ``BESTFIT`` (names, values, errors, covariance; FVAL / VALID /
  ACCURATE, reference :237-289) and HDU ``SCAN`` (one column per parameter and result of the chi2 scan, reference :291-349) -
  what `vega.postprocess.fit_results.FitResults` and `mc_start_from_fit` read back;
* Monte Carlo (`write_monte_carlo`, reference :442-520): HDU 'Bestfit' (names, values, errors, covariance), HDU 'FitInfo'
  (chisq, valid_minima, valid_hesse, failed_mask) and HDU 'Mocks' (one vector column per correlation), one file per rank
  (`monte_carlo_<rank>.fits`) as `bin/run_vega_mc_mpi.py:67-71` writes them.

``write_cf``: the ``Xi_<name>`` component HDUs (reference :375-440) from `VegaInterface.model_components`.  Not written: the ``write_pk`` HDUs (P(k, mu) grids are never formed: the mu sums are fused into the
spectrum kernel) and the hdf flavour (h5py is not a dependency): both raise.
"""
import os
from pathlib import Path

import numpy as np

from . import fitslite


def monte_carlo_tables(analysis):
    """[(extname, columns)] for fitslite.write_tables from a MonteCarlo driver that has run."""
    if getattr(analysis, 'mc_mocks', None) is None:
        raise ValueError('No Monte Carlo results found. Run run_monte_carlo() first.')
    tables = []
    bestfits = getattr(analysis, 'mc_bestfits', None)
    if bestfits:
        # (rows of mc_bestfits / mc_covariances are the successful fits only, as in the reference: analysis.py:279-297)
        names = np.array(list(bestfits.keys()))
        values = np.array([bestfits[n][:, 0] for n in names])          # [parameters][successful mocks]
        errors = np.array([bestfits[n][:, 1] for n in names])
        cov = np.array(analysis.mc_covariances)                         # [mocks][parameters][parameters]
        cov = cov.reshape(values.shape[1] * len(names), len(names)).T   # reference :468
        width = max(len(n) for n in names)
        tables.append(('Bestfit', [('names', f'{width}A', names),
                                   ('values', f'{values.shape[1]}D', values),
                                   ('errors', f'{values.shape[1]}D', errors),
                                   ('covariance', f'{cov.shape[1]}D', cov)]))
        tables.append(('FitInfo', [('chisq', 'D', np.asarray(analysis.mc_chisq, dtype=float)),
                                   ('valid_minima', 'L', np.asarray(analysis.mc_valid_minima)),
                                   ('valid_hesse', 'L', np.asarray(analysis.mc_valid_hesse)),
                                   ('failed_mask', 'L', np.asarray(analysis.mc_failed_mask))]))
    mock_cols = []
    vega = getattr(analysis, 'vega', None)
    items = vega.problem.items if vega is not None else {}
    for name, mocks in analysis.mc_mocks.items():
        table = np.asarray(mocks, dtype=float)
        if name in items and table.shape[1] != items[name].data_vec.size:
            # the reference keeps each mock on the full data grid, NaN outside the mask (data.py:749-753)
            full = np.full((table.shape[0], items[name].data_vec.size), np.nan)
            full[:, items[name].data_mask] = table
            table = full
        mock_cols.append((name, f'{table.shape[1]}D', table))
    tables.append(('Mocks', mock_cols))
    return tables


def write_monte_carlo(analysis, directory, cpu_id=None, overwrite=False):
    """Write `monte_carlo.fits` (or `monte_carlo_<cpu_id>.fits`) under ``directory``; returns the path."""
    directory = Path(directory)
    directory.mkdir(parents=True, exist_ok=True)
    path = directory / ('monte_carlo.fits' if cpu_id is None else f'monte_carlo_{cpu_id}.fits')
    fitslite.write_tables(str(path), monte_carlo_tables(analysis), overwrite=overwrite)
    return path


# ------------------------------------------------------------------------------------------ a fit's results
def _pad(array, size, pad_value=np.nan):
    array = np.asarray(array)
    return np.pad(array, (0, size - len(array)), constant_values=pad_value)


def model_tables(items, corr_funcs, params, bestfit_corr_stats=None):
    """[(extname, columns, header)] of the MODEL_<name> HDUs (reference vega/output.py:144-235).  ``items``: the problem's
    correlation items; ``corr_funcs``: {name: model vector} as compute_model returns it."""
    tables = []
    for name, cf in corr_funcs.items():
        it = items[name]
        cf = np.asarray(cf, dtype=float)
        n = len(cf)
        if len(it.data_vec) > n:
            raise ValueError('Data coordinate grid is larger than the model grid.')
        variance = it.variance if it.variance is not None else \
            (np.ones(it.data_vec.size) if it.cov is None else np.diag(it.cov))
        cols = [(name + '_MODEL', 'D', cf),
                (name + '_MODEL_MASK', 'L', _pad(it.model_mask, n, False)),
                (name + '_MASK', 'L', _pad(it.data_mask, n, False)),
                (name + '_DATA', 'D', _pad(it.data_vec, n)),
                (name + '_VAR', 'D', _pad(variance, n)),
                (name + '_RP', 'D', _pad(it.dist_grid.rp, n)),
                (name + '_RT', 'D', _pad(it.dist_grid.rt, n))]
        z = None if it.model_grid.z is None else np.atleast_1d(np.asarray(it.model_grid.z, dtype=float))
        cols.append((name + '_Z', 'D', np.zeros(n) if z is None or n < z.size else _pad(z, n)))        # (reference :199-205)
        if it.nb is not None:
            cols.append((name + '_NB', 'K', _pad(np.asarray(it.nb, dtype=np.int64), n, 0)))
        header = {}
        for par, val in params.items():
            header[par] = float(val) if isinstance(val, (float, np.floating)) else val
        if bestfit_corr_stats is not None:
            for par, val in bestfit_corr_stats[name].items():
                if par == 'bestfit_marg_coeff':
                    if val is not None:
                        for i, v in enumerate(val):
                            header[f'marg_coeff_{i}'] = float(v)
                else:
                    header[par] = val
        tables.append(('MODEL_' + name, cols, header))
    return tables


def bestfit_table(names, values, errors, covariance, fval, valid, accurate):
    """The BESTFIT HDU (reference vega/output.py:237-289)."""
    names = np.array(list(names))
    width = max(len(n) for n in names)
    cov = np.asarray(covariance, dtype=float).reshape(len(names), len(names))
    cols = [('names', f'{width}A', names), ('values', 'D', np.asarray(values, dtype=float)),
            ('errors', 'D', np.asarray(errors, dtype=float)), ('covariance', f'{len(names)}D', cov)]
    header = {'FVAL': (float(fval), 'Bestfit chi^2 value'), 'VALID': (bool(valid), 'Flag for valid fit'),
              'ACCURATE': (bool(accurate), 'Flag for accurate fit')}
    return ('BESTFIT', cols, header)


def scan_table(scan_results, grids=None):
    """The SCAN HDU (reference vega/output.py:291-349): ``scan_results`` = one dict per grid point."""
    names = np.array(list(scan_results[0].keys()))
    width = max(len(n) for n in names)
    results = np.array([[res[par] for par in names] for res in scan_results], dtype=float)
    # (the reference puts the names column next to the per-point columns: astropy's from_columns makes the table as long as the
    # longest of them and pads the shorter ones - empty strings, zeros)
    rows = max(len(names), len(results))
    name_col = np.array(list(names) + [''] * (rows - len(names)), dtype=f'U{width}')
    cols = [('names', f'{width}A', name_col)] + [(str(n), 'D', _pad(results[:, j], rows, 0.)) for j, n in enumerate(names)]
    header = {}
    for par, grid in (grids or {}).items():
        header[par + '_min'] = (float(grid[0]), 'Grid start for ' + par)
        header[par + '_max'] = (float(grid[-1]), 'Grid end for ' + par)
        header[par + '_num_bins'] = (int(len(grid)), 'Grid size for ' + par)
    return ('SCAN', cols, header)


def component_table(name, components):
    """The ``Xi_<name>`` HDU of `write_cf` (reference vega/output.py:375-440): columns ``raw_<part>_core`` (the model's saved
    `xi`) and ``distorted_<part>_core`` (`xi_distorted`), parts in the reference's order; shorter columns are zero-padded to
    the table's length, as astropy pads them."""
    cols = []
    for prefix, key in (('raw_', 'xi'), ('distorted_', 'xi_distorted')):
        for part in ('peak', 'smooth', 'full'):
            for comp, arr in components.get(key, {}).get(part, {}).items():
                label = 'core' if comp == 'core' else f'{comp[0]}_{comp[1]}'
                cols.append((f'{prefix}{part}_{label}', 'D', np.asarray(arr, dtype=float)))
    rows = max(len(c[2]) for c in cols)
    return ('Xi_' + name, [(n, f, _pad(a, rows, 0.)) for n, f, a in cols], {})


class Output:
    """The reference's ``vega.output`` object for a fit's results (reference vega/output.py:9-123): built from the
    ``[output]`` section; ``write_results(corr_funcs, params, minimizer, bestfit_corr_stats, scan_results)``."""

    def __init__(self, config, items, analysis=None):
        get = config.get if config is not None else (lambda key, default=None: default)
        self.items = items
        self.analysis = analysis
        self.type = get('type', 'fits')
        self.overwrite = str(get('overwrite', False)).lower() in ('true', '1', 'yes')
        filename = get('filename', None)
        self.outfile = None if filename is None else os.path.expandvars(filename)
        flag = (lambda key: config.getboolean(key, False)) if hasattr(config, 'getboolean') else (lambda key: False)
        self.output_cf, self.output_pk = flag('write_cf'), flag('write_pk')
        self.mc_output = get('mc_output', None)

    def write_results(self, corr_funcs, params, minimizer=None, bestfit_corr_stats=None, scan_results=None, models=None):
        """``minimizer``: ``vega.minimizer`` / a `minimizer.FitResult` (fit 0 is written) or None."""
        if self.type not in ('fits',):
            raise NotImplementedError(f'output type {self.type!r}: only the fits flavour is written')
        if self.output_pk:
            raise NotImplementedError('write_pk: P(k, mu) grids are never formed (the mu sums are fused into the spectrum kernel)')
        if self.output_cf and not isinstance(models, dict):
            raise ValueError('write_cf: pass the components (VegaInterface.model_components(params)) as `models`')
        if self.outfile is None:
            raise ValueError('[output] filename is not set')
        tables = model_tables(self.items, corr_funcs, params, bestfit_corr_stats)
        minimizer = getattr(minimizer, 'fit', minimizer)         # (vega.minimizer: a MinimizerView of the FitResult)
        if minimizer is not None:
            accurate = getattr(minimizer, 'has_accurate_covar', None)
            tables.append(bestfit_table(minimizer.names, minimizer.values[0], minimizer.errors[0], minimizer.covariance[0],
                                        minimizer.fval[0], minimizer.is_valid[0],
                                        accurate[0] if accurate is not None else not minimizer.hesse_failed[0]))
        if self.output_cf:
            for name, components in models.items():
                if not (isinstance(components, dict) and 'xi' in components):
                    raise ValueError('write_cf: `models` must be the dictionary VegaInterface.model_components returns')
                tables.append(component_table(name, components))
        if scan_results is not None:
            if minimizer is None:
                raise ValueError('scan results are written next to a fit')
            tables.append(scan_table(scan_results, getattr(self.analysis, 'grids', None)))
        if self.outfile[-5:] != '.fits':
            self.outfile += '.fits'
        Path(self.outfile).parent.mkdir(parents=True, exist_ok=True)
        fitslite.write_tables(self.outfile, tables, overwrite=self.overwrite)
        return self.outfile

    def write_monte_carlo(self, cpu_id=None):
        """``vega.output.write_monte_carlo(rank)`` of the reference's launchers (vega/output.py:442-520,
        bin/run_vega_mc_mpi.py:67-71): `monte_carlo[_<rank>].fits` under ``[output] mc_output``, or under
        ``monte_carlo/`` next to the fit's result file."""
        if self.analysis is None:
            raise ValueError('Output.write_monte_carlo requires an Analysis object')
        if not getattr(self.analysis, 'has_monte_carlo', False):
            raise ValueError('No Monte Carlo results found. Run Analysis.run_monte_carlo() first.')
        if self.mc_output is None and self.outfile is None:
            raise ValueError('[output] mc_output / filename are not set')
        directory = Path(self.outfile).parent / 'monte_carlo' if self.mc_output is None else Path(self.mc_output)
        return write_monte_carlo(self.analysis, directory, cpu_id=cpu_id, overwrite=self.overwrite)
