"""Monte-Carlo result tables in the reference's FITS layout (reference vega/output.py:442-520,
``Output.write_monte_carlo``): HDU 'Bestfit' (names, values, errors, covariance), HDU 'FitInfo' (chisq,
valid_minima, valid_hesse, failed_mask) and HDU 'Mocks' (one vector column per correlation), one file per rank
(`monte_carlo_<rank>.fits`) as `bin/run_vega_mc_mpi.py:67-71` writes them.
"""
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
