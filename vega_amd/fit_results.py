"""Reader of a fit's result file (reference vega/postprocess/fit_results.py:33-141, the consumer of what
`vega_amd.output.Output.write_results` / the reference's `Output.write_results` write): the BESTFIT table and the
MODEL_<correlation> HDUs, through `fitslite`.  The Gaussian chain of the reference's reader (getdist) is not built.

Extension names are upper case in the file (astropy stores `hdu.name` that way), so the correlation names come back
upper case from the HDU names and the columns are matched without regard to case, as astropy matches them; the
`correlations` dictionary is keyed by the lower-case name (reference :131-139).  The per-correlation statistics are read
from the cards the writer writes (`masked_size`, `chisq`, `reduced_chisq`, `p_value`, `marg_coeff_<i>`, reference
vega/output.py:214-228) - the reference's reader asks for `SIZE` / `CHISQ` / ... and so always returns None for them.
"""
from dataclasses import dataclass

import numpy as np
from scipy import stats

from . import fitslite


@dataclass
class CorrelationOutput:
    model: np.ndarray
    model_mask: np.ndarray
    data: np.ndarray
    data_mask: np.ndarray
    variance: np.ndarray
    rp: np.ndarray
    rt: np.ndarray
    z: np.ndarray
    size: int = None
    chisq: float = None
    reduced_chisq: float = None
    p_value: float = None
    bestfit_marg_coeff: np.ndarray = None


def _column(hdu, name):
    for col in hdu.columns.names:
        if col.upper() == name.upper():
            return hdu.data[col]
    raise KeyError(name)


def _card(header, key, default=None):
    for k, v in header.items():
        if k.upper() == key.upper():
            return v
    return default


class FitResults:
    def __init__(self, path, results_only=False):
        hdul = fitslite.open(str(path))
        named = {str(h.header.get('EXTNAME', '')).strip().upper(): h for h in hdul[1:]}
        best = named['BESTFIT']
        self.chisq = best.header['FVAL']
        self.valid = best.header['VALID']
        self.accurate = best.header['ACCURATE']
        self.names = np.array([str(n).strip() for n in best.data['names']])
        self.mean = np.asarray(best.data['values'], dtype=float)
        self.cov = np.asarray(best.data['covariance'], dtype=float)
        self.params = dict(zip(self.names, self.mean))
        self.sigmas = dict(zip(self.names, np.asarray(best.data['errors'], dtype=float)))
        self.num_pars = len(self.names)
        self.marg_coeff = {}
        if not results_only:
            self.read_correlations([(n, h) for n, h in named.items() if n.startswith('MODEL')])

    def read_correlations(self, model_hdus):
        if not model_hdus:
            raise ValueError('No model HDUs found in the fit results file.')
        self.correlations = {}
        self.num_data_points = 0
        for hdu_name, hdu in model_hdus:
            corr = hdu_name.split('_', 1)[1]
            data, data_mask = _column(hdu, corr + '_DATA'), np.asarray(_column(hdu, corr + '_MASK'), dtype=bool)
            self.num_data_points += int(data_mask.sum())
            coeff, i = [], 0
            while _card(hdu.header, f'marg_coeff_{i}') is not None:
                coeff.append(_card(hdu.header, f'marg_coeff_{i}'))
                i += 1
            key = corr.lower()
            self.marg_coeff[key] = np.array(coeff)
            self.correlations[key] = CorrelationOutput(
                _column(hdu, corr + '_MODEL'), np.asarray(_column(hdu, corr + '_MODEL_MASK'), dtype=bool), data, data_mask,
                _column(hdu, corr + '_VAR'), _column(hdu, corr + '_RP'), _column(hdu, corr + '_RT'), _column(hdu, corr + '_Z'),
                size=_card(hdu.header, 'masked_size'), chisq=_card(hdu.header, 'chisq'),
                reduced_chisq=_card(hdu.header, 'reduced_chisq'), p_value=_card(hdu.header, 'p_value'),
                bestfit_marg_coeff=self.marg_coeff[key])
        dof = self.num_data_points - self.num_pars
        self.p_value = float(1 - stats.chi2.cdf(self.chisq, dof))
        self.reduced_chisq = self.chisq / dof
