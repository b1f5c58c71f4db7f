"""iminuit-shaped adapter over the repo's own MIGRAD restatement (vega_amd/migrad.py), for fixture generation only.

iminuit is absent from this image.  So that the UNMODIFIED reference can walk its own fit drivers - `Minimizer.minimize`
(bias pre-fit, then the full fit), `Analysis.chi2_scan`, `initialize_monte_carlo`, `Analysis.run_monte_carlo` - this module
offers the few attributes those drivers touch (vega/minimizer.py:66-97, :104-200; vega/analysis.py:53-122, :279-302) and
runs `vega_amd.migrad` underneath.  Fixtures made this way pin the reference's CONTROL FLOW around the minimiser (grids,
pinned parameters, start values, seeding, which results are kept); MIGRAD's arithmetic itself is pinned by the reference's
own golden value (tests/test_vega.py:18), which the restatement reproduces to 1e-11.
"""
import numpy as np


class _ByName(dict):
    def to_dict(self):
        return dict(self)


class _FMin:
    def __init__(self, fval, is_valid, hesse_failed, has_accurate_covar, edm, nfcn):
        self.fval, self.is_valid, self.hesse_failed = fval, is_valid, hesse_failed
        self.has_accurate_covar, self.edm, self.nfcn = has_accurate_covar, edm, nfcn

    def __repr__(self):
        return f'FMin(fval={self.fval!r}, edm={self.edm!r}, nfcn={self.nfcn}, is_valid={self.is_valid})'


class Minuit:
    def __init__(self, fcn, name=None, **values):
        self._fcn = fcn
        self._names = list(name) if name is not None else list(values)
        self.values = _ByName((n, float(values[n])) for n in self._names)
        self.errors = _ByName((n, 0.1) for n in self._names)
        self.limits = _ByName((n, (None, None)) for n in self._names)
        self.fixed = _ByName((n, False) for n in self._names)
        self.errordef = 1.0
        self.print_level = 0
        self.tol = 0.1
        self.fmin = None
        self.covariance = None

    @property
    def params(self):
        return [(n, self.values[n], self.errors[n]) for n in self._names]

    def migrad(self, ncall=None):
        from vega_amd.migrad import MigradMinimizer
        names = self._names

        def evaluate(theta, fit_index):
            return np.array([self._fcn(*row) for row in theta])
        limits = [self.limits[n] if self.limits[n] is not None else (None, None) for n in names]
        fitter = MigradMinimizer(evaluate, names, [self.values[n] for n in names], [self.errors[n] for n in names], limits,
                                 tol=self.tol, errordef=self.errordef, maxfcn=ncall or 100000)
        res = fitter.minimize(1, fixed=tuple(n for n in names if self.fixed[n]), prefit_bias=False)
        for j, n in enumerate(names):
            self.values[n] = float(res.values[0, j])
            if not self.fixed[n]:
                self.errors[n] = float(res.errors[0, j])
        self.covariance = res.covariance[0]
        self.fmin = _FMin(float(res.fval[0]), bool(res.is_valid[0]), bool(res.hesse_failed[0]),
                          bool(res.has_accurate_covar[0]), float(res.edm[0]), int(res.nfcn[0]))
        return self
