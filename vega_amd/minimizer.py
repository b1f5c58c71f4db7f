"""Batched variable-metric minimiser driven by the engine's batch evaluation.

The reference minimises one chi2 at a time with iminuit's MIGRAD (reference vega/minimizer.py:39-103:
optional bias-only pre-fit, then the full fit with limits, ``errordef = 1``; HESSE errors and the covariance are
read back at vega/analysis.py:294-302).  iminuit is a third-party dependency that is absent from this image, so
this is NOT a port of MIGRAD: it is a variable-metric (BFGS) minimiser with Minuit's conventions -

* parameters with limits are minimised in Minuit's internal coordinates (sine transform for two-sided limits,
  square-root transforms for one-sided ones), so the bounds can never be violated;
* convergence is declared on the estimated distance to the minimum  EDM = g^T V g / 2 < 0.002 * tol * errordef;
* errors come from a numerical Hessian at the minimum: covariance = 2 * errordef * H^-1, mapped to external
  coordinates with the transform's Jacobian;

- restructured for the GPU: F independent fits (Monte-Carlo mocks) advance in lock-step, and every stage
(finite-difference gradients, line-search trial points, Hessian stencils) is ONE batch of parameter points for
the engine.  Parity with MIGRAD is statistical, not bitwise: best fits agree within a small fraction of the
reported errors (see tests/test_minimizer_gpu.py and DESIGN.md).
"""
from dataclasses import dataclass

import numpy as np

SENTINEL = 1e99     # chi2 >= this marks a point the model could not be evaluated at (engine returns 1e100)


class ParameterTransform:
    """Minuit's internal <-> external parameter maps."""

    def __init__(self, limits):
        self.lo = np.array([(-np.inf if lim[0] is None else lim[0]) for lim in limits], dtype=float)
        self.hi = np.array([(np.inf if lim[1] is None else lim[1]) for lim in limits], dtype=float)
        lo_f, hi_f = np.isfinite(self.lo), np.isfinite(self.hi)
        self.both, self.lower, self.upper = lo_f & hi_f, lo_f & ~hi_f, ~lo_f & hi_f

    def to_internal(self, ext):
        ext = np.asarray(ext, dtype=float)
        x = ext.copy()
        b, lo, hi = self.both, self.lo, self.hi
        if b.any():
            frac = np.clip(2 * (ext[..., b] - lo[b]) / (hi[b] - lo[b]) - 1, -1 + 1e-12, 1 - 1e-12)
            x[..., b] = np.arcsin(frac)
        if self.lower.any():
            m = self.lower
            x[..., m] = np.sqrt(np.maximum((ext[..., m] - lo[m] + 1)**2 - 1, 0.))
        if self.upper.any():
            m = self.upper
            x[..., m] = np.sqrt(np.maximum((hi[m] - ext[..., m] + 1)**2 - 1, 0.))
        return x

    def to_external(self, x):
        x = np.asarray(x, dtype=float)
        ext = x.copy()
        b, lo, hi = self.both, self.lo, self.hi
        if b.any():
            ext[..., b] = lo[b] + 0.5 * (hi[b] - lo[b]) * (np.sin(x[..., b]) + 1)
        if self.lower.any():
            m = self.lower
            ext[..., m] = lo[m] - 1 + np.sqrt(x[..., m]**2 + 1)
        if self.upper.any():
            m = self.upper
            ext[..., m] = hi[m] + 1 - np.sqrt(x[..., m]**2 + 1)
        return ext

    def jacobian(self, x):
        """d ext / d int (diagonal)."""
        x = np.asarray(x, dtype=float)
        jac = np.ones_like(x)
        b = self.both
        if b.any():
            jac[..., b] = 0.5 * (self.hi[b] - self.lo[b]) * np.cos(x[..., b])
        for m, sign in ((self.lower, 1.), (self.upper, -1.)):
            if m.any():
                jac[..., m] = sign * x[..., m] / np.sqrt(x[..., m]**2 + 1)
        return jac


@dataclass
class FitResult:
    names: list
    values: np.ndarray          # [F, P] external best-fit values
    errors: np.ndarray          # [F, P]
    covariance: np.ndarray      # [F, P, P]
    fval: np.ndarray            # [F]
    edm: np.ndarray             # [F]
    is_valid: np.ndarray        # [F] converged and positive-definite Hessian
    hesse_failed: np.ndarray    # [F]
    nfcn: np.ndarray            # [F] function evaluations spent on the fit
    n_iter: np.ndarray          # [F]

    def as_dict(self, i=0):
        return {n: float(v) for n, v in zip(self.names, self.values[i])}


class MinimizerView:
    """One fit of a :class:`FitResult` behind the attribute names of the reference's ``Minimizer`` after ``minimize()``
    (vega/minimizer.py:105-187: ``values`` / ``errors`` dictionaries, ``covariance``, ``fmin``, ``minuit``, ``params``) -
    what ``run_vega``, ``Output`` and user scripts read from ``vega.minimizer``.  ``fit`` is the FitResult itself."""

    def __init__(self, fit, index=0):
        from types import SimpleNamespace
        self.fit, self.index = fit, index
        i = index
        accurate = getattr(fit, 'has_accurate_covar', None)
        accurate = bool(accurate[i]) if accurate is not None else not bool(fit.hesse_failed[i])
        self.values = {n: float(v) for n, v in zip(fit.names, fit.values[i])}
        self.errors = {n: float(v) for n, v in zip(fit.names, fit.errors[i])}
        self.covariance = np.array(fit.covariance[i])
        self.fmin = SimpleNamespace(fval=float(fit.fval[i]), edm=float(fit.edm[i]), is_valid=bool(fit.is_valid[i]),
                                    hesse_failed=bool(fit.hesse_failed[i]), has_accurate_covar=accurate,
                                    nfcn=int(fit.nfcn[i]))
        self.minuit = SimpleNamespace(valid=bool(fit.is_valid[i]), accurate=accurate, fval=float(fit.fval[i]))
        self.params = [SimpleNamespace(name=n, value=self.values[n], error=self.errors[n]) for n in fit.names]


class BatchedMinimizer:
    """Minimise chi2 over the sampled parameters for many data realisations at once.

    ``evaluate(theta_ext [n, P], fit_index [n]) -> chi2 [n]`` evaluates arbitrary parameter points; ``fit_index``
    says which fit (mock) each point belongs to.
    """

    def __init__(self, evaluate, names, start, errors, limits, tol=0.1, errordef=1.0, max_iter=120):
        self.evaluate = evaluate
        self.names = list(names)
        self.start = np.asarray(start, dtype=float)
        self.step = np.asarray(errors, dtype=float)
        self.transform = ParameterTransform(limits)
        self.edm_goal = 0.002 * tol * errordef
        self.errordef = errordef
        self.max_iter = max_iter

    # ------------------------------------------------------------------ helpers
    def _f(self, x, fits):
        vals = np.asarray(self.evaluate(self.transform.to_external(x), fits), dtype=float)
        vals = np.where(np.isfinite(vals) & (vals < SENTINEL), vals, np.inf)
        self._nfcn += np.bincount(fits, minlength=self._nfcn.size)
        return vals

    def _gradient(self, x, f0, delta, free, fits):
        """Central differences on the free coordinates: gradient and diagonal second derivative."""
        n, P = x.shape
        nf = free.size
        xp = np.repeat(x, 2 * nf, axis=0).reshape(n, nf, 2, P)
        for a, j in enumerate(free):
            xp[:, a, 0, j] += delta[:, j]
            xp[:, a, 1, j] -= delta[:, j]
        vals = self._f(xp.reshape(-1, P), np.repeat(fits, 2 * nf)).reshape(n, nf, 2)
        g = np.zeros((n, P))
        g2 = np.zeros((n, P))
        d = delta[:, free]
        g[:, free] = (vals[:, :, 0] - vals[:, :, 1]) / (2 * d)
        g2[:, free] = (vals[:, :, 0] + vals[:, :, 1] - 2 * f0[:, None]) / d**2
        bad = ~np.isfinite(g).all(axis=1)
        g[bad] = 0.
        return g, g2, bad

    # ------------------------------------------------------------------ one minimisation stage
    def _stage(self, x, free, fits_all):
        F, P = x.shape
        free = np.asarray(free, dtype=int)
        jac0 = np.abs(self.transform.jacobian(x))
        sigma = np.where(jac0 > 1e-8, self.step / np.maximum(jac0, 1e-8), self.step)      # internal error guess
        V = np.zeros((F, P, P))
        f = self._f(x, fits_all)
        edm = np.full(F, np.inf)
        done = ~np.isfinite(f)                      # a start the model cannot evaluate is a failed fit
        n_iter = np.zeros(F, dtype=int)
        g = np.zeros((F, P))
        have_metric = np.zeros(F, dtype=bool)
        lambdas = np.array([0.25, 0.5, 1.0, 2.0])

        for it in range(self.max_iter):
            act = np.flatnonzero(~done)
            if act.size == 0:
                break
            xa, fa = x[act], f[act]
            delta = np.maximum(1e-3 * sigma[act], 1e-8)
            ga, g2, bad = self._gradient(xa, fa, delta, free, fits_all[act])
            # curvature-based error estimate: sigma^2 = 2 errordef / f''
            pos = g2 > 0
            sig_new = np.where(pos, np.sqrt(2 * self.errordef / np.where(pos, g2, 1.)), sigma[act])
            sigma[act] = np.clip(sig_new, 1e-3 * sigma[act], 1e3 * sigma[act])

            for a, i in enumerate(act):
                if not have_metric[i]:
                    V[i] = 0.
                    V[i][free, free] = np.where(pos[a, free], 1. / np.where(pos[a, free], g2[a, free], 1.),
                                                0.5 * sigma[i, free]**2 / self.errordef)
                    have_metric[i] = True
                else:
                    s = xa[a] - self._x_prev[i]
                    y = ga[a] - g[i]
                    sy = s @ y
                    if sy > 1e-12 * np.linalg.norm(s) * np.linalg.norm(y):
                        Vy = V[i] @ y
                        V[i] += (1 + (y @ Vy) / sy) * np.outer(s, s) / sy - (np.outer(Vy, s) + np.outer(s, Vy)) / sy
            g[act] = ga
            self._x_prev[act] = xa

            d = -np.einsum('fij,fj->fi', V[act], ga)
            slope = np.einsum('fi,fi->f', ga, d)
            uphill = slope >= 0
            for a in np.flatnonzero(uphill):        # metric lost positive definiteness: restart from the diagonal
                i = act[a]
                V[i] = 0.
                V[i][free, free] = 0.5 * sigma[i, free]**2 / self.errordef
                d[a] = -V[i] @ ga[a]
                slope[a] = ga[a] @ d[a]
            edm[act] = np.where(bad, np.inf, -0.5 * slope)
            # a gradient stencil that hit a point the model cannot evaluate ends the fit as a FAILED one (infinite EDM,
            # never reported as a valid minimum), not as a converged one
            self._bad[act[bad]] = True
            conv = (edm[act] < self.edm_goal) | bad
            done[act[conv]] = True
            n_iter[act] += 1
            keep = ~conv
            if not keep.any():
                continue
            act, xa, fa, d, slope = act[keep], xa[keep], fa[keep], d[keep], slope[keep]

            # line search: four trial steps in one batch, then one parabolic refinement
            trial = xa[:, None, :] + lambdas[None, :, None] * d[:, None, :]
            ft = self._f(trial.reshape(-1, P), np.repeat(fits_all[act], lambdas.size)).reshape(act.size, -1)
            lam_all = np.concatenate([np.zeros((act.size, 1)), np.tile(lambdas, (act.size, 1))], axis=1)
            f_all = np.concatenate([fa[:, None], ft], axis=1)
            best = np.argmin(f_all, axis=1)
            lam_b = lam_all[np.arange(act.size), best]
            f_b = f_all[np.arange(act.size), best]
            # parabola through (0, f0, slope) and the best trial (or the smallest trial when none improved)
            lam_ref = np.where(best > 0, lam_b, lambdas[0])
            f_ref = np.where(best > 0, f_b, ft[:, 0])
            curv = 2 * (f_ref - fa - slope * lam_ref) / lam_ref**2
            lam_p = np.where(curv > 0, -slope / np.where(curv > 0, curv, 1.), 2 * lam_ref)
            lam_p = np.clip(lam_p, 0.02 * lambdas[0], 4 * lambdas[-1])
            fp = self._f(xa + lam_p[:, None] * d, fits_all[act])
            better = fp < f_b
            lam_b = np.where(better, lam_p, lam_b)
            f_b = np.where(better, fp, f_b)
            stuck = ~(f_b < fa)
            # accept
            x[act] = xa + lam_b[:, None] * d
            f[act] = np.where(stuck, fa, f_b)
            for a in np.flatnonzero(stuck):         # no progress along this direction: forget the metric once,
                i = act[a]                          # give up when it happens with a fresh diagonal metric
                if self._stalled[i]:
                    done[i] = True
                self._stalled[i] = True
                have_metric[i] = False
            self._stalled[act[~stuck]] = False
        return x, f, edm, n_iter

    # ------------------------------------------------------------------ Hessian at the minimum
    def _hesse(self, x, f, sigma_int, free, fits_all):
        F, P = x.shape
        nf = free.size
        delta = 0.05 * sigma_int
        H = np.zeros((F, P, P))
        g, g2, _ = self._gradient(x, f, delta, free, fits_all)
        for j in free:
            H[:, j, j] = g2[:, j]
        pairs = [(a, b) for ai, a in enumerate(free) for b in free[ai + 1:]]
        if pairs:
            pts = np.repeat(x, 4 * len(pairs), axis=0).reshape(F, len(pairs), 4, P)
            for q, (a, b) in enumerate(pairs):
                for c, (sa, sb) in enumerate(((1, 1), (1, -1), (-1, 1), (-1, -1))):
                    pts[:, q, c, a] += sa * delta[:, a]
                    pts[:, q, c, b] += sb * delta[:, b]
            vals = self._f(pts.reshape(-1, P), np.repeat(fits_all, 4 * len(pairs))).reshape(F, len(pairs), 4)
            for q, (a, b) in enumerate(pairs):
                h = (vals[:, q, 0] - vals[:, q, 1] - vals[:, q, 2] + vals[:, q, 3]) / (4 * delta[:, a] * delta[:, b])
                H[:, a, b] = H[:, b, a] = h
        cov = np.zeros((F, P, P))
        failed = np.zeros(F, dtype=bool)
        idx = np.ix_(free, free)
        for i in range(F):
            Hi = H[i][idx]
            try:
                if not np.isfinite(Hi).all():
                    raise np.linalg.LinAlgError
                np.linalg.cholesky(Hi)
                cov[i][idx] = 2 * self.errordef * np.linalg.inv(Hi)
            except np.linalg.LinAlgError:
                failed[i] = True
        return cov, failed

    # ------------------------------------------------------------------ driver
    def minimize(self, n_fits=1, start=None, fixed=(), prefit_bias=True):
        """Run ``n_fits`` fits in lock-step.  ``start`` [F, P] overrides the configured starting values;
        ``fixed`` lists parameter names held at their start values."""
        P = len(self.names)
        F = int(n_fits)
        ext0 = np.tile(self.start, (F, 1)) if start is None else np.array(start, dtype=float).reshape(F, P)
        x = self.transform.to_internal(ext0)
        fits_all = np.arange(F)
        self._nfcn = np.zeros(F, dtype=np.int64)
        self._x_prev = x.copy()
        self._stalled = np.zeros(F, dtype=bool)
        self._bad = np.zeros(F, dtype=bool)
        free_all = np.array([j for j, n in enumerate(self.names) if n not in fixed], dtype=int)
        n_iter = np.zeros(F, dtype=int)

        # the reference first minimises over the bias parameters alone (vega/minimizer.py:66-86)
        bias = np.array([j for j in free_all if 'bias' in self.names[j]], dtype=int)
        if prefit_bias and 0 < bias.size < free_all.size:
            x, f, edm, it0 = self._stage(x, bias, fits_all)
            n_iter += it0
            self._stalled[:] = False
            self._bad[:] = False        # the full stage decides
        x, f, edm, it1 = self._stage(x, free_all, fits_all)
        n_iter += it1

        # errors: numerical Hessian in internal coordinates, mapped out with the transform's Jacobian
        jac = self.transform.jacobian(x)
        sig_guess = np.where(np.abs(jac) > 1e-8, self.step / np.maximum(np.abs(jac), 1e-8), self.step)
        g, g2, _ = self._gradient(x, f, np.maximum(1e-3 * sig_guess, 1e-8), free_all, fits_all)
        pos = g2 > 0
        sigma_int = np.where(pos, np.sqrt(2 * self.errordef / np.where(pos, g2, 1.)), sig_guess)
        cov_int, failed = self._hesse(x, f, sigma_int, free_all, fits_all)
        cov = cov_int * jac[:, :, None] * jac[:, None, :]
        errors = np.sqrt(np.clip(np.einsum('fii->fi', cov), 0., None))
        valid = (edm < self.edm_goal * 10) & ~failed & np.isfinite(f) & ~self._bad
        return FitResult(names=self.names, values=self.transform.to_external(x), errors=errors, covariance=cov,
                         fval=f, edm=edm, is_valid=valid, hesse_failed=failed, nfcn=self._nfcn.copy(),
                         n_iter=n_iter)
