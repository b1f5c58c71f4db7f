"""MIGRAD, restated: the variable-metric minimiser the reference runs through iminuit (reference vega/minimizer.py:66-97:
``iminuit.Minuit(...).migrad(ncall=100000)`` with ``errordef = 1``, limits, the configured step sizes; optional
bias-only pre-fit first).

iminuit (>= 2.0, i.e. Minuit2 with strategy 1 and tolerance 0.1) is absent from this image, so the algorithm is restated
here from its published description - the MINUIT reference manual (F. James, CERN D506) and the Minuit2 user's guide -
step by step, because the reference's fit result IS MIGRAD's stopping point, not the minimum: the pinned value
``fmin.fval = 0.6409716347033996`` (reference tests/test_vega.py:18) lies 1.1e-4 above the bounded minimum.  What is
reproduced:

* internal coordinates: sine transform for two-sided limits, square-root transforms for one-sided ones, with Minuit's
  guards next to a limit; machine precision eps = 4 * DBL_EPSILON, eps2 = 2 sqrt(eps);
* seed: step-derived first / second derivatives (g2 = 2 up / dirin^2), refined by the two-point gradient with Minuit's
  step control (strategy 1: at most 3 cycles, step tolerance 0.3, gradient tolerance 0.05), diagonal 1 / g2 as first
  metric, dcovar = 1;
* iteration: Newton step -V g, Minuit's parabolic line search (first step 1, at most 12 evaluations, 5 % tolerance, step
  growth limits 5 and 2), gradient at the new point starting from the previous steps, EDM = g^T V g / 2 with the OLD
  metric, Davidon's rank-two update (rank-one correction when delgam > gvg), dcovar = (dcovar + |update| / |V|) / 2, loop
  while EDM (1 + 3 dcovar) > 0.002 * tol * up;
* strategy 1: HESSE (diagonal second derivatives with step adaptation, refined gradient, off-diagonal elements, forced
  positive-definiteness) when dcovar > 0.05 at convergence, and another round of iterations when its EDM exceeds the goal;
* errors and covariance in external coordinates as Minuit reports them (2 up V through the transform's Jacobian; for
  limited parameters the average of the two one-sided excursions).

The function is evaluated in BATCHES: a fit is a coroutine that asks for the points of its next stage (the 2 n points of a
gradient cycle, one line-search trial, the n (n - 1) / 2 off-diagonal points of HESSE) and many fits advance in lock-step,
their requests joined into one engine call - Minuit's sequence of function VALUES per fit is unchanged, only the order in
which the engine sees the points is.  ``MIGRAD parity``: the stopping point on the reference's pinned fit, see
tests/test_minimizer_gpu.py; the trajectory itself cannot be compared (no iminuit here).
"""
import math

import numpy as np

from .minimizer import FitResult, SENTINEL

EPS = 4.0 * np.finfo(float).eps           # MnMachinePrecision::Eps
EPS2 = 2.0 * math.sqrt(EPS)               # ... Eps2


class Transform:
    """Minuit's parameter transformations, one parameter at a time (external limits lo / hi, None = unbounded)."""

    def __init__(self, limits):
        self.lo = [None if lim is None or lim[0] is None or not np.isfinite(lim[0]) else float(lim[0]) for lim in limits]
        self.hi = [None if lim is None or lim[1] is None or not np.isfinite(lim[1]) else float(lim[1]) for lim in limits]

    def has_limits(self, i):
        return self.lo[i] is not None or self.hi[i] is not None

    def ext2int(self, i, value):
        lo, hi = self.lo[i], self.hi[i]
        if lo is not None and hi is not None:
            distnn = 8. * math.sqrt(EPS2)
            yy = 2. * (value - lo) / (hi - lo) - 1.
            if yy * yy > 1. - EPS2:
                return -0.5 * math.pi + distnn if yy < 0. else 0.5 * math.pi - distnn
            return math.asin(yy)
        if lo is not None:
            yy = value - lo + 1.
            return 8. * math.sqrt(EPS2) if yy * yy < 1. + EPS2 else math.sqrt(yy * yy - 1.)
        if hi is not None:
            yy = hi - value + 1.
            return 8. * math.sqrt(EPS2) if yy * yy < 1. + EPS2 else math.sqrt(yy * yy - 1.)
        return value

    def int2ext(self, i, value):
        lo, hi = self.lo[i], self.hi[i]
        if lo is not None and hi is not None:
            return lo + 0.5 * (hi - lo) * (math.sin(value) + 1.)
        if lo is not None:
            return lo - 1. + math.sqrt(value * value + 1.)
        if hi is not None:
            return hi + 1. - math.sqrt(value * value + 1.)
        return value

    def int2ext_array(self, pts):
        """int2ext of whole rows at once ([m, n] internal -> external): the engine-facing side of a batch of trial points."""
        pts = np.asarray(pts, dtype=float)
        out = pts.copy()
        for i in range(pts.shape[1]):
            lo, hi = self.lo[i], self.hi[i]
            if lo is not None and hi is not None:
                out[:, i] = lo + 0.5 * (hi - lo) * (np.sin(pts[:, i]) + 1.)
            elif lo is not None:
                out[:, i] = lo - 1. + np.sqrt(pts[:, i] * pts[:, i] + 1.)
            elif hi is not None:
                out[:, i] = hi + 1. - np.sqrt(pts[:, i] * pts[:, i] + 1.)
        return out

    def dint2ext(self, i, value):
        lo, hi = self.lo[i], self.hi[i]
        if lo is not None and hi is not None:
            return 0.5 * abs((hi - lo) * math.cos(value))
        if lo is not None:
            return value / math.sqrt(value * value + 1.)
        if hi is not None:
            return -value / math.sqrt(value * value + 1.)
        return 1.


_TRIU = {}


def _sum_abs_packed(m):
    """Sum of |elements| of the packed upper triangle (Minuit2's sum_of_elements of a symmetric matrix)."""
    n = m.shape[0]
    if n not in _TRIU:
        _TRIU[n] = np.triu_indices(n)
    return float(np.abs(m[_TRIU[n]]).sum())


def _make_posdef(mat):
    """MnPosDef: (matrix, made_posdef)."""
    err = np.array(mat, dtype=float)
    n = err.shape[0]
    if n == 1:
        if err[0, 0] < EPS:
            return np.array([[1.0]]), True
        return err, False
    epspdf = max(1e-6, EPS2)
    dgmin = err.diagonal().min()
    dg = 0.5 + epspdf - dgmin if dgmin <= 0 else 0.
    d = err.diagonal() + dg
    d = np.where(d < 0., 1., d)
    err[np.diag_indices(n)] = d
    s = 1. / np.sqrt(d)
    p = err * s[:, None] * s[None, :]
    ev = np.linalg.eigvalsh(p)
    pmin, pmax = ev[0], max(abs(ev[-1]), 1.)
    if pmin > epspdf * pmax:
        return err, False
    padd = 0.001 * pmax - pmin
    err[np.diag_indices(n)] = err.diagonal() * (1. + padd)
    return err, True


class _Fit:
    """One MIGRAD fit as a coroutine over batches of internal points (``run`` yields [m, n] arrays, receives [m] values)."""

    # strategy 1
    GRAD_NCYCLES, GRAD_STEP_TOL, GRAD_TOL = 3, 0.3, 0.05
    HESS_NCYCLES, HESS_STEP_TOL, HESS_G2_TOL, HESS_GRAD_NCYCLES = 5, 0.3, 0.05, 2

    def __init__(self, ext_start, ext_errors, limits_free, up=1.0, tol=0.1, maxfcn=100000, seed_V=None):
        self.trafo = Transform(limits_free)
        self.n = len(ext_start)
        self.ext_errors = np.asarray(ext_errors, dtype=float)
        self.x0 = np.array([self.trafo.ext2int(i, v) for i, v in enumerate(ext_start)])
        # a state that carries a covariance (a re-run of iminuit's `iterate` loop): MnSeedGenerator then takes the internal
        # error matrix as the first metric with dcovar = 0 instead of the diagonal 1 / g2 with dcovar = 1
        self.seed_V = None if seed_V is None else np.array(seed_V, dtype=float)
        self.up = up
        self.edmval = 0.002 * max(tol * up, EPS2)
        self.maxfcn = maxfcn
        self.nfcn = 0
        self.n_iter = 0
        self.result = None

    # ---- function values through the driver
    def _eval(self, pts):
        pts = np.atleast_2d(np.asarray(pts, dtype=float))
        vals = yield pts
        self.nfcn += pts.shape[0]
        return np.asarray(vals, dtype=float)

    def to_external(self, x):
        return np.array([self.trafo.int2ext(i, v) for i, v in enumerate(x)])

    # ---- gradients
    def _initial_gradient(self, x):
        n = self.n
        grd, g2, gstep = np.zeros(n), np.zeros(n), np.zeros(n)
        for i in range(n):
            var = x[i]
            werr = self.ext_errors[i]
            sav = self.trafo.int2ext(i, var)
            sav2 = sav + werr
            if self.trafo.hi[i] is not None and sav2 > self.trafo.hi[i]:
                sav2 = self.trafo.hi[i]
            vplu = self.trafo.ext2int(i, sav2) - var
            sav2 = sav - werr
            if self.trafo.lo[i] is not None and sav2 < self.trafo.lo[i]:
                sav2 = self.trafo.lo[i]
            vmin = self.trafo.ext2int(i, sav2) - var
            gsmin = 8. * EPS2 * (abs(var) + EPS2)
            dirin = max(0.5 * (abs(vplu) + abs(vmin)), gsmin)
            g2[i] = 2.0 * self.up / (dirin * dirin)
            gstep[i] = max(gsmin, 0.1 * dirin)
            grd[i] = g2[i] * dirin
            if self.trafo.has_limits(i) and gstep[i] > 0.5:
                gstep[i] = 0.5
        return grd, g2, gstep

    def _gradient(self, x, fval, grd, g2, gstep):
        """Numerical2PGradientCalculator: the parameters are independent of each other, so cycle j of all of them is one
        batch; each keeps Minuit's own sequence of steps and stops on its own criteria."""
        n = self.n
        grd, g2, gstep = grd.copy(), g2.copy(), gstep.copy()
        dfmin = 8. * EPS2 * (abs(fval) + self.up)
        vrysml = 8. * EPS * EPS
        stepb4 = np.zeros(n)
        active = np.ones(n, dtype=bool)
        for _ in range(self.GRAD_NCYCLES):
            todo, steps = [], []
            for i in np.flatnonzero(active):
                epspri = EPS2 + abs(grd[i] * EPS2)
                optstp = math.sqrt(dfmin / (abs(g2[i]) + epspri))
                step = max(optstp, abs(0.1 * gstep[i]))
                if self.trafo.has_limits(i) and step > 0.5:
                    step = 0.5
                step = min(step, 10. * abs(gstep[i]))
                step = max(step, max(vrysml, 8. * abs(EPS2 * x[i])))
                if abs((step - stepb4[i]) / step) < self.GRAD_STEP_TOL:
                    active[i] = False
                    continue
                gstep[i] = step
                stepb4[i] = step
                todo.append(i)
                steps.append(step)
            if not todo:
                break
            pts = np.repeat(x[None, :], 2 * len(todo), axis=0)
            for q, (i, step) in enumerate(zip(todo, steps)):
                pts[2 * q, i] += step
                pts[2 * q + 1, i] -= step
            vals = yield from self._eval(pts)
            for q, (i, step) in enumerate(zip(todo, steps)):
                fs1, fs2 = vals[2 * q], vals[2 * q + 1]
                grdb4 = grd[i]
                grd[i] = 0.5 * (fs1 - fs2) / step
                g2[i] = (fs1 + fs2 - 2. * fval) / step / step
                if abs(grdb4 - grd[i]) / (abs(grd[i]) + dfmin / step) < self.GRAD_TOL:
                    active[i] = False
        return grd, g2, gstep

    # ---- line search (MnLineSearch)
    def _line_search(self, x, f0, step, gdel, lam_only=False):
        # lam_only: the coroutine asks for f(x + lam step) by yielding lam alone and is sent the value - the batch driver forms the
        # points of all its fits in one array operation (no per-request arrays)
        overal, undral, toler, slambg, alpha, maxiter = 1000., -100., 0.05, 5., 2., 12
        niter = 1
        slamin = 0.
        for i in range(self.n):
            if step[i] == 0:
                continue
            ratio = abs(x[i] / step[i])
            if slamin == 0 or ratio < slamin:
                slamin = ratio
        if abs(slamin) < EPS:
            slamin = EPS
        slamin *= EPS2

        f1 = (yield 1.) if lam_only else (yield from self._eval(x + step))[0]
        niter += 1
        fvmin, xvmin = f0, 0.
        if f1 < f0:
            fvmin, xvmin = f1, 1.
        toler8, slamax, flast, slam = toler, slambg, f1, 1.
        p0, p1 = (0., f0), (slam, flast)
        f2 = 0.
        while True:
            iterate = False
            denom = 2. * (flast - f0 - gdel * slam) / (slam * slam)
            if denom != 0:
                slam = -gdel / denom
            else:
                slam = 1.
            if slam < 0.:
                slam = slamax
            if slam > slamax:
                slam = slamax
            if slam < toler8:
                slam = toler8
            if slam < slamin:
                return xvmin, fvmin
            if abs(slam - 1.) < toler8 and p1[1] < p0[1]:
                return xvmin, fvmin
            if abs(slam - 1.) < toler8:
                slam = 1. + toler8
            f2 = (yield slam) if lam_only else (yield from self._eval(x + slam * step))[0]
            niter += 1
            if f2 < fvmin:
                fvmin, xvmin = f2, slam
            if abs(p0[1] - fvmin) < abs(fvmin) * EPS:
                iterate = True
                flast = f2
                toler8 = toler * slam
                overal = slam - toler8
                slamax = overal
                p1 = (slam, flast)
            if not (iterate and niter < maxiter):
                break
        if niter >= maxiter:
            return xvmin, fvmin
        p2 = (slam, f2)

        while True:
            slamax = max(slamax, alpha * abs(xvmin))
            # parabola through p0, p1, p2: y = a x^2 + b x + c
            (x1, y1), (x2, y2), (x3, y3) = p0, p1, p2
            dx12, dx13, dx23 = x1 - x2, x1 - x3, x2 - x3
            xm = (x1 + x2 + x3) / 3.
            dx1, dx2, dx3 = x1 - xm, x2 - xm, x3 - xm
            dx12b, dx13b, dx23b = dx1 - dx2, dx1 - dx3, dx2 - dx3
            a = y1 / (dx12b * dx13b) - y2 / (dx12b * dx23b) + y3 / (dx13b * dx23b)
            b = -y1 * (dx2 + dx3) / (dx12b * dx13b) + y2 * (dx1 + dx3) / (dx12b * dx23b) - y3 * (dx1 + dx2) / (dx13b * dx23b)
            c = y1 - a * dx1 * dx1 - b * dx1
            c += xm * (xm * a - b)
            b -= 2. * xm * a
            del dx12, dx13, dx23, c
            if a < EPS2:
                slopem = 2. * a * xvmin + b
                slam = xvmin + slamax if slopem < 0. else xvmin - slamax
            else:
                slam = -b / (2. * a)
                if slam > xvmin + slamax:
                    slam = xvmin + slamax
                if slam < xvmin - slamax:
                    slam = xvmin - slamax
            if slam > 0.:
                if slam > overal:
                    slam = overal
            elif slam < undral:
                slam = undral

            f3 = 0.
            while True:
                iterate = False
                toler9 = max(toler8, abs(toler8 * slam))
                if abs(p0[0] - slam) < toler9 or abs(p1[0] - slam) < toler9 or abs(p2[0] - slam) < toler9:
                    return xvmin, fvmin
                f3 = (yield slam) if lam_only else (yield from self._eval(x + slam * step))[0]
                if f3 > p0[1] and f3 > p1[1] and f3 > p2[1]:
                    if slam > xvmin:
                        overal = min(overal, slam - toler8)
                    if slam < xvmin:
                        undral = max(undral, slam + toler8)
                    slam = 0.5 * (slam + xvmin)
                    iterate = True
                    niter += 1
                if not (iterate and niter < maxiter):
                    break
            if niter >= maxiter:
                return xvmin, fvmin
            p3 = (slam, f3)
            if p0[1] > p1[1] and p0[1] > p2[1]:
                p0 = p3
            elif p1[1] > p0[1] and p1[1] > p2[1]:
                p1 = p3
            else:
                p2 = p3
            if f3 < fvmin:
                fvmin, xvmin = f3, slam
            else:
                if slam > xvmin:
                    overal = min(overal, slam - toler8)
                if slam < xvmin:
                    undral = max(undral, slam + toler8)
            niter += 1
            if niter >= maxiter:
                break
        return xvmin, fvmin

    # ---- HESSE (MnHesse, strategy 1)
    def _hesse(self, x, state):
        n = self.n
        amin = (yield from self._eval(x))[0]
        aimsag = math.sqrt(EPS2) * (abs(amin) + self.up)
        g2, gst, grd = state['g2'].copy(), state['gstep'].copy(), state['grd'].copy()
        dirin = gst.copy()
        yy = np.zeros(n)
        vhmat = np.zeros((n, n))
        failed = False
        # (the parameters are independent of each other - every trial point differs from x in ONE coordinate - so the steps of all
        # of them travel together: each parameter keeps Minuit's own sequence of step sizes and stops on its own criteria)
        dmin = np.array([8. * EPS2 * (abs(x[i]) + EPS2) for i in range(n)])
        d = np.maximum(np.abs(gst), dmin)
        open_ = list(range(n))              # parameters whose step adaptation goes on
        cyc = np.zeros(n, dtype=int)
        mult = np.zeros(n, dtype=int)       # widening attempts of the current cycle
        while open_ and not failed:
            pts = np.repeat(x[None, :], 2 * len(open_), axis=0)
            for q, i in enumerate(open_):
                pts[2 * q, i] = x[i] + d[i]
                pts[2 * q + 1, i] = x[i] - d[i]
            vals = yield from self._eval(pts)
            nxt = []
            for q, i in enumerate(open_):
                fs1, fs2 = vals[2 * q], vals[2 * q + 1]
                sag = 0.5 * (fs1 + fs2 - 2. * amin)
                if not sag > EPS2:
                    # flat or negative curvature at this step: widen it (at most five times per cycle)
                    mult[i] += 1
                    if self.trafo.has_limits(i):
                        if d[i] > 0.5 or mult[i] >= 5:
                            failed = True
                            break
                        d[i] *= 10.
                        if d[i] > 0.5:
                            d[i] = 0.51
                    else:
                        if mult[i] >= 5:
                            failed = True
                            break
                        d[i] *= 10.
                    nxt.append(i)
                    continue
                mult[i] = 0
                g2bfor = g2[i]
                g2[i] = 2. * sag / (d[i] * d[i])
                grd[i] = (fs1 - fs2) / (2. * d[i])
                gst[i] = d[i]
                dirin[i] = d[i]
                yy[i] = fs1
                dlast = d[i]
                dn = math.sqrt(2. * aimsag / abs(g2[i]))
                if self.trafo.has_limits(i):
                    dn = min(0.5, dn)
                if dn < dmin[i]:
                    dn = dmin[i]
                cyc[i] += 1
                done = abs((dn - dlast) / dn) < self.HESS_STEP_TOL or abs((g2[i] - g2bfor) / g2[i]) < self.HESS_G2_TOL \
                    or cyc[i] >= self.HESS_NCYCLES
                if done:
                    vhmat[i, i] = g2[i]
                    continue
                dn = min(dn, 10. * dlast)
                dn = max(dn, 0.1 * dlast)
                d[i] = dn
                nxt.append(i)
            open_ = nxt
        if failed:
            return dict(state, hesse_failed=True, fval=amin)
        # refine the first derivatives (HessianGradientCalculator)
        dfmin = 4. * EPS2 * (abs(amin) + self.up)
        for i in range(n):
            xtf = x[i]
            dmin = 4. * EPS2 * (xtf + EPS2)
            epspri = EPS2 + abs(grd[i] * EPS2)
            optstp = math.sqrt(dfmin / (abs(g2[i]) + epspri))
            d = 0.2 * abs(gst[i])
            if d > optstp:
                d = optstp
            if d < dmin:
                d = dmin
            chgold = 10000.
            for j in range(self.HESS_GRAD_NCYCLES):
                pts = np.repeat(x[None, :], 2, axis=0)
                pts[0, i] = xtf + d
                pts[1, i] = xtf - d
                fs1, fs2 = (yield from self._eval(pts))
                grdold = grd[i]
                grdnew = (fs1 - fs2) / (2. * d)
                dgmin = EPS * (abs(fs1) + abs(fs2)) / d
                if abs(grdnew) < EPS:
                    break
                change = abs((grdold - grdnew) / grdnew)
                if change > chgold and j > 1:
                    break
                chgold = change
                grd[i] = grdnew
                gst[i] = d
                if change < 0.05:
                    break
                if abs(grdold - grdnew) < dgmin:
                    break
                if d < dmin:
                    break
                d *= 0.2
        # off-diagonal elements: one batch
        pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
        if pairs:
            pts = np.repeat(x[None, :], len(pairs), axis=0)
            for q, (i, j) in enumerate(pairs):
                pts[q, i] += dirin[i]
                pts[q, j] += dirin[j]
            vals = yield from self._eval(pts)
            for q, (i, j) in enumerate(pairs):
                vhmat[i, j] = vhmat[j, i] = (vals[q] + amin - yy[i] - yy[j]) / (dirin[i] * dirin[j])
        hmat, made = _make_posdef(vhmat)
        try:
            V = np.linalg.inv(hmat)
        except np.linalg.LinAlgError:
            return dict(state, hesse_failed=True, fval=amin)
        edm = 0.5 * float(grd @ V @ grd)
        return dict(x=x.copy(), fval=amin, grd=grd, g2=g2, gstep=gst, V=V, dcovar=0., edm=edm, made_posdef=made,
                    hesse_failed=False, accurate=not made)

    # ---- the whole fit
    def run(self, run_hesse_at_end=True):
        n = self.n
        x = self.x0.copy()
        fval = (yield from self._eval(x))[0]
        if not np.isfinite(fval):
            self.result = dict(x=x, fval=np.inf, edm=np.inf, V=np.zeros((n, n)), valid=False, hesse_failed=True,
                               accurate=False, above_max_edm=True, limit=False)
            return
        grd, g2, gstep = self._initial_gradient(x)
        grd, g2, gstep = yield from self._gradient(x, fval, grd, g2, gstep)
        V = np.diag([1. / g2[i] if abs(g2[i]) > EPS2 else 1. for i in range(n)])
        dcovar0 = 1.
        if self.seed_V is not None:
            V, dcovar0 = self.seed_V.copy(), 0.
        state = dict(x=x, fval=fval, grd=grd, g2=g2, gstep=gstep, V=V, dcovar=dcovar0, edm=0.5 * float(grd @ V @ grd),
                     made_posdef=False, hesse_failed=False, accurate=False)
        if (g2 <= 0).any():
            state = yield from self._negative_g2(state)
        maxfcn_eff = self.maxfcn
        ipass = 0
        reached_limit = False
        while True:
            iterate = False
            state, reached_limit = yield from self._iterate(state, maxfcn_eff)
            if reached_limit:
                break
            edm = state['edm']
            if state['dcovar'] > 0.05:
                st = yield from self._hesse(state['x'], state)
                if not st.get('hesse_failed'):
                    state = st
                    edm = st['edm']
                    if edm > self.edmval and edm >= abs(EPS2 * state['fval']):
                        iterate = True
                else:
                    state = st
                    break
            if ipass == 0:
                maxfcn_eff = int(self.maxfcn * 1.3)
            ipass += 1
            if not iterate:
                break
        hesse_failed = bool(state.get('hesse_failed'))
        self.result = dict(x=state['x'], fval=state['fval'], edm=state['edm'], V=state['V'],
                           valid=(not reached_limit) and state['edm'] <= 10 * self.edmval and not hesse_failed
                           and np.isfinite(state['fval']),
                           hesse_failed=hesse_failed, accurate=bool(state.get('accurate')),
                           above_max_edm=state['edm'] > 10 * self.edmval, limit=bool(reached_limit))

    def _negative_g2(self, state):
        """NegativeG2LineSearch: walk along every direction with a negative second derivative until it turns positive."""
        x, fval, grd, g2, gstep = state['x'].copy(), state['fval'], state['grd'].copy(), state['g2'].copy(), state['gstep'].copy()
        n = self.n
        for _ in range(2 * n):
            i_neg = [i for i in range(n) if g2[i] <= 0]
            if not i_neg:
                break
            for i in i_neg:
                if abs(gstep[i]) < EPS2:
                    continue
                step = np.zeros(n)
                step[i] = gstep[i] * (-1. if grd[i] > 0 else 1.)
                gdel = step[i] * grd[i]
                lam, fmin = yield from self._line_search(x, fval, step, gdel)
                x = x + lam * step
                fval = fmin
                grd, g2, gstep = yield from self._gradient(x, fval, grd, g2, gstep)
                break
        V = np.diag([1. / g2[i] if abs(g2[i]) > EPS2 else 1. for i in range(n)])
        return dict(state, x=x, fval=fval, grd=grd, g2=g2, gstep=gstep, V=V, dcovar=1., edm=0.5 * float(grd @ V @ grd))

    def _iterate(self, s0, maxfcn):
        edm = s0['edm']
        while True:
            step = -s0['V'] @ s0['grd']
            gdel = float(step @ s0['grd'])
            if gdel > 0.:
                V, _ = _make_posdef(s0['V'])
                s0 = dict(s0, V=V)
                step = -V @ s0['grd']
                gdel = float(step @ s0['grd'])
                if gdel > 0.:
                    return s0, False
            lam, fnew = yield from self._line_search(s0['x'], s0['fval'], step, gdel)
            if abs(fnew - s0['fval']) <= abs(s0['fval']) * EPS:
                break           # no improvement
            x = s0['x'] + lam * step
            grd, g2, gstep = yield from self._gradient(x, fnew, s0['grd'], s0['g2'], s0['gstep'])
            edm = 0.5 * float(grd @ s0['V'] @ grd)
            if edm != edm:
                return s0, False
            if edm < 0.:
                V, _ = _make_posdef(s0['V'])
                s0 = dict(s0, V=V)
                edm = 0.5 * float(grd @ V @ grd)
                if edm < 0.:
                    return s0, False
            # Davidon's update
            v0 = s0['V']
            dx = x - s0['x']
            dg = grd - s0['grd']
            delgam = float(dx @ dg)
            gvg = float(dg @ v0 @ dg)
            if delgam == 0. or gvg <= 0.:
                V, dcov = v0, s0['dcovar']
            else:
                vg = v0 @ dg
                upd = np.outer(dx, dx) / delgam - np.outer(vg, vg) / gvg
                if delgam > gvg:
                    w = dx / delgam - vg / gvg
                    upd += gvg * np.outer(w, w)
                sum_upd = _sum_abs_packed(upd)
                V = v0 + upd
                dcov = 0.5 * (s0['dcovar'] + sum_upd / _sum_abs_packed(V))
            s0 = dict(x=x, fval=fnew, grd=grd, g2=g2, gstep=gstep, V=V, dcovar=dcov, edm=edm, made_posdef=False,
                      hesse_failed=False, accurate=False)
            self.n_iter += 1
            edm_test = edm * (1. + 3. * dcov)
            if not (edm_test > self.edmval and self.nfcn < maxfcn):
                break
        return s0, self.nfcn >= maxfcn

    # ---- results in external coordinates (MnUserParameterState / MnUserCovariance)
    def external(self):
        r = self.result
        x = r['x']
        n = self.n
        values = self.to_external(x)
        cov_int = 2. * self.up * r['V']
        jac = np.array([self.trafo.dint2ext(i, x[i]) for i in range(n)])
        cov = cov_int * jac[:, None] * jac[None, :]
        errors = np.zeros(n)
        for i in range(n):
            dx = math.sqrt(max(cov_int[i, i], 0.))
            if self.trafo.has_limits(i):
                ui = values[i]
                du1 = self.trafo.int2ext(i, x[i] + dx) - ui
                du2 = self.trafo.int2ext(i, x[i] - dx) - ui
                if dx > 1. and self.trafo.lo[i] is not None and self.trafo.hi[i] is not None:
                    du1 = self.trafo.hi[i] - self.trafo.lo[i]
                errors[i] = 0.5 * (abs(du1) + abs(du2))
            else:
                errors[i] = dx
        return values, errors, cov


def _drive(fits, evaluate_internal):
    """Advance the coroutines in lock-step: one joined evaluation per round."""
    gens = [(k, f.run()) for k, f in enumerate(fits)]
    pending = {}
    for k, g in gens:
        try:
            pending[k] = (g, next(g))
        except StopIteration:
            pass
    while pending:
        keys = list(pending)
        pts = np.concatenate([pending[k][1] for k in keys])
        owner = np.concatenate([np.full(pending[k][1].shape[0], k) for k in keys])
        vals = evaluate_internal(pts, owner)
        nxt = {}
        off = 0
        for k in keys:
            g, req = pending[k]
            m = req.shape[0]
            try:
                nxt[k] = (g, g.send(vals[off:off + m]))
            except StopIteration:
                pass
            off += m
        pending = nxt


class MigradMinimizer:
    """Drop-in for :class:`vega_amd.minimizer.BatchedMinimizer` with MIGRAD's own sequence of steps per fit.

    ``evaluate(theta_ext [m, P], fit_index [m]) -> chi2 [m]``."""

    def __init__(self, evaluate, names, start, errors, limits, tol=0.1, errordef=1.0, maxfcn=100000, vectorised=True,
                 iterate=5, machine=None):
        # machine: the fits run where the walkers live - `machine(plan, ext0 [F, P], fit_ids [F]) -> per-stage results` advances
        # every fit with the state machine of vega_amd/csrc/vmx_migrad.h (the engine's `fit_migrad`: kernels, one thread per fit;
        # tests/test_migrad_machine.py: the same header on the CPU); `evaluate` is then not called at all.  None: the NumPy
        # drivers below.
        self.machine = machine
        # iminuit's Minuit.migrad(ncall, iterate=5) - what the reference calls (vega/minimizer.py:79, :97) - runs MnMigrad again,
        # up to `iterate` times in all, while the minimum is neither valid nor at the call limit; every re-run starts from the
        # previous run's state: its values, its parameter errors as step sizes and its error matrix as the first metric
        self.iterate = int(iterate)
        self.evaluate = evaluate
        self.vectorised = vectorised        # False: one coroutine per fit for every stage (`_Fit.run`, the readable reference)
        self.names = list(names)
        self.start = np.asarray(start, dtype=float)
        self.step = np.asarray(errors, dtype=float)
        self.limits = [tuple(lim) if lim is not None else (None, None) for lim in limits]
        self.tol, self.errordef, self.maxfcn = tol, errordef, maxfcn

    def _stage(self, ext0, free, fit_ids):
        """One Minuit object per fit over the parameters ``free`` (the others held at ext0): dict of per-fit arrays."""
        free = np.asarray(free, dtype=int)
        F = ext0.shape[0]
        limits = [self.limits[j] for j in free]
        trafo = _VecTransform(limits)

        def evaluate_internal(pts, owner):
            theta = ext0[owner]
            theta[:, free] = trafo.int2ext(np.atleast_2d(pts))
            vals = np.asarray(self.evaluate(theta, fit_ids[owner]), dtype=float)
            return np.where(np.isfinite(vals) & (vals < SENTINEL), vals, np.inf)

        if self.vectorised:
            batch = _Batch(ext0[:, free], self.step[free], limits, evaluate_internal, up=self.errordef, tol=self.tol,
                           maxfcn=self.maxfcn)
            S = batch.run()
            x, V, fval, edm = S['x'], S['V'], S['fval'], S['edm']
            hesse_failed, accurate = S['hesse_failed'].copy(), S['accurate'].copy()
            valid = batch.alive & ~S['limit'] & (edm <= 10 * batch.edmval) & ~hesse_failed & np.isfinite(fval)
            nfcn, n_iter = batch.nfcn.copy(), batch.n_iter.copy()
            at_limit = S['limit'].copy()
            redo = batch.slow
        else:
            x = np.zeros((F, free.size)); V = np.zeros((F, free.size, free.size)); fval = np.full(F, np.inf); edm = np.full(F, np.inf)
            hesse_failed = np.zeros(F, dtype=bool); accurate = np.zeros(F, dtype=bool); valid = np.zeros(F, dtype=bool)
            nfcn = np.zeros(F, dtype=np.int64); n_iter = np.zeros(F, dtype=int)
            at_limit = np.zeros(F, dtype=bool)
            redo = list(range(F))
        if redo:
            # the reference implementation, one coroutine per fit: everything when not vectorised, else the fits that left the
            # common path (a negative second derivative at the seed, a metric that lost positive-definiteness)
            redo = np.array(redo, dtype=int)
            fits = [_Fit(ext0[f][free], self.step[free], limits, up=self.errordef, tol=self.tol, maxfcn=self.maxfcn) for f in redo]
            _drive(fits, lambda pts, owner: evaluate_internal(pts, redo[owner]))
            for f, fit in zip(redo, fits):
                r = fit.result
                nfcn[f] = fit.nfcn          # (Minuit's count of this fit; what the common path spent on it before is not part of it)
                n_iter[f] = fit.n_iter
                if r is None or not np.isfinite(r['fval']):
                    fval[f], edm[f], valid[f], hesse_failed[f] = np.inf, np.inf, False, True
                    continue
                x[f], V[f], fval[f], edm[f] = r['x'], r['V'], r['fval'], r['edm']
                valid[f], hesse_failed[f], accurate[f] = r['valid'], r['hesse_failed'], r['accurate']
                at_limit[f] = r['limit']
        # iminuit's `iterate` loop: the fits that ended neither valid nor at the call limit run again from their last state
        # (values, parameter errors as steps, error matrix as first metric), up to `iterate` runs in all.  Rare (hard mocks):
        # the readable one-coroutine-per-fit implementation serves them.
        for _ in range(max(self.iterate - 1, 0)):
            again = np.flatnonzero(~valid & ~at_limit & np.isfinite(fval))
            if not again.size:
                break
            vt = Transform(limits)
            fits = []
            for f in again:
                ext_vals = np.array([vt.int2ext(i, x[f, i]) for i in range(free.size)])
                dxs_f = np.sqrt(np.clip(2. * self.errordef * np.diag(V[f]), 0., None))
                ext_err = np.array([dxs_f[i] * abs(vt.dint2ext(i, x[f, i])) if vt.has_limits(i) else dxs_f[i] for i in range(free.size)])
                ext_err = np.where(ext_err > 0, ext_err, self.step[free])
                fits.append(_Fit(ext_vals, ext_err, limits, up=self.errordef, tol=self.tol, maxfcn=self.maxfcn, seed_V=V[f]))
            _drive(fits, lambda pts, owner: evaluate_internal(pts, again[owner]))
            for f, fit in zip(again, fits):
                r = fit.result
                nfcn[f] += fit.nfcn
                n_iter[f] += fit.n_iter
                if r is None or not np.isfinite(r['fval']):
                    at_limit[f] = True      # (nothing to restart from: keep the previous state, stop iterating)
                    continue
                x[f], V[f], fval[f], edm[f] = r['x'], r['V'], r['fval'], r['edm']
                valid[f], hesse_failed[f], accurate[f] = r['valid'], r['hesse_failed'], r['accurate']
                at_limit[f] = r['limit']
        values, errors, cov = _external_results(trafo, x, V, self.errordef)
        return dict(values=values, errors=errors, cov=cov, fval=fval, edm=edm, valid=valid, hesse_failed=hesse_failed,
                    accurate=accurate, nfcn=nfcn, n_iter=n_iter)

    def plan(self, free_sets):
        """The Minuit objects of a fit as the state machine takes them (vega_amd/csrc/vmx_migrad.h `Spec`): per stage the free
        parameters' indices, limits and step sizes; errordef, tolerance, call limit, iminuit's `iterate`."""
        stages = []
        for free in free_sets:
            free = np.asarray(free, dtype=int)
            stages.append(dict(free=free, limits=[self.limits[j] for j in free], errors=self.step[free].copy()))
        return dict(stages=stages, n_params=len(self.names), up=self.errordef, tol=self.tol, maxfcn=self.maxfcn,
                    iterate=max(self.iterate, 1))

    def _minimize_on_machine(self, ext, free_sets, fit_ids):
        """The fits advanced by the state machine: results of the last Minuit object in `_stage`'s form, call / iteration counts
        summed over the objects."""
        plan = self.plan(free_sets)
        outs = self.machine(plan, ext, fit_ids)
        nfcn = sum(np.asarray(o['nfcn'], dtype=np.int64) for o in outs)
        n_iter = sum(np.asarray(o['n_iter'], dtype=int) for o in outs)
        for stage, o in zip(plan['stages'][:-1], outs[:-1]):
            ok = np.isfinite(o['fval'])
            ext[np.ix_(ok, stage['free'])] = o['ext'][ok]          # (what the machine did before it started the next object)
        last, o = plan['stages'][-1], outs[-1]
        trafo = _VecTransform(last['limits'])
        flags = np.asarray(o['flags'])
        values, errors, cov = _external_results(trafo, o['x'], o['V'], self.errordef)
        values = np.asarray(o['ext'], dtype=float)                 # (the machine's own transform: what it would start from)
        res = dict(values=values, errors=errors, cov=cov, fval=np.asarray(o['fval'], dtype=float), edm=np.asarray(o['edm'], dtype=float),
                   valid=(flags & 1) != 0, hesse_failed=(flags & 2) != 0, accurate=(flags & 4) != 0, nfcn=nfcn, n_iter=n_iter)
        return res

    def minimize(self, n_fits=1, start=None, fixed=(), prefit_bias=True):
        P = len(self.names)
        F = int(n_fits)
        ext = np.tile(self.start, (F, 1)) if start is None else np.array(start, dtype=float).reshape(F, P)
        fit_ids = np.arange(F)
        free_all = np.array([j for j, n in enumerate(self.names) if n not in fixed], dtype=int)
        nfcn = np.zeros(F, dtype=np.int64)
        n_iter = np.zeros(F, dtype=int)
        # the reference first minimises over the bias parameters alone, a Minuit object of its own (vega/minimizer.py:66-86)
        bias = np.array([j for j in free_all if 'bias' in self.names[j]], dtype=int)
        if self.machine is not None and free_all.size > 0:
            free_sets = ([bias] if prefit_bias and bias.size > 0 else []) + [free_all]
            res = self._minimize_on_machine(ext, free_sets, fit_ids)
            nfcn += res['nfcn']
            n_iter += res['n_iter']
        else:
            if prefit_bias and bias.size > 0:
                pre = self._stage(ext, bias, fit_ids)
                nfcn += pre['nfcn']
                n_iter += pre['n_iter']
                ok = np.isfinite(pre['fval'])
                ext[np.ix_(ok, bias)] = pre['values'][ok]
            res = self._stage(ext, free_all, fit_ids)
            nfcn += res['nfcn']
            n_iter += res['n_iter']
        ok = np.isfinite(res['fval'])
        values = ext.copy()
        errors = np.zeros((F, P))
        cov = np.zeros((F, P, P))
        values[np.ix_(ok, free_all)] = res['values'][ok]
        errors[np.ix_(ok, free_all)] = res['errors'][ok]
        cov[np.ix_(np.flatnonzero(ok), free_all, free_all)] = res['cov'][ok]
        out = FitResult(names=self.names, values=values, errors=errors, covariance=cov, fval=np.where(ok, res['fval'], np.inf),
                        edm=np.where(ok, res['edm'], np.inf), is_valid=res['valid'] & ok, hesse_failed=res['hesse_failed'] | ~ok,
                        nfcn=nfcn, n_iter=n_iter)
        out.has_accurate_covar = res['accurate'] & ok
        return out

def _external_results(trafo, x, V, errordef):
    """External values, errors and covariance of fits at internal points x [F, n] with error matrices V [F, n, n]
    (MnUserParameterState / MnUserCovariance): 2 up V through the transform's Jacobian; for limited parameters the average of
    the two one-sided excursions."""
    x = np.asarray(x, dtype=float)
    V = np.asarray(V, dtype=float)
    values = trafo.int2ext(x)
    cov_int = 2. * errordef * V
    jac = trafo.dint2ext(x)
    cov = cov_int * jac[:, :, None] * jac[:, None, :]
    dxs = np.sqrt(np.clip(np.einsum('fii->fi', cov_int), 0., None))
    errors = dxs.copy()
    for i in range(trafo.n):
        if trafo.has_limits[i]:
            du1 = trafo.int2ext_col(i, x[:, i] + dxs[:, i]) - values[:, i]
            du2 = trafo.int2ext_col(i, x[:, i] - dxs[:, i]) - values[:, i]
            if trafo.lo[i] is not None and trafo.hi[i] is not None:
                du1 = np.where(dxs[:, i] > 1., trafo.hi[i] - trafo.lo[i], du1)
            errors[:, i] = 0.5 * (np.abs(du1) + np.abs(du2))
    return values, errors, cov


# ---------------------------------------------------------------------------------------------------------------------
# The same algorithm, array-oriented: every stage but the line search runs for all fits at once (NumPy over fits), the
# line searches are per-fit coroutines (MnLineSearch is a small state machine of scalar decisions) advanced in rounds.
# A fit's sequence of function values is that of `_Fit.run`; what changes is who does the bookkeeping.  `_Fit.run` stays
# as the readable reference (tests/test_migrad.py compares the two) and takes over the rare branches (a negative second
# derivative at the seed, a metric that lost positive-definiteness).
# ---------------------------------------------------------------------------------------------------------------------
class _VecTransform:
    """Minuit's transformations for arrays [F, n] (column i has the limits of parameter i)."""

    def __init__(self, limits):
        t = Transform(limits)
        self.lo, self.hi = t.lo, t.hi
        self.n = len(limits)
        self.has_limits = np.array([t.has_limits(i) for i in range(self.n)])

    def ext2int_col(self, i, v):
        lo, hi = self.lo[i], self.hi[i]
        d = 8. * math.sqrt(EPS2)
        if lo is not None and hi is not None:
            yy = 2. * (v - lo) / (hi - lo) - 1.
            edge = yy * yy > 1. - EPS2
            return np.where(edge, np.where(yy < 0., -0.5 * math.pi + d, 0.5 * math.pi - d), np.arcsin(np.clip(yy, -1., 1.)))
        if lo is not None:
            yy = v - lo + 1.
            return np.where(yy * yy < 1. + EPS2, d, np.sqrt(np.maximum(yy * yy - 1., 0.)))
        if hi is not None:
            yy = hi - v + 1.
            return np.where(yy * yy < 1. + EPS2, d, np.sqrt(np.maximum(yy * yy - 1., 0.)))
        return np.array(v, dtype=float)

    def int2ext_col(self, i, v):
        lo, hi = self.lo[i], self.hi[i]
        if lo is not None and hi is not None:
            return lo + 0.5 * (hi - lo) * (np.sin(v) + 1.)
        if lo is not None:
            return lo - 1. + np.sqrt(v * v + 1.)
        if hi is not None:
            return hi + 1. - np.sqrt(v * v + 1.)
        return np.array(v, dtype=float)

    def int2ext(self, x):
        return np.stack([self.int2ext_col(i, x[:, i]) for i in range(self.n)], axis=1) if self.n else x.copy()

    def dint2ext(self, x):
        out = np.ones_like(x)
        for i in range(self.n):
            lo, hi = self.lo[i], self.hi[i]
            v = x[:, i]
            if lo is not None and hi is not None:
                out[:, i] = 0.5 * np.abs((hi - lo) * np.cos(v))
            elif lo is not None:
                out[:, i] = v / np.sqrt(v * v + 1.)
            elif hi is not None:
                out[:, i] = -v / np.sqrt(v * v + 1.)
        return out


def _triu_abs_sum(m):
    n = m.shape[-1]
    if n not in _TRIU:
        _TRIU[n] = np.triu_indices(n)
    iu = _TRIU[n]
    return np.abs(m[:, iu[0], iu[1]]).sum(axis=1)


class _Batch:
    """F fits over the same n free parameters; ``evaluate(pts [m, n] internal, owner [m]) -> values [m]``."""

    G_NC, G_STOL, G_TOL = _Fit.GRAD_NCYCLES, _Fit.GRAD_STEP_TOL, _Fit.GRAD_TOL
    H_NC, H_STOL, H_G2TOL, HG_NC = _Fit.HESS_NCYCLES, _Fit.HESS_STEP_TOL, _Fit.HESS_G2_TOL, _Fit.HESS_GRAD_NCYCLES

    def __init__(self, ext0, ext_errors, limits, evaluate, up=1.0, tol=0.1, maxfcn=100000):
        self.F, self.n = ext0.shape
        self.T = _VecTransform(limits)
        self.limits = limits
        self.err = np.asarray(ext_errors, dtype=float)
        self.evaluate = evaluate
        self.up, self.tol, self.maxfcn = up, tol, maxfcn
        self.edmval = 0.002 * max(tol * up, EPS2)
        self.nfcn = np.zeros(self.F, dtype=np.int64)
        self.n_iter = np.zeros(self.F, dtype=int)
        self.x0 = np.stack([self.T.ext2int_col(i, ext0[:, i]) for i in range(self.n)], axis=1)
        self.ext0, self.ext_errors = ext0, ext_errors

    def _f(self, pts, owner):
        vals = np.asarray(self.evaluate(pts, owner), dtype=float)
        np.add.at(self.nfcn, owner, 1)
        return vals

    # ---- seed
    def _initial_gradient(self, x):
        F, n = x.shape
        grd, g2, gstep = np.zeros((F, n)), np.zeros((F, n)), np.zeros((F, n))
        for i in range(n):
            var = x[:, i]
            werr = self.err[i]
            sav = self.T.int2ext_col(i, var)
            sav2 = sav + werr
            if self.T.hi[i] is not None:
                sav2 = np.where(sav2 > self.T.hi[i], self.T.hi[i], sav2)
            vplu = self.T.ext2int_col(i, sav2) - var
            sav2 = sav - werr
            if self.T.lo[i] is not None:
                sav2 = np.where(sav2 < self.T.lo[i], self.T.lo[i], sav2)
            vmin = self.T.ext2int_col(i, sav2) - var
            gsmin = 8. * EPS2 * (np.abs(var) + EPS2)
            dirin = np.maximum(0.5 * (np.abs(vplu) + np.abs(vmin)), gsmin)
            g2[:, i] = 2.0 * self.up / (dirin * dirin)
            gs = np.maximum(gsmin, 0.1 * dirin)
            if self.T.has_limits[i]:
                gs = np.minimum(gs, 0.5)
            gstep[:, i] = gs
            grd[:, i] = g2[:, i] * dirin
        return grd, g2, gstep

    # ---- two-point gradient of the fits `idx` at x (rows aligned with idx), starting from (grd, g2, gstep)
    def _gradient(self, idx, x, fval, grd, g2, gstep):
        m, n = x.shape
        grd, g2, gstep = grd.copy(), g2.copy(), gstep.copy()
        dfmin = 8. * EPS2 * (np.abs(fval) + self.up)
        vrysml = 8. * EPS * EPS
        stepb4 = np.zeros((m, n))
        active = np.ones((m, n), dtype=bool)
        lim = self.T.has_limits[None, :]
        for _ in range(self.G_NC):
            epspri = EPS2 + np.abs(grd * EPS2)
            optstp = np.sqrt(dfmin[:, None] / (np.abs(g2) + epspri))
            step = np.maximum(optstp, np.abs(0.1 * gstep))
            step = np.where(lim & (step > 0.5), 0.5, step)
            step = np.minimum(step, 10. * np.abs(gstep))
            step = np.maximum(step, np.maximum(vrysml, 8. * np.abs(EPS2 * x)))
            with np.errstate(invalid='ignore', divide='ignore'):
                conv = np.abs((step - stepb4) / step) < self.G_STOL
            active &= ~conv
            if not active.any():
                break
            r, c = np.nonzero(active)
            gstep[r, c] = step[r, c]
            stepb4[r, c] = step[r, c]
            pts = np.repeat(x[r], 2, axis=0)
            k = np.arange(r.size)
            pts[2 * k, c] += step[r, c]
            pts[2 * k + 1, c] -= step[r, c]
            vals = self._f(pts, np.repeat(idx[r], 2))
            fs1, fs2 = vals[0::2], vals[1::2]
            st = step[r, c]
            grdb4 = grd[r, c]
            gnew = 0.5 * (fs1 - fs2) / st
            grd[r, c] = gnew
            g2[r, c] = (fs1 + fs2 - 2. * fval[r]) / st / st
            with np.errstate(invalid='ignore', divide='ignore'):
                done = np.abs(grdb4 - gnew) / (np.abs(gnew) + dfmin[r] / st) < self.G_TOL
            active[r[done], c[done]] = False
        return grd, g2, gstep

    # ---- line searches of the fits idx: per-fit coroutines, joined evaluations
    def _line_searches(self, idx, x, f0, step, gdel):
        ls = self._ls_helper
        m = idx.size
        lam, fnew = np.zeros(m), np.array(f0, dtype=float)
        gens, want = [None] * m, np.zeros(m)
        live = []
        for q in range(m):
            g = ls._line_search(x[q], float(f0[q]), step[q], float(gdel[q]), lam_only=True)
            try:
                want[q] = next(g)
                gens[q] = g
                live.append(q)
            except StopIteration as stop:
                lam[q], fnew[q] = stop.value
        while live:
            keys = np.array(live, dtype=int)
            vals = self._f(x[keys] + want[keys, None] * step[keys], idx[keys]).tolist()
            nxt = []
            for q, v in zip(live, vals):
                try:
                    want[q] = gens[q].send(v)
                    nxt.append(q)
                except StopIteration as stop:
                    lam[q], fnew[q] = stop.value
            live = nxt
        return lam, fnew

    # ---- the iterations (VariableMetricBuilder) for the fits `act`, until each converges
    def _iterate(self, act, S, maxfcn):
        ls = _Fit(self.ext0[0], self.ext_errors, self.limits, up=self.up, tol=self.tol)
        self._ls_helper = ls
        act = np.array(act, dtype=int)
        slow = []
        while act.size:
            V, g = S['V'][act], S['grd'][act]
            step = -np.einsum('fij,fj->fi', V, g)
            gdel = np.einsum('fi,fi->f', step, g)
            bad = gdel > 0.
            if bad.any():           # a metric that is not positive definite: the reference implementation takes the fit over
                slow += list(act[bad])
                act, step, gdel = act[~bad], step[~bad], gdel[~bad]
                if not act.size:
                    break
            x0, f0 = S['x'][act], S['fval'][act]
            lam, fnew = self._line_searches(act, x0, f0, step, gdel)
            improved = ~(np.abs(fnew - f0) <= np.abs(f0) * EPS)
            act_i = act[improved]
            if act_i.size:
                x1 = x0[improved] + lam[improved, None] * step[improved]
                f1 = fnew[improved]
                g1, g21, gs1 = self._gradient(act_i, x1, f1, S['grd'][act_i], S['g2'][act_i], S['gstep'][act_i])
                V0 = S['V'][act_i]
                edm = 0.5 * np.einsum('fi,fij,fj->f', g1, V0, g1)
                weird = ~(edm >= 0.)            # NaN or negative: the reference implementation's business
                dx = x1 - S['x'][act_i]
                dg = g1 - S['grd'][act_i]
                delgam = np.einsum('fi,fi->f', dx, dg)
                vg = np.einsum('fij,fj->fi', V0, dg)
                gvg = np.einsum('fi,fi->f', dg, vg)
                ok = ~((delgam == 0.) | (gvg <= 0.))
                with np.errstate(invalid='ignore', divide='ignore'):
                    upd = dx[:, :, None] * dx[:, None, :] / delgam[:, None, None] - vg[:, :, None] * vg[:, None, :] / gvg[:, None, None]
                    w = dx / delgam[:, None] - vg / gvg[:, None]
                    upd = np.where((delgam > gvg)[:, None, None], upd + gvg[:, None, None] * (w[:, :, None] * w[:, None, :]), upd)
                    V1 = V0 + upd
                    dcov = 0.5 * (S['dcovar'][act_i] + _triu_abs_sum(upd) / _triu_abs_sum(V1))
                V1 = np.where(ok[:, None, None], V1, V0)
                dcov = np.where(ok, dcov, S['dcovar'][act_i])
                S['x'][act_i], S['fval'][act_i] = x1, f1
                S['grd'][act_i], S['g2'][act_i], S['gstep'][act_i] = g1, g21, gs1
                S['V'][act_i], S['dcovar'][act_i], S['edm'][act_i] = V1, dcov, edm
                self.n_iter[act_i] += 1
                if weird.any():
                    slow += list(act_i[weird])
                go_on = (edm * (1. + 3. * dcov) > self.edmval) & (self.nfcn[act_i] < maxfcn) & ~weird
                S['limit'][act_i] = self.nfcn[act_i] >= maxfcn
                act = act_i[go_on]
            else:
                act = act_i
        return slow

    # ---- HESSE for the fits idx (state S is updated in place); returns the mask of failures
    def _hesse(self, idx, S):
        idx = np.array(idx, dtype=int)
        m, n = idx.size, self.n
        x = S['x'][idx].copy()
        amin = self._f(x, idx)
        aimsag = math.sqrt(EPS2) * (np.abs(amin) + self.up)
        g2, gst, grd = S['g2'][idx].copy(), S['gstep'][idx].copy(), S['grd'][idx].copy()
        dirin, yy = gst.copy(), np.zeros((m, n))
        vh = np.zeros((m, n, n))
        failed = np.zeros(m, dtype=bool)
        lim = self.T.has_limits
        dmin = 8. * EPS2 * (np.abs(x) + EPS2)
        d = np.maximum(np.abs(gst), dmin)
        open_ = np.ones((m, n), dtype=bool)
        cyc, mult = np.zeros((m, n), dtype=int), np.zeros((m, n), dtype=int)
        while open_.any():
            open_ &= ~failed[:, None]
            r, c = np.nonzero(open_)
            if not r.size:
                break
            k = np.arange(r.size)
            pts = np.repeat(x[r], 2, axis=0)
            pts[2 * k, c] = x[r, c] + d[r, c]
            pts[2 * k + 1, c] = x[r, c] - d[r, c]
            vals = self._f(pts, np.repeat(idx[r], 2))
            fs1, fs2 = vals[0::2], vals[1::2]
            for q in range(r.size):             # (scalar decisions per parameter, as MnHesse takes them)
                f_, i = r[q], c[q]
                if failed[f_]:
                    continue
                sag = 0.5 * (fs1[q] + fs2[q] - 2. * amin[f_])
                if not sag > EPS2:
                    mult[f_, i] += 1
                    if lim[i]:
                        if d[f_, i] > 0.5 or mult[f_, i] >= 5:
                            failed[f_] = True
                            continue
                        d[f_, i] *= 10.
                        if d[f_, i] > 0.5:
                            d[f_, i] = 0.51
                    else:
                        if mult[f_, i] >= 5:
                            failed[f_] = True
                            continue
                        d[f_, i] *= 10.
                    continue
                mult[f_, i] = 0
                g2bfor = g2[f_, i]
                di = d[f_, i]
                g2[f_, i] = 2. * sag / (di * di)
                grd[f_, i] = (fs1[q] - fs2[q]) / (2. * di)
                gst[f_, i] = di
                dirin[f_, i] = di
                yy[f_, i] = fs1[q]
                dn = math.sqrt(2. * aimsag[f_] / abs(g2[f_, i]))
                if lim[i]:
                    dn = min(0.5, dn)
                if dn < dmin[f_, i]:
                    dn = dmin[f_, i]
                cyc[f_, i] += 1
                if abs((dn - di) / dn) < self.H_STOL or abs((g2[f_, i] - g2bfor) / g2[f_, i]) < self.H_G2TOL \
                        or cyc[f_, i] >= self.H_NC:
                    vh[f_, i, i] = g2[f_, i]
                    open_[f_, i] = False
                    continue
                dn = min(dn, 10. * di)
                dn = max(dn, 0.1 * di)
                d[f_, i] = dn
        good = ~failed
        # refined first derivatives (HessianGradientCalculator), all parameters of all fits together
        dfmin = 4. * EPS2 * (np.abs(amin) + self.up)
        dminh = 4. * EPS2 * (x + EPS2)
        epspri = EPS2 + np.abs(grd * EPS2)
        optstp = np.sqrt(dfmin[:, None] / (np.abs(g2) + epspri))
        dd = 0.2 * np.abs(gst)
        dd = np.where(dd > optstp, optstp, dd)
        dd = np.where(dd < dminh, dminh, dd)
        chgold = np.full((m, n), 10000.)
        open_ = np.repeat(good[:, None], n, axis=1)
        for j in range(self.HG_NC):
            r, c = np.nonzero(open_)
            if not r.size:
                break
            k = np.arange(r.size)
            pts = np.repeat(x[r], 2, axis=0)
            pts[2 * k, c] = x[r, c] + dd[r, c]
            pts[2 * k + 1, c] = x[r, c] - dd[r, c]
            vals = self._f(pts, np.repeat(idx[r], 2))
            fs1, fs2 = vals[0::2], vals[1::2]
            dq = dd[r, c]
            grdold = grd[r, c]
            grdnew = (fs1 - fs2) / (2. * dq)
            dgmin = EPS * (np.abs(fs1) + np.abs(fs2)) / dq
            tiny = np.abs(grdnew) < EPS
            with np.errstate(invalid='ignore', divide='ignore'):
                change = np.abs((grdold - grdnew) / grdnew)
            worse = (change > chgold[r, c]) & (j > 1)
            take = ~tiny & ~worse
            chgold[r[take], c[take]] = change[take]
            grd[r[take], c[take]] = grdnew[take]
            gst[r[take], c[take]] = dq[take]
            stop = tiny | worse | (change < 0.05) | (np.abs(grdold - grdnew) < dgmin) | (dq < dminh[r, c])
            open_[r[stop], c[stop]] = False
            dd[r[~stop], c[~stop]] *= 0.2
        # off-diagonal elements: one batch
        pairs = [(i, j) for i in range(n) for j in range(i + 1, n)]
        if pairs and good.any():
            gi = np.flatnonzero(good)
            pts = np.repeat(x[gi], len(pairs), axis=0).reshape(gi.size, len(pairs), n)
            for q, (i, j) in enumerate(pairs):
                pts[:, q, i] += dirin[gi, i]
                pts[:, q, j] += dirin[gi, j]
            vals = self._f(pts.reshape(-1, n), np.repeat(idx[gi], len(pairs))).reshape(gi.size, len(pairs))
            for q, (i, j) in enumerate(pairs):
                el = (vals[:, q] + amin[gi] - yy[gi, i] - yy[gi, j]) / (dirin[gi, i] * dirin[gi, j])
                vh[gi, i, j] = el
                vh[gi, j, i] = el
        for q in np.flatnonzero(good):
            hmat, made = _make_posdef(vh[q])
            try:
                Vq = np.linalg.inv(hmat)
            except np.linalg.LinAlgError:
                failed[q] = True
                continue
            f_ = idx[q]
            S['x'][f_], S['fval'][f_] = x[q], amin[q]
            S['grd'][f_], S['g2'][f_], S['gstep'][f_] = grd[q], g2[q], gst[q]
            S['V'][f_], S['dcovar'][f_] = Vq, 0.
            S['edm'][f_] = 0.5 * float(grd[q] @ Vq @ grd[q])
            S['accurate'][f_] = not made
        S['hesse_failed'][idx[failed]] = True
        S['fval'][idx[failed]] = amin[failed]
        return failed

    # ---- the whole thing
    def run(self):
        F, n = self.F, self.n
        all_idx = np.arange(F)
        x = self.x0.copy()
        fval = self._f(x, all_idx)
        S = dict(x=x, fval=fval, grd=np.zeros((F, n)), g2=np.zeros((F, n)), gstep=np.zeros((F, n)), V=np.zeros((F, n, n)),
                 dcovar=np.ones(F), edm=np.full(F, np.inf), limit=np.zeros(F, dtype=bool), hesse_failed=np.zeros(F, dtype=bool),
                 accurate=np.zeros(F, dtype=bool))
        alive = np.isfinite(fval)
        idx = np.flatnonzero(alive)
        slow = []
        if idx.size:
            grd, g2, gstep = self._initial_gradient(x[idx])
            grd, g2, gstep = self._gradient(idx, x[idx], fval[idx], grd, g2, gstep)
            S['grd'][idx], S['g2'][idx], S['gstep'][idx] = grd, g2, gstep
            diag = np.where(np.abs(g2) > EPS2, 1. / np.where(g2 != 0, g2, 1.), 1.)
            V = np.zeros((idx.size, n, n))
            V[:, np.arange(n), np.arange(n)] = diag
            S['V'][idx] = V
            S['edm'][idx] = 0.5 * np.einsum('fi,fij,fj->f', grd, V, grd)
            neg = (g2 <= 0).any(axis=1)
            slow += list(idx[neg])
            todo = idx[~neg]
            maxfcn_eff = self.maxfcn
            for ipass in range(20):
                slow += self._iterate(todo, S, maxfcn_eff)
                todo = np.array([f for f in todo if f not in set(slow) and not S['limit'][f]], dtype=int)
                need = todo[S['dcovar'][todo] > 0.05]
                again = []
                if need.size:
                    failed = self._hesse(need, S)
                    okk = need[~failed]
                    e = S['edm'][okk]
                    again = list(okk[(e > self.edmval) & (e >= np.abs(EPS2 * S['fval'][okk]))])
                if ipass == 0:
                    maxfcn_eff = int(self.maxfcn * 1.3)
                todo = np.array(again, dtype=int)
                if not todo.size:
                    break
        self.S, self.alive, self.slow = S, alive, sorted(set(int(f) for f in slow))
        return S
