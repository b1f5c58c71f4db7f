"""The MIGRAD state machine (vega_amd/csrc/vmx_migrad.h) on the CPU, under AddressSanitizer / UBSan, against the readable
reference implementation `vega_amd.migrad._Fit`: the header the library compiles into its fit kernels (one thread per fit) is
built here with g++ into tests/helpers/migrad_driver.cpp, which plays the kernels' part - advance every fit, hand out the
parameter rows they ask for, take the function values back - while the function lives in this file.  Asserted fit by fit: the
same NUMBER of function calls and iterations, the same multiset of evaluated points (a decision taken differently anywhere
changes it), the same minimum, error matrix and flags - for the bias pre-fit chained into the full fit, limits of every kind,
fixed parameters, a fit whose model cannot be evaluated, a start with negative curvature, a start at a limit, iminuit's
`iterate` re-runs and a fit that ends at its call limit."""
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REPO
from vega_amd.migrad import MigradMinimizer


@pytest.fixture(scope='module')
def driver(tmp_path_factory):
    gxx = shutil.which('g++')
    if gxx is None:
        pytest.skip('g++ is not installed')
    exe = tmp_path_factory.mktemp('migrad') / 'migrad_driver'
    cmd = [gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-Wall', '-Wextra',
           '-o', str(exe), str(REPO / 'tests' / 'helpers' / 'migrad_driver.cpp')]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert built.returncode == 0, built.stderr[-4000:]
    assert 'warning' not in built.stderr, built.stderr[-4000:]
    return exe


def machine_on_cpu(exe, evaluate, log=None, capacity=None):
    """`machine(plan, ext0, fit_ids)` for MigradMinimizer: the driver process advances the fits, `evaluate(theta, fit)` answers."""
    def fmt(v):
        return repr(float(v))

    def machine(plan, ext0, fit_ids):
        F, P = ext0.shape
        head = [str(len(plan['stages'])), str(P), str(plan['iterate']), str(plan['maxfcn']), fmt(plan['up']), fmt(plan['tol'])]
        for st in plan['stages']:
            head.append(str(len(st['free'])))
            for j, lim, err in zip(st['free'], st['limits'], st['errors']):
                lo, hi = lim
                lo = None if lo is None or not np.isfinite(lo) else float(lo)
                hi = None if hi is None or not np.isfinite(hi) else float(hi)
                head += [str(int(j)), str(int(lo is not None)), str(int(hi is not None)), fmt(lo or 0.), fmt(hi or 0.), fmt(err)]
        head.append(str(F))
        head += [fmt(v) for v in ext0.ravel()]
        cap = capacity or next(c for c in (4, 8, 16, 32) if c >= max(len(st['free']) for st in plan['stages']))
        proc = subprocess.Popen([str(exe), str(cap)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                env={'ASAN_OPTIONS': 'detect_leaks=1', 'UBSAN_OPTIONS': 'print_stacktrace=1'})
        proc.stdin.write(' '.join(head) + '\n')
        proc.stdin.flush()
        outs = [dict(x=np.zeros((F, len(st['free']))), ext=np.zeros((F, len(st['free']))), V=np.zeros((F, len(st['free']), len(st['free']))),
                     fval=np.zeros(F), edm=np.zeros(F), flags=np.zeros(F, dtype=np.int32), nfcn=np.zeros(F, dtype=np.int64),
                     n_iter=np.zeros(F, dtype=np.int32)) for st in plan['stages']]
        rounds = None
        while True:
            line = proc.stdout.readline()
            if not line:
                break
            tok = line.split()
            if tok[0] == 'ROUND':
                rows = [proc.stdout.readline().split() for _ in range(int(tok[1]))]
                owner = np.array([int(r[0]) for r in rows])
                theta = np.array([[float(v) for v in r[1:]] for r in rows])
                if log is not None:
                    log.append((owner.copy(), theta.copy()))
                vals = np.asarray(evaluate(theta, np.asarray(fit_ids)[owner]), dtype=float)
                proc.stdin.write(' '.join(fmt(v) for v in vals) + '\n')
                proc.stdin.flush()
            elif tok[0] == 'RESULT':
                s, f = int(tok[1]), int(tok[2])
                o = outs[s]
                o['fval'][f], o['edm'][f], o['flags'][f], o['nfcn'][f], o['n_iter'][f] = float(tok[3]), float(tok[4]), int(tok[5]), int(tok[6]), int(tok[7])
                n = o['x'].shape[1]
                o['x'][f] = [float(v) for v in proc.stdout.readline().split()[1:]]
                o['ext'][f] = [float(v) for v in proc.stdout.readline().split()[1:]]
                o['V'][f] = np.array([float(v) for v in proc.stdout.readline().split()[1:]]).reshape(n, n)
            elif tok[0] == 'END':
                rounds = int(tok[1])
            else:
                raise AssertionError(line)
        proc.stdin.close()
        err = proc.stderr.read()
        assert proc.wait(timeout=60) == 0 and rounds is not None, err[-3000:]
        assert 'runtime error' not in err and 'AddressSanitizer' not in err, err[-3000:]
        machine.rounds = rounds
        return outs
    return machine


def both(exe, evaluate, names, start0, errors, limits, n_fits, **kw):
    """(reference result, machine result, per-fit evaluated points of each)."""
    minimize_kw = {k: kw.pop(k) for k in ('start', 'fixed', 'prefit_bias') if k in kw}
    ref_calls = []

    def logged(theta, fit):
        ref_calls.append((np.asarray(fit).copy(), np.asarray(theta).copy()))
        return evaluate(theta, fit)
    ref = MigradMinimizer(logged, names, start0, errors, limits, vectorised=False, **kw).minimize(n_fits, **minimize_kw)
    mach_calls = []
    m = MigradMinimizer(None, names, start0, errors, limits, machine=machine_on_cpu(exe, evaluate, mach_calls), **kw)
    got = m.minimize(n_fits, **minimize_kw)

    def per_fit(calls):
        out = {}
        for owner, theta in calls:
            for f, row in zip(owner, theta):
                out.setdefault(int(f), []).append(row)
        return {f: np.array(sorted(map(tuple, rows))) for f, rows in out.items()}
    return ref, got, per_fit(ref_calls), per_fit(mach_calls), m.machine.rounds


def assert_same(ref, got, pts_ref, pts_got, tol=1e-9, pts_tol=1e-10, derived=1.):
    np.testing.assert_array_equal(ref.nfcn, got.nfcn)
    np.testing.assert_array_equal(ref.n_iter, got.n_iter)
    np.testing.assert_array_equal(ref.is_valid, got.is_valid)
    np.testing.assert_array_equal(ref.hesse_failed, got.hesse_failed)
    np.testing.assert_array_equal(ref.has_accurate_covar, got.has_accurate_covar)
    fin = np.isfinite(ref.fval)
    np.testing.assert_array_equal(fin, np.isfinite(got.fval))
    np.testing.assert_allclose(got.values, ref.values, rtol=0, atol=tol)
    np.testing.assert_allclose(got.fval[fin], ref.fval[fin], rtol=tol, atol=tol)
    np.testing.assert_allclose(got.edm[fin], ref.edm[fin], rtol=1e-6 * derived, atol=1e-14)
    np.testing.assert_allclose(got.errors, ref.errors, rtol=1e-7 * derived, atol=1e-14)
    np.testing.assert_allclose(got.covariance, ref.covariance, rtol=1e-6 * derived, atol=1e-13 * derived)
    assert set(pts_ref) == set(pts_got)
    for f in pts_ref:
        assert pts_ref[f].shape == pts_got[f].shape, f
        np.testing.assert_allclose(pts_got[f], pts_ref[f], rtol=0, atol=pts_tol)


def _quadratic(n, n_fits, seed=1):
    rng = np.random.default_rng(seed)
    M = rng.normal(size=(n, n))
    A = M @ M.T + n * np.eye(n)
    centre = rng.normal(size=(n_fits, n)) * 0.05

    def evaluate(theta, fit):
        d = theta - centre[fit]
        return np.einsum('ni,ij,nj->n', d, A, d) * 100 + 0.3 * np.sin(3 * d[:, 0])**2
    return A, centre, evaluate


def test_bias_prefit_chained_into_the_full_fit_with_every_kind_of_limit(driver):
    n = 6
    A, centre, evaluate = _quadratic(n, 48)
    names = ['ap', 'at', 'bias_a', 'b', 'c', 'bias_d']
    limits = [(-1, 1), (-1, 1), (None, None), (-5, None), (None, 5), (-1, 1)]
    ref, got, pr, pg, rounds = both(driver, evaluate, names, [0.] * n, [0.05] * n, limits, 48)
    assert_same(ref, got, pr, pg)
    assert got.is_valid.all()
    # the library instantiates the machine for 4, 8, 16 or 32 free parameters (the smallest that holds the stages: 8 above);
    # the capacity is storage only
    m32 = MigradMinimizer(None, names, [0.] * n, [0.05] * n, limits, machine=machine_on_cpu(driver, evaluate, capacity=32)).minimize(48)
    np.testing.assert_array_equal(m32.nfcn, got.nfcn)
    np.testing.assert_array_equal(m32.values, got.values)
    np.testing.assert_array_equal(m32.covariance, got.covariance)
    # the fits run at their own pace: the machine needs as many rounds as its slowest fit, not the sum of the stages' maxima
    assert rounds < 150


def test_fixed_parameters_and_a_fit_that_cannot_be_evaluated(driver):
    n = 4
    A, centre, evaluate = _quadratic(n, 8, seed=3)

    def failing(theta, fit):
        vals = evaluate(theta, fit)
        vals[np.asarray(fit) == 5] = 1e100
        return vals
    ref, got, pr, pg, _ = both(driver, failing, ['ap', 'bias_x', 'beta', 'bias_y'], [0.1, 0., 0., 0.], [0.05] * n, [(-1, 1)] * n, 8,
                               fixed=('ap',))
    assert_same(ref, got, pr, pg)
    assert np.all(got.values[:, 0] == 0.1) and np.all(got.errors[:, 0] == 0.)
    assert not got.is_valid[5] and got.hesse_failed[5] and not np.isfinite(got.fval[5])


def test_negative_curvature_at_the_start_and_a_start_at_a_limit(driver):
    def evaluate(theta, fit):
        x, y = theta[:, 0], theta[:, 1]
        return 50. * (1. - np.cos(x - 0.3)) + 20. * (y - 0.2)**2 + 5. * x * y
    start = np.array([[2.9, 0.9999], [0.5, 0.0], [-2.8, -0.99], [3.4, 0.3]])
    ref, got, pr, pg, _ = both(driver, evaluate, ['x', 'y'], [2.9, 0.9999], [0.1, 0.1], [(-3.5, 3.5), (-1., 1.)], 4, start=start)
    assert_same(ref, got, pr, pg)
    assert got.is_valid.all()


def test_rosenbrock_one_sided_limits_and_a_single_parameter(driver):
    def rosen(theta, fit):
        x, y = theta[:, 0], theta[:, 1]
        return 100. * (y - x * x)**2 + (1. - x)**2
    ref, got, pr, pg, _ = both(driver, rosen, ['x', 'y'], [-1.2, 1.0], [0.1, 0.1], [(None, None), (None, None)], 1, prefit_bias=False)
    assert_same(ref, got, pr, pg, tol=1e-8, pts_tol=1e-8)      # (a curved valley amplifies last-bit differences of the dot products)
    assert got.values[0] == pytest.approx([1., 1.], abs=2e-3)

    def lower(theta, fit):
        return ((theta[:, 0] - 2.0) / 0.3)**2 + ((theta[:, 1] + 3.0) / 0.5)**2
    ref, got, pr, pg, _ = both(driver, lower, ['x', 'y'], [3.0, 1.0], [0.1, 0.1], [(1.0, None), (None, 2.5)], 2, prefit_bias=False)
    assert_same(ref, got, pr, pg)

    def one(theta, fit):
        return ((theta[:, 0] - 1.0) / 0.25)**2
    ref, got, pr, pg, _ = both(driver, one, ['x'], [0.], [0.1], [(None, None)], 3, prefit_bias=False)
    assert_same(ref, got, pr, pg)
    assert got.errors[0, 0] == pytest.approx(0.25, rel=1e-4)


def test_iterate_reruns_and_the_call_limit(driver):
    # a curved valley with a tight call limit: the run ends at its limit and is reported so (no re-run); with Minuit's default
    # limit the same fits converge after some hundred calls (last-bit differences of the dot products grow along such a path -
    # the two NumPy drivers of vega_amd/migrad.py differ by 1e-5 between THEMSELVES in the middle of fit 0: the calls are counted
    # exactly, the points compared to 1e-4, the minimum to 2e-6)
    def valley(theta, fit):
        x, y, z = theta[:, 0], theta[:, 1], theta[:, 2]
        return 100. * (y - x * x)**2 + (1. - x)**2 + 50. * (z - y * y)**2
    start = np.array([[-1.2, 1.0, 0.5], [0.3, 0.1, 0.9], [2.0, -1.0, 2.0]])
    for maxfcn, tol in ((60, 1e-7), (100000, 2e-6)):
        ref, got, pr, pg, _ = both(driver, valley, ['x', 'y', 'z'], [-1.2, 1.0, 0.5], [0.1, 0.1, 0.1], [(None, None)] * 3, 3,
                                   prefit_bias=False, maxfcn=maxfcn, start=start)
        assert_same(ref, got, pr, pg, tol=tol, pts_tol=1e-7 if maxfcn == 60 else 1e-4, derived=100.)
        assert got.is_valid.all() == (maxfcn > 60)

    # a flat direction: HESSE finds no curvature after five widenings and fails - the minimum is neither valid nor at the call
    # limit, so iminuit's `iterate` loop runs the object again from its last state (values, errors as steps, error matrix as
    # first metric), four more times, and the fit is reported invalid
    def flat(theta, fit):
        return (theta[:, 0] - 0.5)**2 / 0.01 + 0. * theta[:, 1]
    ref, got, pr, pg, _ = both(driver, flat, ['x', 'y'], [0., 0.], [0.1, 0.1], [(None, None), (-1., 1.)], 2, prefit_bias=False)
    assert_same(ref, got, pr, pg)
    assert not got.is_valid.any() and got.hesse_failed.all() and got.nfcn[0] > 250


def test_the_library_compiles_the_tested_header():
    """libvegamx's fit kernels advance the fits with the header the driver tests - no second copy of the algorithm."""
    fit = (REPO / 'vega_amd' / 'csrc' / 'vmx_fit.h').read_text()
    assert '#include "vmx_migrad.h"' in fit and 'vmx_migrad::advance<N>(' in fit
    head = (REPO / 'vega_amd' / 'csrc' / 'vmx_migrad.h').read_text()
    assert 'hip/' not in head and 'hipLaunch' not in head
