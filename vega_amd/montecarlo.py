"""Monte-Carlo mocks and their batched fits (BASELINE configs[4]).

Mirrors ``Data.create_monte_carlo`` (reference vega/data.py:689-760), ``Analysis.create_monte_carlo_sim`` /
``create_global_monte_carlo`` (reference vega/analysis.py:124-222) and ``Analysis.run_monte_carlo``
(reference vega/analysis.py:224-308, driven by bin/run_vega_mc_mpi.py:52-71).

* independent correlations: a mock is the fiducial model plus ``cholesky(scale * C_masked) . randn(n_masked)`` drawn
  from NumPy's legacy global generator after ``np.random.seed(seed)`` - per mock, item by item;
* global covariance (``global-cov-file``): ONE ``randn(sum n_masked)`` per mock through the Cholesky factor of the
  masked global covariance, so the mocks carry the cross-covariance between the correlations;

either way the same seed gives the reference's mocks to rounding.  The fits are where the GPU changes the algorithm:
instead of one MIGRAD after another, all mocks of a rank are minimised in lock-step by
:class:`vega_amd.minimizer.BatchedMinimizer`, each batch of trial points being one engine call with a per-walker mock
index (``vmx_set_mock_index``).
"""
import os

import numpy as np

from .minimizer import BatchedMinimizer
from .parallel import shard_bounds


def _fiducial_on_data_grid(item, model):
    """The fiducial model on the item's data grid (reference vega/data.py:737-742, analysis.py:197-206)."""
    model = np.asarray(model, dtype=float)
    if model.size == item.data_vec.size:
        return model
    if model.size != item.dist_grid.size:
        raise ValueError('Could not match fiducial model to data or model size.')
    # distorted-model grid -> data grid (reference vega/coordinates.py:127-144)
    keep = (item.dist_grid.rp >= item.data_grid.rp_min) & (item.dist_grid.rp <= item.data_grid.rp_max)
    keep &= item.dist_grid.rt <= item.data_grid.rt_max
    return model[keep]


def item_scales(problem, scale):
    """Covariance scale of every item as ``create_monte_carlo_sim`` resolves it (reference analysis.py:147-156):
    None -> the item's ``cov_rescale`` (and Data.create_monte_carlo turns a None into 1), a number -> that number,
    a dict -> its entry, 1 for items it does not name."""
    out = {}
    for name, item in problem.items.items():
        if scale is None:
            s = item.cov_rescale
        elif isinstance(scale, (int, float)) and not isinstance(scale, bool):
            s = scale
        elif isinstance(scale, dict) and name in scale:
            s = scale[name]
        else:
            s = 1.
        out[name] = 1. if s is None else float(s)
    return out


def _default_matmul(L, Z):
    return Z @ L.T


def create_mocks(problem, fiducial_model, num_mocks, seed=0, scale=None, forecast=False, matmul=None,
                 reseed_per_item=False):
    """dict name -> [num_mocks, n_masked] masked mock data vectors of independent correlations, in the reference's
    draw order.  ``reseed_per_item``: every item's draw starts from ``seed`` again - what
    ``create_monte_carlo_sim(seed=seed)`` does when ``initialize_monte_carlo`` calls it (reference analysis.py:159-160 ->
    data.py:735-736), as opposed to the one seeding of ``run_monte_carlo`` (analysis.py:246).  ``matmul(L, Z) -> Z @ L.T`` (default: NumPy) applies the Cholesky factor to all draws of an item at
    once - the driver passes the engine's product so that the only O(n^2) step per mock runs on the GPU."""
    if problem.global_cov is not None:
        raise ValueError('this problem has a global covariance: its mocks come from create_global_mocks')
    scales = item_scales(problem, scale)
    matmul = matmul or _default_matmul
    if seed is not None:            # (None: the caller's generator state goes on, reference data.py:735-736)
        np.random.seed(seed)
    chol, fid = {}, {}
    for name, item in problem.items.items():
        fid[name] = _fiducial_on_data_grid(item, fiducial_model[name])
        if forecast:
            continue
        if item.cov is None:
            raise ValueError(f'{name}: Monte-Carlo mocks need a covariance matrix')
        cache = item.__dict__.setdefault('_cholesky', {})       # one factorisation per (item, scale, masked or full)
        key = (scales[name], bool(item.cholesky_masked_cov))
        if key not in cache:
            cov = item.cov[:, item.data_mask][item.data_mask, :] if item.cholesky_masked_cov else item.cov
            cache[key] = np.linalg.cholesky(scales[name] * cov)
        chol[name] = cache[key]
    if forecast:
        return {name: np.tile(fid[name][item.data_mask], (num_mocks, 1)) for name, item in problem.items.items()}
    # the legacy global generator is consumed mock by mock, item by item (reference vega/data.py:748-757): n_masked
    # numbers per item, or the full data size with `cholesky-masked-cov = False`
    draws = {name: np.empty((num_mocks, chol[name].shape[0])) for name in problem.items}
    if reseed_per_item and seed is not None:
        for i in range(num_mocks):
            for name in problem.items:
                np.random.seed(seed)
                draws[name][i] = np.random.randn(chol[name].shape[0])
    else:
        # (the legacy generator's stream does not depend on how it is cut into calls - its cached second Gaussian of a pair goes on
        # to the next call - so all draws come from ONE call and are dealt out in the reference's order: a mock's items side by side)
        sizes = [chol[name].shape[0] for name in problem.items]
        flat = np.random.randn(num_mocks * sum(sizes)).reshape(num_mocks, sum(sizes))
        lo = 0
        for name, n in zip(problem.items, sizes):
            draws[name] = np.ascontiguousarray(flat[:, lo:lo + n])
            lo += n
    out = {}
    for name, item in problem.items.items():
        noise = matmul(chol[name], draws[name])
        if item.cholesky_masked_cov:
            out[name] = fid[name][item.data_mask][None, :] + noise
        else:
            out[name] = (fid[name][None, :] + noise)[:, item.data_mask]
    return out


def create_global_mocks(problem, fiducial_model, num_mocks, seed=0, scale=None, forecast=False, matmul=None):
    """[num_mocks, sum n_masked] mocks of the concatenated masked data vector from the global covariance (reference
    vega/analysis.py:164-222): one Cholesky factor of ``scale *`` the masked global covariance - cached with the FIRST
    scale it is built with, as the reference caches it - and one ``randn`` per mock over all correlations."""
    gm = problem.global_masks()
    if gm is None:
        raise ValueError('create_global_mocks requires a global covariance matrix')
    matmul = matmul or _default_matmul
    if seed is not None:
        np.random.seed(seed)
    mask = gm['data_mask']
    fid = np.concatenate([_fiducial_on_data_grid(item, fiducial_model[name])
                          for name, item in problem.items.items()])[mask]
    if forecast:
        return np.tile(fid, (num_mocks, 1))
    if gm.get('cholesky') is None:
        masked = problem.global_cov[:, mask][mask, :]
        gm['cholesky'] = np.linalg.cholesky((1 if scale is None else scale) * masked)
    n = int(mask.sum())
    draws = np.empty((num_mocks, n))
    for i in range(num_mocks):
        draws[i] = np.random.randn(n)
    return fid[None, :] + matmul(gm['cholesky'], draws)


def split_global(problem, vectors):
    """[num_mocks, sum n_masked] -> dict name -> [num_mocks, n_masked] in item order."""
    out, lo = {}, 0
    for name, item in problem.items.items():
        out[name] = np.ascontiguousarray(vectors[:, lo:lo + item.data_size])
        lo += item.data_size
    return out


class MonteCarlo:
    """Results container with the attribute names of the reference's ``Analysis`` (vega/analysis.py:246-308)."""

    def __init__(self, vega):
        self.vega = vega
        self.has_monte_carlo = False
        self.current_mc_mock = None
        # who advances the MIGRAD fits: 'device' - the state machines of vega_amd/csrc/vmx_migrad.h in the engine's fit kernels
        # (parameter rows, chi2 and the fits' states stay in HBM, include/vegamx.h: vmx_fit_migrad) - or 'python' - the NumPy
        # lock-step drivers of vega_amd/migrad.py through the engine's host entry (the readable reference; the only one for
        # engine groups and for method='bfgs')
        self.driver = os.environ.get('VEGA_AMD_FIT_DRIVER', 'device')
        self.driver_stats = None
        # with the device driver the mocks of `run_monte_carlo` are made while their fits run (the normal draws on a host thread -
        # NumPy's legacy generator is sequential -, everything else on the device, wave by wave: include/vegamx.h vmx_mock_stream);
        # False: all mocks first, as the reference's loop reads (same numbers to rounding)
        self.stream_mocks = os.environ.get('VEGA_AMD_STREAM_MOCKS', '1') != '0'
        self._mock_stream = None

    def minimizer(self, sample_params=None, tol=0.1, method='migrad'):
        """Minimiser over ``sample_params`` (default: the interface's [sample] section) - ``method='migrad'``: MIGRAD's own
        sequence of steps per fit (vega_amd/migrad.py: what the reference runs through iminuit, vega/minimizer.py:66-97),
        ``'bfgs'``: the vectorised variable-metric minimiser of vega_amd/minimizer.py (Minuit conventions, not its
        trajectory; fewer Python steps per fit) - whose objective
        is the engine: walkers = defaults with the sampled columns replaced, data = the walker's mock."""
        vega = self.vega
        sp = vega.sample_params if sample_params is None else sample_params
        names = list(sp['limits'].keys())
        if not names:
            raise ValueError('no sampled parameters')
        eng = vega.engine
        cols = np.array([eng.low.slot[n] for n in names])
        theta0 = eng.low.theta0

        def evaluate(x_ext, fit_index):
            n = x_ext.shape[0]
            out = np.empty(n)
            mb = eng.max_batch
            for lo in range(0, n, mb):
                hi = min(lo + mb, n)
                theta = np.tile(theta0, (hi - lo, 1))
                theta[:, cols] = x_ext[lo:hi]
                if self._mock_rows is not None:
                    eng.set_mock_index(self._mock_rows[fit_index[lo:hi]])
                out[lo:hi] = eng.eval(theta)[0]
            return out

        self._mock_rows = None
        start = [sp['values'][n] for n in names]
        errors = [sp['errors'][n] for n in names]
        limits = [sp['limits'][n] for n in names]
        self._fixed = tuple(n for n in names if sp.get('fix', {}).get(n, False))
        if method == 'migrad':
            from .migrad import MigradMinimizer
            machine = None
            if self.driver == 'device' and hasattr(eng, 'fit_migrad'):
                def machine(plan, ext0, fit_ids):
                    theta = np.tile(theta0, (ext0.shape[0], 1))
                    theta[:, cols] = ext0
                    on_engine = dict(plan, stages=[dict(st, free=cols[st['free']]) for st in plan['stages']])
                    rows = None if self._mock_rows is None else self._mock_rows[np.asarray(fit_ids)]
                    outs, self.driver_stats = eng.fit_migrad(on_engine, theta, rows, mock_stream=self._mock_stream,
                                                             chunk=int(os.environ.get('VEGA_AMD_FIT_CHUNK', '0')))
                    return outs
            return MigradMinimizer(evaluate, names, start, errors, limits, tol=tol, machine=machine)
        if method != 'bfgs':
            raise ValueError("method: 'migrad' or 'bfgs'")
        return BatchedMinimizer(evaluate, names, start, errors, limits, tol=tol)

    def chi2_scan(self, method='migrad'):
        """chi2 scan over the one or two parameters of the ``[chi2 scan]`` section (``name = start end num_points``;
        reference Analysis.chi2_scan, vega/analysis.py:53-122): at every grid point the scanned parameters are pinned
        (``fix`` / ``values`` overrides of the sampling table, errors 0) and the others are fitted from their configured
        start values.  The reference runs one MIGRAD after the other; here ALL grid points' fits advance in lock-step, their
        trial points joined into engine batches.  Returns the reference's list of dicts (best-fit values + ``'fval'``),
        first parameter outermost; ``self.grids`` keeps the grids."""
        vega = self.vega
        config = vega.main_config
        if config is None or 'chi2 scan' not in config:
            raise ValueError('Called chi2_scan, but no config specified in main.ini. Add a "[chi2 scan]" section to main.')
        self.grids = {}
        for param, value in config.items('chi2 scan'):
            start, end, num = value.split()[:3]
            self.grids[param] = np.linspace(float(start), float(end), int(num))
        if len(self.grids) > 2:
            raise ValueError('chi2_scan only supports one/two parameter scans')
        sp = vega.sample_params
        names = list(sp['limits'].keys())
        for param in self.grids:
            if param not in names:
                raise KeyError(f'{param}: a scanned parameter must be listed under [sample] (the scan pins it there)')
        grid_names = list(self.grids)
        mesh = np.meshgrid(*[self.grids[g] for g in grid_names], indexing='ij')
        points = np.stack([m.ravel() for m in mesh], axis=1)            # first parameter outermost
        vega.freeze_metals()
        vega._sync_monte_carlo()
        sample = {key: dict(sp.get(key, {})) for key in ('limits', 'values', 'errors', 'fix')}
        for g in grid_names:
            sample['fix'][g] = True
        fitter = self.minimizer(sample, method=method)
        start = np.tile([sample['values'][n] for n in names], (points.shape[0], 1))
        for c, g in enumerate(grid_names):
            start[:, names.index(g)] = points[:, c]
        res = fitter.minimize(n_fits=points.shape[0], start=start, fixed=self._fixed)
        self.scan_fits = res
        self.scan_results = []
        for i in range(points.shape[0]):
            row = {n: float(res.values[i, j]) for j, n in enumerate(names)}
            row['fval'] = float(res.fval[i])
            self.scan_results.append(row)
        return self.scan_results

    def create_mocks(self, fiducial_model, num_mocks=1, seed=0, scale=None, forecast=False, reseed_per_item=False):
        """dict name -> [num_mocks, n_masked]: the reference's mocks for ``seed``, per item or - when the problem has a
        global covariance - split from the global draw (kept whole in ``mc_mocks['global']``)."""
        vega = self.vega
        prob = vega.problem
        if prob.global_cov is not None:
            # (scale = None means 1 here, as in Analysis.create_global_monte_carlo: `[control] global_cov_rescale` is read by
            # VegaInterface.initialize_monte_carlo only - reference vega_interface.py:531-533 - not by run_monte_carlo /
            # bin/run_vega_mc_mpi.py)
            whole = create_global_mocks(prob, fiducial_model, num_mocks, seed=seed, scale=scale, forecast=forecast,
                                        matmul=vega.engine.matmul_host)
            self.mc_mocks = {'global': whole}
            self.current_mc_mock = whole[-1]
            return split_global(prob, whole)
        mocks = create_mocks(prob, fiducial_model, num_mocks, seed=seed, scale=scale, forecast=forecast,
                             matmul=vega.engine.matmul_host, reseed_per_item=reseed_per_item)
        self.mc_mocks = mocks
        return mocks

    def install_mocks(self, mocks, scale=None):
        """Make row 0 of every correlation's mocks the data the following chi2 / fits read in Monte-Carlo mode - what
        ``Data.create_monte_carlo`` leaves on the reference's Data objects (vega/data.py:711-722, :749-759): ``masked_mc_mock``
        and, for a rescaled covariance, ``scaled_inv_masked_cov`` / ``scaled_log_cov_det``.  Returns the mocks on the full data
        grids (NaN outside the masks)."""
        vega = self.vega
        scales = item_scales(vega.problem, scale)
        out = {}
        for name, pool in mocks.items():
            view, item = vega.data[name], vega.problem.items[name]
            view.masked_mc_mock = np.array(pool[0])
            if not vega._use_global_cov and item.cov is not None and scales[name] != 1.:
                # reference data.py:717-719: the mock's covariance scale carries over to the fit (its log-determinant
                # term as the reference writes it: log(scale) + log det C)
                view.scaled_inv_masked_cov = item.inv_masked_cov / scales[name]      # (marginalize-in-fit: projected when it is sent)
                view.scaled_log_cov_det = np.log(scales[name]) + item.log_cov_det
            elif not vega._use_global_cov and item.cov is not None and getattr(view, 'scaled_inv_masked_cov', None) is not None:
                # (the reference resets both on every call, data.py:711-722: a scale of 1 after another scale must not keep the
                # other scale's matrix on the view - _sync_monte_carlo would leave it on the engine)
                view.scaled_inv_masked_cov = item.inv_masked_cov
                view.scaled_log_cov_det = item.log_cov_det
            full = np.full(item.data_vec.size, np.nan)
            full[item.data_mask] = pool[0]
            out[name] = full
        return out

    def create_monte_carlo_sim(self, fiducial_model, seed=None, scale=None, forecast=False):
        """One mock per correlation, installed as its Monte-Carlo data (reference Analysis.create_monte_carlo_sim,
        vega/analysis.py:126-162 -> Data.create_monte_carlo, vega/data.py:689-760): every correlation's draw starts from
        ``seed``; ``seed=None`` goes on with the caller's generator state.  dict name -> mock on the full data grid."""
        vega = self.vega
        mocks = create_mocks(vega.problem, fiducial_model, 1, seed=seed, scale=scale, forecast=forecast,
                             matmul=vega.engine.matmul_host, reseed_per_item=True)
        self.mc_mocks = mocks
        return self.install_mocks(mocks, scale)

    def create_global_monte_carlo(self, fiducial_model, seed=None, scale=None, forecast=False):
        """One mock of the global masked data vector from the global covariance, kept as ``current_mc_mock`` (reference
        Analysis.create_global_monte_carlo, vega/analysis.py:164-222); returned."""
        vega = self.vega
        whole = create_global_mocks(vega.problem, fiducial_model, 1, seed=seed, scale=scale, forecast=forecast,
                                    matmul=vega.engine.matmul_host)
        self.mc_mocks = {'global': whole}
        self.current_mc_mock = whole[-1]
        return self.current_mc_mock

    def run_monte_carlo(self, fiducial_model, num_mocks=1, seed=0, scale=None, forecast=False,
                        run_mc_fits=True, sample_params=None, method='migrad'):
        vega = self.vega
        eng = vega.engine
        prob = vega.problem
        if (run_mc_fits and method == 'migrad' and self.driver == 'device' and self.stream_mocks and hasattr(eng, 'fit_migrad')
                and prob.global_cov is None and not vega._use_global_cov and not forecast
                and all(item.cov is not None and item.cholesky_masked_cov for item in prob.items.values())):
            return self._fit_streamed_mocks(fiducial_model, num_mocks, seed=seed, scale=scale, sample_params=sample_params)
        mocks = self.create_mocks(fiducial_model, num_mocks, seed=seed, scale=scale, forecast=forecast)
        if not run_mc_fits:
            self.has_monte_carlo = True
            return None
        return self._fit_mocks(mocks, num_mocks, scale=scale, sample_params=sample_params, method=method)

    def _fit_streamed_mocks(self, fiducial_model, num_mocks, seed=0, scale=None, sample_params=None, wave=64):
        """`run_monte_carlo` with the mocks made WHILE their fits run.  The reference draws a mock's numbers from NumPy's legacy
        global generator, mock after mock, a mock's correlations in turn (vega/data.py:748-757) - a sequential stream, ~10 ns per
        number, a fifth of the whole run when it comes first.  Here a host thread produces exactly that stream (the generator's
        output does not depend on how it is cut into calls) wave by wave into a buffer the fit driver reads; everything else of a
        mock - fiducial + cholesky . draws, its row of the pools, its terms of the quadratic form - happens on the device when
        its wave joins the fits (include/vegamx.h: vmx_mock_stream).  Same mocks as `create_mocks` to rounding."""
        import threading
        vega = self.vega
        eng = vega.engine
        prob = vega.problem
        if sample_params is None and prob.mc_config is not None:
            sample_params = prob.mc_config['sample']
        scales = item_scales(prob, scale)
        stride = int(sum(item.data_size for item in prob.items.values()))
        # (page-locked: the waves' uploads are plain DMA, and the producer below does not first-touch pages the runtime is pinning)
        draws = eng.pinned_empty((num_mocks, stride)) if hasattr(eng, 'pinned_empty') else np.empty((num_mocks, stride))
        counter = np.zeros(1, dtype=np.int32)
        failure = []

        def produce():
            try:
                if seed is not None:
                    np.random.seed(seed)
                for a in range(0, num_mocks, wave):
                    b = min(a + wave, num_mocks)
                    draws[a:b] = np.random.randn((b - a) * stride).reshape(b - a, stride)
                    counter[0] = b
            except BaseException as exc:        # (the consumer must not wait for ever)
                failure.append(exc)
                counter[0] = num_mocks
        producer = threading.Thread(target=produce, name='mock-draws', daemon=True)
        producer.start()        # (first thing: the stream is the one sequential part, everything below runs next to it)
        try:
            for name, item in prob.items.items():
                cache = item.__dict__.setdefault('_cholesky', {})
                key = (scales[name], True)
                if key not in cache:
                    cache[key] = np.linalg.cholesky(scales[name] * item.cov[:, item.data_mask][item.data_mask, :])
                fid = _fiducial_on_data_grid(item, fiducial_model[name])[item.data_mask]
                sent = eng.__dict__.setdefault('_mock_factor_sent', {})
                if name not in sent or sent[name][0] is not cache[key] or not np.array_equal(sent[name][1], fid):
                    eng.set_mock_factor(name, cache[key], fid)         # (kept on the device until another factor / fiducial comes)
                    sent[name] = (cache[key], fid.copy())
                if scales[name] != 1.:
                    eng.set_invcov(name, item.chi2_matrix / scales[name])
        except BaseException:
            producer.join()
            raise
        fitter = self.minimizer(sample_params, method='migrad')
        self._mock_rows = np.arange(num_mocks, dtype=np.int32)
        self._mock_stream = dict(draws=draws, counter=counter, wave=wave, timeout=300.)
        try:
            res = fitter.minimize(n_fits=num_mocks, fixed=self._fixed)
        finally:
            producer.join()
            self._mock_rows = None
            self._mock_stream = None
            for name, item in prob.items.items():
                if scales[name] != 1.:
                    eng.set_invcov(name, item.chi2_matrix)
        if failure:
            raise failure[0]
        self.mc_mocks = {name: eng.get_mock_pool(name, num_mocks) for name in prob.items}
        return self._keep_results(res)

    def fit_global_mocks(self, mocks, start1=None, end1=None, start2=None, end2=None, scale=None, sample_params=None,
                         method='migrad'):
        """Fit EXISTING mocks of the global masked data vector instead of drawing them (reference
        bin/run_vega_mc_fits_mpi.py:11-79): ``mocks`` [n, length]; with the four slice bounds every mock is cut to
        ``r_[mock[start1:end1], mock[start2:end2]]`` first (mocks of a longer joint vector, :41-47).  All fits run in lock-step;
        results in the attributes of ``run_monte_carlo``, ``mc_mocks = {'global': the (cut) mocks}``."""
        vega = self.vega
        prob = vega.problem
        if prob.global_cov is None:
            raise ValueError('fit_global_mocks: the mocks are vectors of the global masked data - a `global-cov-file` is needed')
        whole = np.atleast_2d(np.asarray(mocks, dtype=float))
        if not (start1 is None or end1 is None or start2 is None or end2 is None):
            whole = np.concatenate([whole[:, start1:end1], whole[:, start2:end2]], axis=1)
        n_global = sum(item.data_size for item in prob.items.values())
        if whole.shape[1] != n_global:
            raise ValueError(f'the mocks have {whole.shape[1]} entries, the global masked data vector {n_global}')
        whole = np.ascontiguousarray(whole)
        self.mc_mocks = {'global': whole}
        self.current_mc_mock = whole[-1]
        vega.freeze_metals()
        return self._fit_mocks(split_global(prob, whole), whole.shape[0], scale=scale, sample_params=sample_params, method=method)

    def _fit_mocks(self, mocks, num_mocks, scale=None, sample_params=None, method='migrad'):
        vega = self.vega
        eng = vega.engine
        prob = vega.problem
        if sample_params is None and prob.mc_config is not None:
            sample_params = prob.mc_config['sample']       # the [monte carlo] section (reference analysis.py:249)
        scales = item_scales(prob, scale)
        for name, pool in mocks.items():
            eng.set_mock_pool(name, pool)
            if scales[name] != 1. and prob.items[name].cov is not None and not vega._use_global_cov:
                # (marginalize-in-fit: chi2_matrix is P^T C^-1 P with the projector of the unscaled covariance, which the
                # reference keeps when the covariance is rescaled - vega_interface.py:282-292, :311-313)
                eng.set_invcov(name, prob.items[name].chi2_matrix / scales[name])
        fitter = self.minimizer(sample_params, method=method)
        self._mock_rows = np.arange(num_mocks, dtype=np.int32)
        try:
            res = fitter.minimize(n_fits=num_mocks, fixed=self._fixed)
        finally:
            if getattr(fitter, 'machine', None) is None:     # (the device-resident fits state their rows per call, nothing is left on the engine)
                eng.set_mock_index(None)
            self._mock_rows = None
            for name, item in prob.items.items():
                if scales[name] != 1. and item.cov is not None and not vega._use_global_cov:
                    eng.set_invcov(name, item.chi2_matrix)
        return self._keep_results(res)

    def _keep_results(self, res):
        self.fit_result = res
        res.driver_stats = self.driver_stats if self.driver == 'device' else None
        # a fit that could not run has no Bestfit / covariance row and chisq = NaN (reference analysis.py:279-297)
        failed = ~np.isfinite(res.fval)
        ok = ~failed
        self.mc_bestfits = {n: np.stack([res.values[ok, j], res.errors[ok, j]], axis=1)
                            for j, n in enumerate(res.names)}
        self.mc_covariances = list(res.covariance[ok])
        self.mc_chisq = list(np.where(failed, np.nan, res.fval))
        self.mc_valid_minima = list(res.is_valid)
        self.mc_valid_hesse = list(~res.hesse_failed)
        self.mc_failed_mask = list(failed)
        self.has_monte_carlo = True
        return res

    def write(self, directory, cpu_id=None, overwrite=False):
        """`monte_carlo[_<cpu_id>].fits` in the reference's layout (reference vega/output.py:442-520)."""
        from .output import write_monte_carlo
        return write_monte_carlo(self, directory, cpu_id=cpu_id, overwrite=overwrite)


def contiguous_share(n, world_size, rank):
    """[start, stop) of rank's share of n tasks, the first ``n % world_size`` ranks taking one more (reference
    bin/run_vega_mc_fits_mpi.py:134-141)."""
    per, rem = divmod(int(n), int(world_size))
    if rank < rem:
        start = rank * (per + 1)
        return start, start + per + 1
    start = rank * per + rem
    return start, start + per


def fit_mocks_sharded(vega, mocks, slices=(None, None, None, None), rank=0, world_size=1, output_dir=None, **kw):
    """The rank's contiguous share of a file's global mocks, fitted in lock-step; one result file per rank when
    ``output_dir`` is given (reference bin/run_vega_mc_fits_mpi.py:127-163)."""
    lo, hi = contiguous_share(len(mocks), world_size, rank)
    mc = MonteCarlo(vega)
    vega.analysis = mc
    res = mc.fit_global_mocks(mocks[lo:hi], *slices, **kw) if hi > lo else None
    if output_dir is not None and res is not None:
        mc.write(output_dir, cpu_id=rank if world_size > 1 else None, overwrite=True)
    return mc, res, (lo, hi)


def run_monte_carlo_sharded(vega, fiducial_model, num_mc_mocks, seed=0, rank=0, world_size=1, output_dir=None,
                            **kw):
    """The rank's share of ``num_mc_mocks`` with the reference's seeding: ceil(N / size) mocks per rank drawn
    from ``seed + rank``, and one result file per rank when ``output_dir`` is given (reference
    bin/run_vega_mc_mpi.py:52-71; scripts/run_mc_sharded.py is the launcher)."""
    lo, hi = shard_bounds(num_mc_mocks, world_size, rank)
    per = -(-num_mc_mocks // world_size)
    mc = MonteCarlo(vega)
    vega.analysis = mc
    res = mc.run_monte_carlo(fiducial_model, num_mocks=per, seed=int(seed + rank), **kw)
    if output_dir is not None:
        mc.write(output_dir, cpu_id=rank if world_size > 1 else None, overwrite=True)
    return mc, res, (lo, hi)
