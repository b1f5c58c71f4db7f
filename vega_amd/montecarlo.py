"""Monte-Carlo mocks and their batched fits (BASELINE configs[4]).

Mirrors ``Data.create_monte_carlo`` (reference vega/data.py:689-760) and ``Analysis.run_monte_carlo``
(reference vega/analysis.py:224-308, driven by bin/run_vega_mc_mpi.py:52-71): a mock is the fiducial model
plus ``cholesky(scale * C_masked) . randn(n_masked)`` drawn from NumPy's legacy global generator after
``np.random.seed(seed)`` - per mock, item by item - so the same seed gives the reference's mocks bit for bit.
The fits are where the GPU changes the algorithm: instead of one MIGRAD after another, all mocks of a rank are
minimised in lock-step by :class:`vega_amd.minimizer.BatchedMinimizer`, each batch of trial points being one
engine call with a per-walker mock index (``vmx_set_mock_index``).
"""
import numpy as np

from .minimizer import BatchedMinimizer
from .parallel import shard_bounds


def create_mocks(problem, fiducial_model, num_mocks, seed=0, scale=None, forecast=False, matmul=None):
    """dict name -> [num_mocks, n_masked] masked mock data vectors, in the reference's draw order.
    ``matmul(L, Z) -> Z @ L.T`` (default: NumPy) applies the Cholesky factor to all draws of an item at once -
    the driver passes the engine's product so that the only O(n^2) step per mock runs on the GPU."""
    scale = 1. if scale is None else scale
    np.random.seed(seed)
    chol = {}
    fid = {}
    for name, item in problem.items.items():
        model = np.asarray(fiducial_model[name], dtype=float)
        if model.size != item.data_vec.size:
            if model.size != item.dist_grid.size:
                raise ValueError('Could not match fiducial model to data or model size.')
            # distorted-model grid -> data grid (reference vega/coordinates.py:127-144)
            keep = (item.dist_grid.rp >= item.data_grid.rp_min) & (item.dist_grid.rp <= item.data_grid.rp_max)
            keep &= item.dist_grid.rt <= item.data_grid.rt_max
            model = model[keep]
        fid[name] = model[item.data_mask]
        if not forecast:
            cache = item.__dict__.setdefault('_masked_cholesky', {})       # one factorisation per (item, scale)
            if scale not in cache:
                n = item.data_size
                cov = np.eye(n) if item.cov is None else item.cov[:, item.data_mask][item.data_mask, :]
                cache[scale] = np.linalg.cholesky(scale * cov)
            chol[name] = cache[scale]
    if forecast:
        return {name: np.tile(fid[name], (num_mocks, 1)) for name in problem.items}
    # the legacy global generator is consumed mock by mock, item by item (reference vega/data.py:751-753)
    draws = {name: np.empty((num_mocks, item.data_size)) for name, item in problem.items.items()}
    for i in range(num_mocks):
        for name, item in problem.items.items():
            draws[name][i] = np.random.randn(item.data_size)
    if matmul is None:
        def matmul(L, Z):
            return Z @ L.T
    return {name: fid[name][None, :] + matmul(chol[name], draws[name]) for name in problem.items}


class MonteCarlo:
    """Results container with the attribute names of the reference's ``Analysis`` (vega/analysis.py:246-308)."""

    def __init__(self, vega):
        self.vega = vega
        self.has_monte_carlo = False

    def minimizer(self, sample_params=None, tol=0.1):
        """BatchedMinimizer over ``sample_params`` (default: the interface's [sample] section) whose objective
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
        return BatchedMinimizer(evaluate, names, start, errors, limits, tol=tol)

    def run_monte_carlo(self, fiducial_model, num_mocks=1, seed=0, scale=None, forecast=False,
                        run_mc_fits=True, sample_params=None):
        vega = self.vega
        eng = vega.engine
        mocks = create_mocks(vega.problem, fiducial_model, num_mocks, seed=seed, scale=scale, forecast=forecast,
                             matmul=eng.matmul_host)
        self.mc_mocks = mocks
        if not run_mc_fits:
            self.has_monte_carlo = True
            return None
        for name, pool in mocks.items():
            eng.set_mock_pool(name, pool)
            if scale is not None and vega.problem.items[name].cov is not None and not vega._use_global_cov:
                eng.set_invcov(name, vega.problem.items[name].chi2_matrix / scale)
        fitter = self.minimizer(sample_params)
        self._mock_rows = np.arange(num_mocks, dtype=np.int32)
        try:
            res = fitter.minimize(n_fits=num_mocks)
        finally:
            eng.set_mock_index(None)
            self._mock_rows = None
            if scale is not None:
                for name, item in vega.problem.items.items():
                    if item.cov is not None and not vega._use_global_cov:
                        eng.set_invcov(name, item.chi2_matrix)
        self.fit_result = res
        self.mc_bestfits = {n: np.stack([res.values[:, j], res.errors[:, j]], axis=1)
                            for j, n in enumerate(res.names)}
        self.mc_covariances = list(res.covariance)
        self.mc_chisq = list(res.fval)
        self.mc_valid_minima = list(res.is_valid)
        self.mc_valid_hesse = list(~res.hesse_failed)
        self.mc_failed_mask = list(~np.isfinite(res.fval))
        self.has_monte_carlo = True
        return res

    def write(self, directory, cpu_id=None, overwrite=False):
        """`monte_carlo[_<cpu_id>].fits` in the reference's layout (reference vega/output.py:442-520)."""
        from .output import write_monte_carlo
        return write_monte_carlo(self, directory, cpu_id=cpu_id, overwrite=overwrite)


def run_monte_carlo_sharded(vega, fiducial_model, num_mc_mocks, seed=0, rank=0, world_size=1, output_dir=None,
                            **kw):
    """The rank's share of ``num_mc_mocks`` with the reference's seeding: ceil(N / size) mocks per rank drawn
    from ``seed + rank``, and one result file per rank when ``output_dir`` is given (reference
    bin/run_vega_mc_mpi.py:52-71)."""
    lo, hi = shard_bounds(num_mc_mocks, world_size, rank)
    per = -(-num_mc_mocks // world_size)
    mc = MonteCarlo(vega)
    res = mc.run_monte_carlo(fiducial_model, num_mocks=per, seed=int(seed + rank), **kw)
    if output_dir is not None:
        mc.write(output_dir, cpu_id=rank if world_size > 1 else None, overwrite=True)
    return mc, res, (lo, hi)
