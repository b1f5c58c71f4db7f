"""world_size-2 gloo test (CPU) of the sharded Monte-Carlo launcher ``scripts/run_mc_sharded.py`` - the reference's
``bin/run_vega_mc_mpi.py:17-71`` with one process per GPU.  The GPU engine is replaced by a small stand-in with the same
methods (a linear model with a diagonal chi2), so what is tested is the driver: rank -> seed / mock-count arithmetic,
mock generation in the reference's draw order per rank, lock-step fits through per-walker mock indices, barriers, and
one ``monte_carlo_<rank>.fits`` per rank in the reference's layout.
"""
import os
import socket
import sys
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import GOLDEN, REPO

NUM_MOCKS, SEED = 5, 11
NAMES = ['ap', 'at']


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _problem():
    from vega_amd import synthetic
    from vega_amd.setup import build_problem
    prob = build_problem('configs/joint/main.ini', search_dirs=[GOLDEN])
    for item in prob.items.values():
        item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt))
    prob.mc_config = {'params': {}, 'sample': {
        'limits': {'ap': (0.5, 1.5), 'at': (0.5, 1.5)}, 'values': {'ap': 1.0, 'at': 1.0},
        'errors': {'ap': 0.01, 'at': 0.01}, 'fix': {'ap': False, 'at': False}}}
    prob.sample_params = {'limits': {}, 'values': {}, 'errors': {}, 'fix': {}}
    prob.main_config['control'] = {'run_montecarlo': 'True', 'num_mc_mocks': str(NUM_MOCKS), 'mc_seed': str(SEED)}
    return prob


class _StandInEngine:
    """Same methods as vega_amd.engine.Engine, evaluated with NumPy: model = fid + sum_j (theta_j - 1) t_j per item,
    chi2 = sum |data - model|^2 / var."""

    def __init__(self, prob):
        self.prob = prob
        self.names = sorted(prob.params)
        self.low = SimpleNamespace(slot={n: i for i, n in enumerate(self.names)},
                                   theta0=np.array([prob.params[n] for n in self.names], dtype=float))
        self.max_batch = 64
        rng = np.random.default_rng(3)
        self.fid = {n: 1e-3 * rng.standard_normal(it.data_size) for n, it in prob.items.items()}
        self.templates = {n: 1e-3 * rng.standard_normal((len(NAMES), it.data_size)) for n, it in prob.items.items()}
        self.var = {n: np.diag(it.cov)[it.data_mask] for n, it in prob.items.items()}
        self.pools, self.index = {}, None

    def matmul_host(self, A, X):
        return X @ A.T

    def set_mock_pool(self, name, pool):
        self.pools[name] = np.asarray(pool)

    def set_mock_index(self, index=None):
        self.index = None if index is None else np.asarray(index)

    def set_invcov(self, name, m):
        raise AssertionError('no rescaled covariance in this test')

    def eval(self, theta, want_model=False):
        cols = [self.low.slot[n] for n in NAMES]
        d = theta[:, cols] - self.low.theta0[cols]
        chi2 = np.zeros(theta.shape[0])
        for name in self.prob.items:
            model = self.fid[name][None, :] + d @ self.templates[name]
            data = self.pools[name][self.index] if self.index is not None else self.fid[name][None, :]
            chi2 += (((data - model)**2) / self.var[name][None, :]).sum(axis=1)
        return chi2, np.zeros(theta.shape[0], dtype=np.int32), None


def _make_vega(config, device):
    prob = _problem()
    eng = _StandInEngine(prob)
    full = {n: np.zeros(it.data_vec.size) for n, it in prob.items.items()}
    for n, it in prob.items.items():
        full[n][it.data_mask] = eng.fid[n]
    return SimpleNamespace(problem=prob, engine=eng, main_config=prob.main_config, sample_params=prob.sample_params,
                           _use_global_cov=False, get_fiducial_for_monte_carlo=lambda print_func=print: full)


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port))
    sys.path.insert(0, str(REPO / 'scripts'))
    sys.path.insert(0, str(REPO / 'tests'))
    import torch.distributed as dist
    import run_mc_sharded
    mc, res, block = run_mc_sharded.run('unused.ini', output_dir=out_dir, make_vega=_make_vega, backend='gloo')
    assert block == ((0, 3) if rank == 0 else (3, 5))
    assert res.values.shape == (3, 2) and res.is_valid.all()
    np.save(Path(out_dir) / f'values_{rank}.npy', res.values)
    dist.destroy_process_group()


def test_sharded_monte_carlo_launcher_gloo_world2(tmp_path):
    from vega_amd import fitslite
    from vega_amd.montecarlo import create_mocks
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    vega = _make_vega(None, 0)
    fid = vega.get_fiducial_for_monte_carlo()
    for rank in range(2):
        hdul = fitslite.open(tmp_path / f'monte_carlo_{rank}.fits')
        tabs = {h.header['EXTNAME'].upper(): h for h in hdul[1:]}
        assert set(tabs) == {'BESTFIT', 'FITINFO', 'MOCKS'}
        want = create_mocks(vega.problem, fid, 3, seed=SEED + rank)          # ceil(5 / 2) mocks from seed + rank
        for name, item in vega.problem.items.items():
            col = np.asarray(tabs['MOCKS'].data[name])
            assert col.shape == (3, item.data_vec.size)
            np.testing.assert_allclose(col[:, item.data_mask], want[name], rtol=0, atol=1e-15)
        values = np.asarray(tabs['BESTFIT'].data['values'])
        assert [str(n).strip() for n in tabs['BESTFIT'].data['names']] == NAMES
        np.testing.assert_allclose(values.T, np.load(tmp_path / f'values_{rank}.npy'), rtol=1e-12)
        # the linear stand-in model has its minimum within a few sigma of the truth (the configured ap, at)
        errors = np.asarray(tabs['BESTFIT'].data['errors'])
        truth = np.array([vega.problem.params[n] for n in NAMES])[:, None]
        assert np.abs((values - truth) / errors).max() < 6
        assert np.asarray(tabs['FITINFO'].data['valid_minima']).all()


def test_launcher_refuses_a_config_without_monte_carlo(tmp_path):
    sys.path.insert(0, str(REPO / 'scripts'))
    import run_mc_sharded

    def make(config, device):
        vega = _make_vega(config, device)
        vega.main_config['control'] = {'run_montecarlo': 'False'}
        return vega
    os.environ.update(RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    with pytest.raises(ValueError, match='run_montecarlo'):
        run_mc_sharded.run('unused.ini', output_dir=str(tmp_path), make_vega=make)


def test_launcher_fits_the_mocks_of_a_file(tmp_path):
    """`--fit-mocks` = the reference's bin/run_vega_mc_fits_mpi.py (:112-163): HDU MOCKS / column 'global' of `[control]
    mc_mocks`, cut by the four slice bounds, this rank's contiguous share fitted and written (one rank here; the share
    arithmetic has its own test, the fits against the drawn mocks' run on the GPU: tests/test_round2_gpu.py)."""
    sys.path.insert(0, str(REPO / 'scripts'))
    import run_mc_sharded
    from vega_amd import fitslite
    base = _make_vega(None, 0)
    prob = base.problem
    sizes = [it.data_size for it in prob.items.values()]
    rng = np.random.default_rng(5)
    mocks = np.concatenate([base.engine.fid[n][None, :] + 1e-4 * rng.standard_normal((4, s))
                            for n, s in zip(prob.items, sizes)], axis=1)
    longer = np.concatenate([np.zeros((4, 2)), mocks[:, :sizes[0]], np.ones((4, 5)), mocks[:, sizes[0]:]], axis=1)
    path = tmp_path / 'mocks.fits'
    fitslite.write_tables(str(path), [('MOCKS', [('global', f'{longer.shape[1]}D', longer)])])

    def make(config, device):
        vega = _make_vega(config, device)
        vega.problem.global_cov = np.eye(sum(sizes))
        vega.problem.search_dirs = [tmp_path]
        vega._use_global_cov = True
        vega.freeze_metals = lambda: None
        c = vega.main_config['control']
        c['mc_mocks'] = str(path)
        c['slice_start1'], c['slice_end1'] = '2', str(2 + sizes[0])
        c['slice_start2'], c['slice_end2'] = str(2 + sizes[0] + 5), str(longer.shape[1])
        return vega
    lines = []
    mc, res, block = run_mc_sharded.run('unused.ini', output_dir=tmp_path / 'out', make_vega=make, print_func=lines.append,
                                        fit_mocks=True)
    assert block == (0, 4) and res.values.shape == (4, 2) and res.is_valid.all()
    np.testing.assert_array_equal(mc.mc_mocks['global'], mocks)
    tabs = {h.header['EXTNAME'].upper(): h for h in fitslite.open(tmp_path / 'out' / 'monte_carlo.fits')[1:]}
    np.testing.assert_array_equal(np.asarray(tabs['MOCKS'].data['global']), mocks)
    np.testing.assert_allclose(np.asarray(tabs['BESTFIT'].data['values']).T, res.values, rtol=1e-12)
    assert any('4 mocks of' in line and '4 valid fits' in line for line in lines)
