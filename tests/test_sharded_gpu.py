"""N > 1 on real engines: two fresh processes, each with its own engine on the GPU, a gloo group between them.

The walker partition of `vega_amd/parallel.py` and the mock partition of `scripts/run_mc_sharded.py` (the reference's
bin/run_vega_mc_mpi.py:17-71) must give, bit for bit, what one process gives: the engine's arithmetic is a pure function
of the walker and the batch-size class - no timing-dependent choice, no process-wide state (include/vegamx.h).
"""
import socket
import sys

import numpy as np
import pytest

from conftest import GOLDEN, REPO, mc_launcher_config, run_programs

pytestmark = pytest.mark.gpu

N_WALKERS = 4096
N_MOCKS = 16
MC_SEED = 3


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def test_two_processes_with_real_engines_match_one_process_bitwise(tmp_path):
    sys.path.insert(0, str(REPO / 'tests' / 'helpers'))
    sys.path.insert(0, str(REPO / 'scripts'))
    import shard_worker
    from bench import build_problem
    from vega_amd import VegaInterface
    from vega_amd.montecarlo import run_monte_carlo_sharded
    from vega_amd.parallel import shard_bounds
    mc_dir = tmp_path / 'mc'
    mc_dir.mkdir()
    mc_launcher_config(mc_dir, num_mocks=N_MOCKS, seed=MC_SEED)
    out_dir = tmp_path / 'ranks'
    out_dir.mkdir()
    port = _free_port()
    argv = [sys.executable, str(REPO / 'tests' / 'helpers' / 'shard_worker.py'), str(out_dir), str(N_WALKERS), str(mc_dir)]
    envs = [{'RANK': str(r), 'WORLD_SIZE': '2', 'MASTER_ADDR': '127.0.0.1', 'MASTER_PORT': str(port),
             'HSA_ENABLE_IPC_MODE_LEGACY': '0'} for r in range(2)]
    results = run_programs([argv, argv], envs, timeout=1000)
    for rank, (rc, output) in enumerate(results):
        assert rc == 0, f'rank {rank} failed:\n{output[-4000:]}'

    # one process, the same walkers: 4096 joint + metals walkers in calls of max_batch
    vega = VegaInterface(None, problem=build_problem('joint_metals'), max_batch=shard_worker.MAX_BATCH)
    theta = shard_worker.walkers(vega, N_WALKERS)
    want = vega.chi2_batch(theta)
    assert np.isfinite(want).all() and (want < 1e99).all()
    for rank in range(2):
        np.testing.assert_array_equal(np.load(out_dir / f'chi2_host_{rank}.npy'), want)
        np.testing.assert_array_equal(np.load(out_dir / f'chi2_device_{rank}.npy'), want)
    # ... and a repeat in this process, on a second engine of the same shape
    again = VegaInterface(None, problem=build_problem('joint_metals'), max_batch=shard_worker.MAX_BATCH)
    np.testing.assert_array_equal(again.chi2_batch(theta), want)
    again.close()
    vega.close()

    # Monte Carlo: rank r fitted ceil(16 / 2) mocks drawn from seed + r; this process does both shares in turn
    from run_mc_sharded import run      # noqa: F401  (the launcher the ranks went through)
    vega = VegaInterface('configs/mc/main.ini', search_dirs=[mc_dir, GOLDEN], max_batch=128)
    fid = vega.get_fiducial_for_monte_carlo(print_func=lambda message: None)
    for rank in range(2):
        assert tuple(np.load(out_dir / f'mc_block_{rank}.npy')) == shard_bounds(N_MOCKS, 2, rank)
        _, res, _ = run_monte_carlo_sharded(vega, fid, N_MOCKS, seed=MC_SEED, rank=rank, world_size=2)
        np.testing.assert_array_equal(np.load(out_dir / f'mc_values_{rank}.npy'), res.values)
        np.testing.assert_array_equal(np.load(out_dir / f'mc_fval_{rank}.npy'), res.fval)
        assert (out_dir / 'monte_carlo' / f'monte_carlo_{rank}.fits').is_file()
    vega.close()
