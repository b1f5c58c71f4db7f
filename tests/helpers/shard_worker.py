"""One rank of the two-process GPU test (tests/test_sharded_gpu.py): a fresh process that builds a REAL engine on GPU 0,
joins a gloo group and runs the sharded entry points - `parallel.chi2_sharded` over joint + metals walkers (host and
device-resident form) and `scripts/run_mc_sharded.run` over Monte-Carlo mocks.  Started by tests/helpers/spawn_server.py.

    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/helpers/shard_worker.py OUT_DIR N_WALKERS MC_DIR
"""
import os
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / 'scripts'))
sys.path.insert(0, str(REPO / 'tests'))

WALKER_SEED = 424242
MAX_BATCH = 1024


def walkers(vega, n):
    from vega_amd import synthetic
    from bench import VARIED
    eng = vega.engine
    return synthetic.walkers(eng.low.theta0, eng.names, n, varied=VARIED, seed=WALKER_SEED)


def main():
    out_dir, n_walkers, mc_dir = Path(sys.argv[1]), int(sys.argv[2]), Path(sys.argv[3])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    import torch
    torch.cuda.init()               # torch's HIP runtime first, then libvegamx.so (tests/conftest.py)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from bench import build_problem
    from vega_amd import VegaInterface
    from vega_amd.parallel import chi2_sharded
    import run_mc_sharded

    vega = VegaInterface(None, problem=build_problem('joint_metals'), max_batch=MAX_BATCH, device=0)
    theta = walkers(vega, n_walkers)
    host = chi2_sharded(vega.chi2_batch, theta)
    # device-resident walkers: the caller vouches for what the host entry detects by itself - these walkers share their
    # Arinyo / smoothing parameters (a violation would be flagged per walker), so both entries take the same kernels
    vega.engine.set_constant_nl_hint(True, gaussian=True)
    dev = chi2_sharded(vega.chi2_batch_device, torch.from_numpy(theta).to('cuda:0'))
    np.save(out_dir / f'chi2_host_{rank}.npy', host)
    np.save(out_dir / f'chi2_device_{rank}.npy', dev.cpu().numpy())
    vega.close()

    os.environ['LOCAL_RANK'] = '0'          # both ranks share the one GPU of the box
    from conftest import GOLDEN
    mc, res, block = run_mc_sharded.run('configs/mc/main.ini', output_dir=str(out_dir / 'monte_carlo'),
                                        search_dirs=[mc_dir, GOLDEN], max_batch=128, backend='gloo')
    np.save(out_dir / f'mc_values_{rank}.npy', res.values)
    np.save(out_dir / f'mc_fval_{rank}.npy', res.fval)
    np.save(out_dir / f'mc_block_{rank}.npy', np.array(block))
    mc.vega.close()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
