#!/usr/bin/env python
"""Monte-Carlo over mocks, one process per GPU: the reference's ``bin/run_vega_mc_mpi.py`` (:17-71) with torchrun in
place of mpirun; with ``--fit-mocks`` the reference's ``bin/run_vega_mc_fits_mpi.py`` (:112-163): the global mocks of the
file ``[control] mc_mocks`` (optionally cut to two slices) are fitted, a contiguous share per rank.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29555 \
        scripts/run_mc_sharded.py main.ini --output-dir out/monte_carlo

Rank r takes ceil(num_mc_mocks / world) mocks drawn from ``mc_seed + r`` (reference :52-65), fits them all in
lock-step on GPU ``LOCAL_RANK`` and writes ``monte_carlo_<r>.fits`` (``monte_carlo.fits`` with one rank), the
reference's file layout (vega/output.py:442-520).  Mocks are independent: there is NO data-path collective; the process
group (RCCL when the ranks hold GPUs) only carries the start / end barriers of the reference's ``print_func`` and one
gather of per-rank fit counts for the summary line.  The device is chosen before anything touches the GPU.
"""
import argparse
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))


def build_library_once():
    """Every rank needs libvegamx.so; only one compiles it (a file lock), the others wait on the lock - not inside a
    collective."""
    import fcntl
    import __graft_entry__ as entry
    lock_path = REPO / 'vega_amd' / '.build.lock'
    with open(lock_path, 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            entry.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def run(config, output_dir=None, search_dirs=(), max_batch=1024, backend=None, make_vega=None, print_func=None,
        fit_mocks=None):
    """The body of the launcher; ``make_vega(config, device)`` may replace the interface (CPU tests of the sharding
    logic).  Returns (MonteCarlo driver, FitResult or None, (lo, hi) block of this rank)."""
    import numpy as np
    import torch.distributed as dist
    from vega_amd.montecarlo import run_monte_carlo_sharded

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_dist = world > 1
    if use_dist and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29555')
        if backend is None:
            backend = 'nccl' if make_vega is None else 'gloo'
        kw = {}
        if backend == 'nccl':
            import torch
            torch.cuda.set_device(local_rank)
            kw['device_id'] = torch.device('cuda', local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    def say(message):
        if rank == 0:
            print(message)
        sys.stdout.flush()
        if use_dist:
            dist.barrier()
    say = print_func or say

    say('Initializing Vega')
    if make_vega is None:
        from vega_amd import VegaInterface
        vega = VegaInterface(config, search_dirs=search_dirs, max_batch=max_batch, device=local_rank)
    else:
        vega = make_vega(config, local_rank)
    control = vega.main_config['control'] if 'control' in vega.main_config else None
    if control is None or not control.getboolean('run_montecarlo', False) or vega.problem.mc_config is None:
        raise ValueError('Warning: You called "run_mc_sharded.py" without asking for monte carlo. Add '
                         '"run_montecarlo = True" to the "[control]" section.')
    say('Finished initializing Vega')
    if fit_mocks:
        # the reference's bin/run_vega_mc_fits_mpi.py (:112-163): fit the global mocks of a file (HDU MOCKS, column
        # 'global'; `[control] mc_mocks`, optionally cut by slice_start1 / slice_end1 / slice_start2 / slice_end2), a contiguous
        # share per rank
        from vega_amd.montecarlo import fit_mocks_sharded
        from vega_amd.tables import find_file, read_tables
        if control.get('mc_mocks', None) is None:
            raise ValueError('--fit-mocks: `[control] mc_mocks` names the file with the mocks')
        vega.monte_carlo = True
        table = [t for t in read_tables(find_file(control.get('mc_mocks'), vega.problem.search_dirs))
                 if str(t.header.get('EXTNAME', '')).strip().upper() == 'MOCKS'][0]
        mocks = np.asarray(table.data['global'], dtype=float)
        slices = tuple(control.getint(key, None) for key in ('slice_start1', 'slice_end1', 'slice_start2', 'slice_end2'))
        if output_dir is None:
            out = vega.main_config['output'].get('mc_output', None) if 'output' in vega.main_config else None
            output_dir = out if out is not None else Path(vega.main_config['output']['filename']).parent / 'monte_carlo'
        t0 = time.perf_counter()
        mc, res, block = fit_mocks_sharded(vega, mocks, slices, rank=rank, world_size=world, output_dir=output_dir)
        dt = time.perf_counter() - t0
        n_valid = int(np.sum(res.is_valid)) if res is not None else 0
        counts = [(block[1] - block[0], n_valid, dt)]
        if use_dist:
            counts = [None] * world
            dist.all_gather_object(counts, (block[1] - block[0], n_valid, dt))
        say(f'{sum(c[0] for c in counts)} mocks of {control.get("mc_mocks")} on {world} rank(s): {sum(c[1] for c in counts)} '
            f'valid fits, slowest rank {max(c[2] for c in counts):.2f} s; results under {output_dir}')
        return mc, res, block
    fiducial_model = vega.get_fiducial_for_monte_carlo(print_func=say)
    if control.getboolean('forecast', False):
        raise ValueError('You asked to run a forecast. Use a single process instead.')
    seed = control.getint('mc_seed', 0)
    num_mc_mocks = control.getint('num_mc_mocks', 1)
    run_mc_fits = control.getboolean('run_mc_fits', True)
    if output_dir is None:
        out = vega.main_config['output'].get('mc_output', None) if 'output' in vega.main_config else None
        output_dir = out if out is not None else Path(vega.main_config['output']['filename']).parent / 'monte_carlo'
    t0 = time.perf_counter()
    mc, res, block = run_monte_carlo_sharded(vega, fiducial_model, num_mc_mocks, seed=seed, rank=rank,
                                             world_size=world, output_dir=output_dir, run_mc_fits=run_mc_fits)
    dt = time.perf_counter() - t0
    n_local = -(-num_mc_mocks // world)
    n_valid = int(np.sum(res.is_valid)) if res is not None else 0
    if use_dist:
        counts = [None] * world
        dist.all_gather_object(counts, (n_local, n_valid, dt))
    else:
        counts = [(n_local, n_valid, dt)]
    say(f'{sum(c[0] for c in counts)} mocks on {world} rank(s): {sum(c[1] for c in counts)} valid fits, '
        f'slowest rank {max(c[2] for c in counts):.2f} s; results under {output_dir}')
    return mc, res, block


def main(argv=None):
    ap = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter,
                                 description='Run the Vega Monte Carlo sharded over the GPUs of a node.')
    ap.add_argument('config', type=str, help='main.ini')
    ap.add_argument('--output-dir', default=None, help='directory of monte_carlo_<rank>.fits (default: next to [output] filename)')
    ap.add_argument('--search-dir', action='append', default=[], help='extra directory for relative paths in the configs')
    ap.add_argument('--max-batch', type=int, default=1024, help='walkers per engine call')
    ap.add_argument('--fit-mocks', action='store_true',
                    help="fit the global mocks of `[control] mc_mocks` instead of drawing mocks (the reference's "
                         'bin/run_vega_mc_fits_mpi.py)')
    args = ap.parse_args(argv)
    build_library_once()
    import torch.distributed as dist
    try:
        run(args.config, args.output_dir, args.search_dir, args.max_batch, fit_mocks=args.fit_mocks)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == '__main__':
    main()
