"""world_size-2 gloo test (CPU) of the walker sharding + single all_gather used for N > 1 GPUs."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, out_dir):
    import torch.distributed as dist
    from vega_amd.parallel import chi2_sharded, shard_bounds
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    rng = np.random.default_rng(7)
    theta = rng.standard_normal((n, 5))
    calls = []

    def evaluate(block):            # stand-in for the per-GPU engine: any per-walker function
        calls.append(block.shape[0])
        return (block**2).sum(axis=1) + 3.0

    full = chi2_sharded(evaluate, theta)
    lo, hi = shard_bounds(n, world, rank)
    assert calls == ([hi - lo] if hi > lo else [])
    np.save(os.path.join(out_dir, f'rank{rank}.npy'), full)
    # tensors in -> tensor out (the device-resident form; CPU tensors under gloo)
    import torch
    as_tensor = chi2_sharded(lambda block: (block**2).sum(dim=1) + 3.0, torch.from_numpy(theta))
    assert isinstance(as_tensor, torch.Tensor)
    np.save(os.path.join(out_dir, f'rank{rank}_tensor.npy'), as_tensor.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize('n', [8, 7, 1])
def test_sharded_chi2_gloo_world2(tmp_path, n):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, n, str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(7)
    theta = rng.standard_normal((n, 5))
    expect = (theta**2).sum(axis=1) + 3.0
    for rank in range(2):
        got = np.load(tmp_path / f'rank{rank}.npy')
        np.testing.assert_array_equal(got, expect)
        np.testing.assert_array_equal(np.load(tmp_path / f'rank{rank}_tensor.npy'), expect)


def test_shard_bounds_cover_everything():
    from vega_amd.parallel import shard_bounds
    for n in (0, 1, 7, 8, 4096, 8192):
        for world in (1, 2, 3, 8):
            blocks = [shard_bounds(n, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(blocks[:-1], blocks[1:]))
