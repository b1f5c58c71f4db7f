"""Walker sharding over the GPUs of one node: one process per GPU, one collective per batch.

The reference parallelises only over independent work items (Monte-Carlo mocks / sampler points) with
mpi4py rank/size arithmetic and no payload-carrying collective (reference bin/run_vega_mc_mpi.py:17-25,
:54-65: ``num_local_mc = ceil(N / size)``, ``seed + rank``).  Here walkers are block-partitioned the same
way, every rank evaluates its block on its own engine, and the chi2 values are exchanged with ONE
``all_gather`` per batch (RCCL over xGMI when the tensors live on GPUs, gloo on CPU for tests).
"""
import math

import numpy as np


def shard_bounds(n_items, world_size, rank):
    """Contiguous block [lo, hi) of rank ``rank``: blocks of ceil(n / world) items, the last may be short."""
    per = math.ceil(n_items / world_size)
    lo = min(rank * per, n_items)
    return lo, min(lo + per, n_items)


def chi2_sharded(evaluate, theta, group=None, device=None):
    """Evaluate ``theta`` [n, P] across the ranks of ``group``; every rank returns the full chi2 [n].

    ``evaluate(theta_block) -> chi2_block`` is the local engine call (``VegaInterface.chi2_batch``).
    """
    import torch
    import torch.distributed as dist

    theta = np.atleast_2d(theta)
    n = theta.shape[0]
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(evaluate(theta))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = math.ceil(n / world)
    lo, hi = shard_bounds(n, world, rank)
    local = torch.zeros(per, dtype=torch.float64, device=device)
    if hi > lo:
        local[:hi - lo] = torch.as_tensor(np.asarray(evaluate(theta[lo:hi])), dtype=torch.float64, device=device)
    out = torch.empty(world * per, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out[:n].cpu().numpy()
