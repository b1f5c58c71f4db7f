"""Walker sharding over the GPUs of one node: one process per GPU, one collective per batch.

The reference parallelises only over independent work items (Monte-Carlo mocks / sampler points) with
mpi4py rank/size arithmetic and no payload-carrying collective (reference bin/run_vega_mc_mpi.py:17-25,
:54-65: ``num_local_mc = ceil(N / size)``, ``seed + rank``).  Here walkers are block-partitioned the same
way - rank r owns the contiguous block [r * ceil(n / size), (r + 1) * ceil(n / size)) - every rank evaluates its
block on its own engine, and the chi2 values are exchanged with ONE ``all_gather`` per call (RCCL over xGMI when the
tensors live on GPUs, gloo on CPU for tests).

Two forms of the same partition:

* ``chi2_sharded(evaluate, theta)`` with ``theta`` a CUDA tensor: ``evaluate`` maps a device block to a device vector
  (``VegaInterface.chi2_batch_device``), the gather runs on device buffers and the result stays on the device - no
  host round trip anywhere;
* with ``theta`` a NumPy array: ``evaluate`` is the host entry (``VegaInterface.chi2_batch``); the gather buffer lives
  on ``device`` (CPU for gloo) and the result comes back as NumPy - one copy at the very end.

Every rank evaluates its block in calls of the engine's own ``max_batch``, so a walker's chi2 does not depend on the
number of ranks: the engine's arithmetic is a pure function of (walker, batch-size class), never of timing or
process-wide state (include/vegamx.h).
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

    ``evaluate(theta_block) -> chi2_block`` is the local engine call; tensors in -> tensor out (device-resident),
    NumPy in -> NumPy out.  Without an initialised process group the call is ``evaluate(theta)``.
    """
    import torch
    import torch.distributed as dist

    on_device = isinstance(theta, torch.Tensor)
    if not on_device:
        theta = np.atleast_2d(theta)
    n = theta.shape[0]
    if not (dist.is_available() and dist.is_initialized()):
        return evaluate(theta) if on_device else np.asarray(evaluate(theta))
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = math.ceil(n / world)
    lo, hi = shard_bounds(n, world, rank)
    if on_device:
        # RCCL gathers device buffers in place; a gloo group (CPU tests, or several ranks sharing one GPU) stages them
        device = theta.device if dist.get_backend(group) == 'nccl' else torch.device('cpu')
    local = torch.zeros(per, dtype=torch.float64, device=device)
    if hi > lo:
        block = evaluate(theta[lo:hi])
        if on_device:
            local[:hi - lo].copy_(block)
        else:
            local[:hi - lo] = torch.as_tensor(np.asarray(block), dtype=torch.float64)
    out = torch.empty(world * per, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(out, local, group=group)
    return out[:n].to(theta.device) if on_device else out[:n].cpu().numpy()
