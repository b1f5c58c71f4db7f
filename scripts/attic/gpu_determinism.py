"""Bitwise repeatability probes (development aid): the same walkers through fresh engines, repeated calls, different
call orders, with the items forked or not."""
import os
import sys
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
torch.cuda.init()
from bench import build_problem, VARIED
from vega_amd import VegaInterface, synthetic

workload = sys.argv[1] if len(sys.argv) > 1 else 'joint_metals'
MB = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
prob = build_problem(workload)


def engine():
    return VegaInterface(None, problem=prob, max_batch=MB)


v = engine()
theta = synthetic.walkers(v.engine.low.theta0, v.engine.names, 4 * MB, varied=VARIED, seed=424242)
a = v.chi2_batch(theta)
b = v.chi2_batch(theta)
print('same engine, repeat:', int((a != b).sum()))
c = v.chi2_batch(theta[2 * MB:])
print('same engine, second half first:', int((a[2 * MB:] != c).sum()))
w = engine()
d = w.chi2_batch(theta[2 * MB:])
print('fresh engine, second half as its first calls:', int((a[2 * MB:] != d).sum()))
e = w.chi2_batch(theta)
print('fresh engine, all:', int((a != e).sum()))
dev = v.chi2_batch_device(torch.from_numpy(theta).cuda()).cpu().numpy()
print('device entry vs host entry:', int((a != dev).sum()))
for k in range(4):
    blk = slice(k * MB, (k + 1) * MB)
    print('  block', k, 'mismatches vs fresh engine', int((a[blk] != e[blk]).sum()), 'vs repeat', int((a[blk] != b[blk]).sum()))
v.close(); w.close()
