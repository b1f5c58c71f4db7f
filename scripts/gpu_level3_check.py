"""Level-3 tables (vmx_set_hcd_level3) against the node rule / plain loop of k_pk_tab2 on the bench workload: chi2 and P_ell of
the same walkers before the tables exist (first two evaluations) and after, walkers outside the box, timing of the step.
    python3 scripts/gpu_level3_check.py [B]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
import bench
from vega_amd import VegaInterface, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = bench.build_problem('joint')
dev = torch.device('cuda', 0)
res = {}
for label, hw in (('off', '0'), ('on', '0.125')):
    os.environ['VEGA_AMD_LEVEL3'] = hw
    vega = VegaInterface(None, problem=prob, max_batch=B, device=0)
    eng = vega.engine
    eng.set_constant_nl_hint(True, gaussian=True)
    theta = synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=synthetic.SEED)
    if label == 'on':
        theta[3, eng.low.slot['L0_hcd']] *= 1.4         # outside the box: this walker keeps k_pk_tab2
        theta[5, eng.low.slot['L0_hcd']] = eng.low.theta0[eng.low.slot['L0_hcd']]      # on the centre node
    else:
        theta[3, eng.low.slot['L0_hcd']] *= 1.4
        theta[5, eng.low.slot['L0_hcd']] = eng.low.theta0[eng.low.slot['L0_hcd']]
    d_theta = torch.from_numpy(theta).to(dev)
    out = torch.zeros(B, dtype=torch.float64, device=dev)
    chis, served = [], []
    for it in range(4):
        eng.eval_device(d_theta.data_ptr(), B, out.data_ptr())
        eng.sync()
        chis.append(out.cpu().numpy().copy())
        served.append(eng.level3_served() if hw != '0' else 0)
    pl = eng.pk_multipoles(B)
    for _ in range(50):
        eng.eval_device(d_theta.data_ptr(), B, out.data_ptr())
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(100):
        eng.eval_device(d_theta.data_ptr(), B, out.data_ptr())
    eng.sync()
    dt = (time.perf_counter() - t0) / 100
    eng.set_profiling(True)
    for _ in range(20):
        eng.eval_device(d_theta.data_ptr(), B, out.data_ptr())
    eng.sync()
    tm = eng.timings(reset=True)
    eng.set_profiling(False)
    res[label] = (chis, pl, dt, served, {k: v[0] / max(v[1], 1) * 1e3 for k, v in tm.items() if v[1]})
    print(label, 'served', served, f'{dt * 1e6:.1f} us / step = {B / dt:.0f} evals/s', {k: round(v, 1) for k, v in res[label][4].items()}, flush=True)
    vega.close()
off, on = res['off'], res['on']
print('chi2 on[0] vs off[0] (both k_pk_tab2):', float(np.max(np.abs(on[0][0] - off[0][0]) / np.abs(off[0][0]))))
print('chi2 level 3 (4th evaluation) vs k_pk_tab2:', float(np.max(np.abs(on[0][3] - off[0][3]) / np.abs(off[0][3]))))
print('  walker 3 (outside the box):', float(abs(on[0][3][3] - off[0][3][3]) / abs(off[0][3][3])), ' walker 5 (centre):', float(abs(on[0][3][5] - off[0][3][5]) / abs(off[0][3][5])))
worst = 0.
for pid in on[1]:
    a, b = on[1][pid], off[1][pid]
    scale = np.abs(b).max(axis=2, keepdims=True)
    worst = max(worst, float((np.abs(a - b) / np.where(scale > 0, scale, 1)).max()))
print('P_ell: max |level 3 - tab2| / max |P_ell| per (walker, ell):', worst)
