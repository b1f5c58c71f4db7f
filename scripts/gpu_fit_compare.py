"""Device-resident MIGRAD fits (vmx_fit_migrad) against the NumPy lock-step driver on the bench's Monte-Carlo workload:
per-fit call counts, minima, errors; then the timing of both.   python3 scripts/gpu_fit_compare.py [n_mocks]"""
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench
from vega_amd import VegaInterface

n_mocks = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prob = bench.build_problem('joint')
vega = VegaInterface(None, problem=prob, max_batch=4096, device=0)
vega.chi2()
names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd']
limits = {'ap': (0.5, 1.5), 'at': (0.5, 1.5), 'bias_eta_LYA': (-2., 0.), 'beta_LYA': (0., 5.), 'beta_QSO': (0., 1.), 'bias_hcd': (-0.5, 0.)}
errors = {'ap': 0.01, 'at': 0.01, 'bias_eta_LYA': 0.01, 'beta_LYA': 0.1, 'beta_QSO': 0.1, 'bias_hcd': 0.01}
sample = {'limits': limits, 'values': {n: vega.params[n] for n in names}, 'errors': errors, 'fix': {n: False for n in names}}
out = {}
for driver in ('python', 'device', 'device', 'streamed', 'streamed'):
    os.environ['VEGA_AMD_FIT_DRIVER'] = 'python' if driver == 'python' else 'device'
    os.environ['VEGA_AMD_STREAM_MOCKS'] = '1' if driver == 'streamed' else '0'
    t0 = time.perf_counter()
    res = vega.run_monte_carlo(num_mocks=n_mocks, seed=11, sample_params=sample)
    dt = time.perf_counter() - t0
    out[driver] = res
    print(driver, f'{dt:.3f} s', f'{n_mocks / dt:.0f} fits/s', 'evals', int(res.nfcn.sum()), 'valid', float(res.is_valid.mean()), flush=True)
    if getattr(res, 'driver_stats', None):
        print(res.driver_stats)
for label in ('device', 'streamed'):
  a, b = out['python'], out[label]
  print('---', label, 'against the NumPy driver')
  print('nfcn equal:', int((a.nfcn == b.nfcn).sum()), 'of', n_mocks, '| n_iter equal:', int((a.n_iter == b.n_iter).sum()))
  same = a.nfcn == b.nfcn
  print('max |dvalue| / error (same call count):', float((np.abs(a.values - b.values) / a.errors)[same].max()))
  print('max |dvalue| / error (all):', float((np.abs(a.values - b.values) / a.errors).max()))
  print('max rel dfval:', float((np.abs(a.fval - b.fval) / np.abs(a.fval)).max()), 'max rel derror', float((np.abs(a.errors - b.errors) / a.errors).max()))
vega.close()
