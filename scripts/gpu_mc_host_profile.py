"""Where the host time of a 1024-mock Monte-Carlo run goes outside the device-resident fit loop (cProfile + wall clocks)."""
import cProfile
import io
import pstats
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench
from vega_amd import VegaInterface

prob = bench.build_problem('joint')
vega = VegaInterface(None, problem=prob, max_batch=4096, device=0)
vega.chi2()
names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd']
limits = {'ap': (0.5, 1.5), 'at': (0.5, 1.5), 'bias_eta_LYA': (-2., 0.), 'beta_LYA': (0., 5.), 'beta_QSO': (0., 1.), 'bias_hcd': (-0.5, 0.)}
errors = {'ap': 0.01, 'at': 0.01, 'bias_eta_LYA': 0.01, 'beta_LYA': 0.1, 'beta_QSO': 0.1, 'bias_hcd': 0.01}
sample = {'limits': limits, 'values': {n: vega.params[n] for n in names}, 'errors': errors, 'fix': {n: False for n in names}}
vega.run_monte_carlo(num_mocks=128, seed=5, sample_params=sample)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for rep in range(2):
    t0 = time.perf_counter()
    res = vega.run_monte_carlo(num_mocks=n, seed=11, sample_params=sample)
    dt = time.perf_counter() - t0
    print(f'plain run {rep}: {dt:.4f} s = {n / dt:.0f} fits/s; device loop {res.driver_stats["seconds"]:.4f} s (setup {res.driver_stats["seconds_setup"]:.4f})')
pr = cProfile.Profile()
pr.enable()
res = vega.run_monte_carlo(num_mocks=n, seed=11, sample_params=sample)
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28)
print(s.getvalue()[:6000])
vega.close()
