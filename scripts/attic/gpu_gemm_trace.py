"""Block timeline of the FFTLog product at B=256 (VMX_GEMM_TRACE), or of the quadratic-form launch (TRACE_WHICH=VMX_QUAD_TRACE)."""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
out = REPO / 'gpurun_out' / 'gemm_trace.bin'
os.environ[os.environ.get('TRACE_WHICH', 'VMX_GEMM_TRACE')] = str(out)
import numpy as np
from vega_amd import VegaInterface, synthetic
B = int(os.environ.get('PKB', '256'))
vega = VegaInterface('configs/joint/main.ini', search_dirs=[REPO / 'tests' / 'golden'], max_batch=B)
eng = vega.engine
theta = synthetic.walkers(eng.low.theta0, eng.names, B, seed=3,
                          varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO',
                                  'drp_QSO', 'bias_hcd', 'beta_hcd', 'L0_hcd'])
eng.set_profiling(True)
for _ in range(4):
    eng.eval(theta)
eng.sync()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
us = (t - t0) / 100.0
print('live blocks', len(t), 'span', us[:, 3].max())
print('start quantiles', np.percentile(us[:, 0], [0, 50, 100]))
print('first-stage wait (start -> landed): median', np.median(us[:, 1] - us[:, 0]), 'p90', np.percentile(us[:, 1] - us[:, 0], 90))
print('K loop: median', np.median(us[:, 2] - us[:, 1]), 'p10', np.percentile(us[:, 2] - us[:, 1], 10), 'p90', np.percentile(us[:, 2] - us[:, 1], 90))
print('epilogue: median', np.median(us[:, 3] - us[:, 2]))
print('end quantiles', np.percentile(us[:, 3], [0, 10, 50, 90, 100]))
