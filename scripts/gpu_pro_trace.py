"""Where k_prologue's time goes (experiment build -DVMX_EXP_PRO_TRACE, VEGAMX_LIBRARY pointing at it): per block, 100 MHz stamps
at entry / after the staging barrier / after the scalars / after the window atomics / before and after the stores."""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
out = REPO / 'gpurun_out' / 'pro_trace.bin'
os.environ['VMX_PK_TRACE'] = str(out)
os.environ['VMX_EXP_NO_PK'] = '1'
import numpy as np
from vega_amd import VegaInterface, synthetic
B = int(os.environ.get('PKB', '256'))
vega = VegaInterface('configs/joint/main.ini', search_dirs=[REPO / 'tests' / 'golden'], max_batch=B)
eng = vega.engine
theta = synthetic.walkers(eng.low.theta0, eng.names, B, seed=3,
                          varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO',
                                  'drp_QSO', 'bias_hcd', 'beta_hcd', 'L0_hcd'])
for _ in range(6):
    eng.eval(theta)
eng.sync()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t0 = t[:, 0][t[:, 0] > 0].min()
np.set_printoptions(linewidth=200, precision=2, suppress=True)
print('k_prologue, B = 256: 4 chunks of 64 walkers x (n_pipe P(k,mu)-half slots, n_pipe xi-half slots, 1 walker-level slot), one single-wave block each')
print('block | entry (us after the first block) | us since entry of: staging barrier passed, P(k,mu) half: scalars done, xi half: window done, '
      'xi half: all formed, xi half: stored, walker slot: done, staging loads landed')
for i, row in enumerate(t):
    if row[0] == 0: continue
    rel = [(x - row[0]) / 100.0 if x >= row[0] else float("nan") for x in row[1:8]]       # (older stamps: another launch's)
    print(i, f'{(row[0] - t0) / 100.0:7.2f}', np.array(rel))
