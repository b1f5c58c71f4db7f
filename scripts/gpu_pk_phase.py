"""k_pk_tab2 block phases at B=256 (experiment build -DVMX_EXP_PK_PHASE via VEGAMX_LIBRARY): set-up, mu loop, tail - by round."""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
out = REPO / 'gpurun_out' / 'pk_phase.bin'
os.environ['VMX_PK_TRACE'] = str(out)
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
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0; end = (t[:, 1] - t0) / 100.0
setup = (t[:, 2] - t[:, 0]) / 100.0; loop = (t[:, 3] - t[:, 2]) / 100.0; tail = (t[:, 1] - t[:, 3]) / 100.0
print('blocks', len(t), 'span', end.max())
for lo, hi in ((0, 5), (5, 40), (40, 70), (70, 200)):
    sel = (start >= lo) & (start < hi)
    if sel.sum():
        print(f'start in [{lo},{hi}) us: {sel.sum():5d} blocks  set-up {np.median(setup[sel]):6.2f} (p90 {np.percentile(setup[sel], 90):6.2f})  '
              f'mu loop {np.median(loop[sel]):6.2f} (p90 {np.percentile(loop[sel], 90):6.2f})  tail {np.median(tail[sel]):5.2f}')
