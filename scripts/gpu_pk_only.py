"""Run the joint engine at B=256 a few times (profiling target)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from vega_amd import VegaInterface, synthetic  # noqa: E402
vega = VegaInterface('configs/joint/main.ini', search_dirs=[REPO / 'tests' / 'golden'], max_batch=256)
eng = vega.engine
theta = synthetic.walkers(eng.low.theta0, eng.names, 256, seed=3,
                          varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO',
                                  'drp_QSO', 'bias_hcd', 'beta_hcd', 'L0_hcd'])
for _ in range(3):
    eng.eval(theta)
