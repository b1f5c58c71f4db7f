import sys, time
from pathlib import Path
import numpy as np
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from vega_amd import VegaInterface, synthetic
from vega_amd.setup import build_problem
GOLD = REPO / 'tests' / 'golden'
for tag, drop in (('joint', 'lyalya_lyalya'), ('joint_metals', 'lyalya_lyalya')):
    prob = build_problem(f'configs/{tag}/main.ini', search_dirs=[GOLD])
    prob.items.pop(drop)
    vega = VegaInterface(None, problem=prob, max_batch=256)
    eng = vega.engine
    theta = synthetic.walkers(eng.low.theta0, eng.names, 256, seed=3, varied=['ap', 'at', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO', 'L0_hcd'])
    eng.set_profiling(True)
    for _ in range(3):
        eng.eval(theta)
    eng.timings()
    for _ in range(5):
        eng.eval(theta)
    print(tag, 'cross only', {k: round(v[0] / v[1], 4) for k, v in eng.timings().items() if v[1]})
    vega.close()
