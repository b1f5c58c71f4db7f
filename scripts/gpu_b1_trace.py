"""B = 1 chain (configs[1]): 300 chi2 evaluations through the host entry point, for a rocprofv3 kernel trace."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / 'tests'))
import torch
torch.cuda.init()
from conftest import config1_problem
from vega_amd import VegaInterface
vega = VegaInterface(None, problem=config1_problem(), max_batch=1)
theta = vega.engine.theta_from_params()[None, :]
for _ in range(20):
    vega.engine.eval(theta)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(100):
        vega.engine.eval(theta)
    print('us/eval', (time.perf_counter() - t0) / 100 * 1e6)

import ctypes as C, numpy as np
eng = vega.engine
th = np.ascontiguousarray(theta); chi2 = np.empty(1); st = np.empty(1, dtype=np.int32)
from vega_amd.engine import _dp, _ip
args = (eng._h, _dp(th), 1, _dp(chi2), None, _ip(st))
t0 = time.perf_counter()
for _ in range(200):
    eng.lib.vmx_eval(*args)
print('raw ctypes call us', (time.perf_counter() - t0) / 200 * 1e6)
vega.close()
