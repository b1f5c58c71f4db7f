"""Kernel table of the full chain (a model is asked for) at B = 256: rocprofv3 --kernel-trace --stats -- python3 this"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from vega_amd import VegaInterface, synthetic
prob = bench.build_problem('joint')
vega = VegaInterface(None, problem=prob, max_batch=256, device=0)
eng = vega.engine
eng.set_constant_nl_hint(True, gaussian=True)
pool = torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, 256, varied=bench.VARIED)).to('cuda:0')
chi2 = torch.zeros(256, dtype=torch.float64, device='cuda:0')
model = torch.zeros(256, eng.model_size, dtype=torch.float64, device='cuda:0')
for _ in range(200):
    eng.eval_device(pool.data_ptr(), 256, chi2.data_ptr(), model.data_ptr())
eng.sync()
import time
t0 = time.perf_counter()
for _ in range(100):
    eng.eval_device(pool.data_ptr(), 256, chi2.data_ptr(), model.data_ptr())
eng.sync()
print('ms per step', (time.perf_counter() - t0) * 10)
