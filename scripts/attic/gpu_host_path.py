"""Host entry (vmx_eval: host theta in, host chi2 out) against the device entry at B = 256, joint workload."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
import torch
import bench
from vega_amd import VegaInterface, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = bench.build_problem('joint')
vega = VegaInterface(None, problem=prob, max_batch=B)
eng = vega.engine
pools = [synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=100 + i) for i in range(4)]
for i in range(6):
    eng.eval(pools[i % 4])
t0 = time.perf_counter()
n = 100
for i in range(n):
    eng.eval(pools[i % 4])
dt = (time.perf_counter() - t0) / n
print(f'host entry: {dt * 1e3:.4f} ms/step  {B / dt:.0f} evals/s')
dev = torch.device('cuda', 0)
dp = [torch.from_numpy(p).to(dev) for p in pools]
out = torch.zeros(B, dtype=torch.float64, device=dev)
eng.set_constant_nl_hint(True, gaussian=True)
for i in range(6):
    eng.eval_device(dp[i % 4].data_ptr(), B, out.data_ptr())
eng.sync()
t0 = time.perf_counter()
for i in range(n):
    eng.eval_device(dp[i % 4].data_ptr(), B, out.data_ptr())
eng.sync()
dt2 = (time.perf_counter() - t0) / n
print(f'device entry: {dt2 * 1e3:.4f} ms/step  {B / dt2:.0f} evals/s')
t0 = time.perf_counter()
for i in range(n):
    eng.eval_device(dp[i % 4].data_ptr(), B, out.data_ptr())
    eng.sync()
dt3 = (time.perf_counter() - t0) / n
print(f'device entry, synchronised every step: {dt3 * 1e3:.4f} ms/step')
import os
os.environ['VMX_TRACE_HOST'] = '1'
