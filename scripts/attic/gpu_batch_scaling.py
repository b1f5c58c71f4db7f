"""Throughput of the joint workload against the batch size (device entry, walkers resident in HBM)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
import bench
from vega_amd import VegaInterface, synthetic
prob = bench.build_problem('joint')
dev = torch.device('cuda', 0)
vega = VegaInterface(None, problem=prob, max_batch=4096)
eng = vega.engine
eng.set_constant_nl_hint(True, gaussian=True)
for B in (16, 32, 64, 128, 256, 512, 1024, 2048, 4096):
    pools = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=100 + i)).to(dev) for i in range(4)]
    out = torch.zeros(B, dtype=torch.float64, device=dev)
    for i in range(6):
        eng.eval_device(pools[i % 4].data_ptr(), B, out.data_ptr())
    eng.sync()
    n = max(10, min(100, 40000 // B))
    t0 = time.perf_counter()
    for i in range(n):
        eng.eval_device(pools[i % 4].data_ptr(), B, out.data_ptr())
    eng.sync()
    dt = (time.perf_counter() - t0) / n
    print(f'B = {B:5d}: {dt * 1e3:8.4f} ms / step  {B / dt:10.0f} evals/s  {dt / B * 1e6:6.3f} us / eval', flush=True)
vega.close()
