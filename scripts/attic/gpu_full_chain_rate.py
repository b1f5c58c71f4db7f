"""Full-chain (model requested) throughput of the bench workload at several batch sizes."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
torch.cuda.init()
import bench
from vega_amd import VegaInterface, synthetic
prob = bench.build_problem('joint')
for B in (64, 256, 1024):
    vega = VegaInterface(None, problem=prob, max_batch=B)
    eng = vega.engine
    eng.set_constant_nl_hint(True, gaussian=True)
    th = torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=1)).cuda()
    out = torch.zeros(B, dtype=torch.float64, device='cuda')
    model = torch.zeros(B, eng.model_size, dtype=torch.float64, device='cuda')
    for _ in range(30):
        eng.eval_device(th.data_ptr(), B, out.data_ptr(), model.data_ptr())
    eng.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            eng.eval_device(th.data_ptr(), B, out.data_ptr(), model.data_ptr())
        eng.sync()
        best = min(best, time.perf_counter() - t0)
    eng.set_profiling(True); eng.timings(reset=True)
    for _ in range(5):
        eng.eval_device(th.data_ptr(), B, out.data_ptr(), model.data_ptr())
    eng.sync()
    t = eng.timings(reset=True)
    print(B, round(B * 20 / best), {k: round(v[0] / v[1] * 1e3, 1) for k, v in t.items() if v[1] and k in ('distortion_product', 'invcov_product')}, float(out[0]), flush=True)
    vega.close()
