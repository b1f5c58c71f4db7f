"""Cost of the per-kernel event timing: graph replay vs eager vs eager + events (one class / all classes)."""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402
import bench  # noqa: E402
from vega_amd import VegaInterface, synthetic  # noqa: E402

B = 256
prob = bench.build_problem('joint')
dev = torch.device('cuda', 0)
vega = VegaInterface(None, problem=prob, max_batch=B)
eng = vega.engine
eng.set_constant_nl_hint(True)
pools = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=100 + i)).to(dev) for i in range(8)]
out = torch.zeros(B, dtype=torch.float64, device=dev)

def run(label, n=40):
    for i in range(5):
        eng.eval_device(pools[i % 8].data_ptr(), B, out.data_ptr())
    eng.sync(); eng.timings(reset=True)
    t0 = time.perf_counter()
    for i in range(n):
        eng.eval_device(pools[i % 8].data_ptr(), B, out.data_ptr())
    eng.sync()
    dt = time.perf_counter() - t0
    print(f'{label}: {dt / n * 1e3:.4f} ms/step', {k: round(v[0] / v[1], 4) for k, v in eng.timings(reset=True).items() if v[1]}, flush=True)

run('graph')
eng.set_profiling(True); eng.set_profiling_classes([])
run('eager, no events')
eng.set_profiling_classes(['distortion_product'])
run('eager, distortion events')
eng.set_profiling_classes(['pk_multipoles'])
run('eager, pk events')
eng.set_profiling(True)
run('eager, all events')
eng.set_profiling(False)
run('graph again')
vega.close()
