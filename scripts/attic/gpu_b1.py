import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench
from vega_amd import VegaInterface
prob = bench.build_problem('auto')
for item in prob.items.values():
    item.core.xi.ell_max = 4
vega = VegaInterface(None, problem=prob, max_batch=1)
theta = vega.engine.theta_from_params()[None, :]
for _ in range(10):
    vega.engine.eval(theta)
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(200):
        vega.engine.eval(theta)
    print('us/eval', (time.perf_counter() - t0) / 200 * 1e6)
vega.engine.set_profiling(True)
for _ in range(5):
    vega.engine.eval(theta)
vega.engine.timings()
for _ in range(50):
    vega.engine.eval(theta)
print({k: round(v[0] / v[1] * 1e3, 2) for k, v in vega.engine.timings().items() if v[1]})
