import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa
import bench  # noqa
which = sys.argv[1]
prob = bench.build_problem('joint')
if which == 'mc':
    print(bench.monte_carlo_fits(prob, 0)['fits_per_s'])
elif which == 'single':
    print(bench.single_point_latency(0)['us_per_eval'])
elif which == 'micro':
    from vega_amd import VegaInterface
    v = VegaInterface(None, problem=prob, max_batch=256)
    print(bench.distortion_microbench(v.engine, torch)['achieved'])
print(bench.metals_throughput(0)['static_basis'])
