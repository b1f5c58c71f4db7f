"""Which leg of bench.py slows the Monte-Carlo leg that follows it?  python3 scripts/attic/gpu_mc_after_legs.py <legs>
legs: comma-separated subset of distortion,single,csr (run in that order before monte_carlo_fits)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import torch
import bench
from vega_amd import VegaInterface

legs = [x for x in (sys.argv[1] if len(sys.argv) > 1 else '').split(',') if x]
prob = bench.build_problem('joint')
vega = VegaInterface(None, problem=prob, max_batch=256, device=0)
if 'distortion' in legs:
    bench.distortion_microbench(vega.engine, torch)
if 'single' in legs:
    bench.single_point_latency(0)
if 'csr' in legs:
    bench.distortion_csr(0)
out = bench.monte_carlo_fits(prob, 0)
print(legs, round(out['migrad']['fits_per_s']), out['migrad']['driver_seconds'])
