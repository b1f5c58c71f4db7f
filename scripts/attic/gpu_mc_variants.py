"""Why is bench.monte_carlo_fits slower than scripts/gpu_mc_timeline.py?  python3 scripts/attic/gpu_mc_variants.py <variant>"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
variant = sys.argv[1]
import os
if os.environ.get('PIN_NODE'):
    cpus = set()
    for part in Path(f"/sys/devices/system/node/node{os.environ['PIN_NODE']}/cpulist").read_text().strip().split(','):
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    os.sched_setaffinity(0, cpus)
if variant.startswith('torch_first'):
    import torch
import bench
from vega_amd import VegaInterface
prob = bench.build_problem('joint')
if variant.endswith('idle_engine'):
    vega = VegaInterface(None, problem=prob, max_batch=256, device=0)
if variant == 'used_engine':
    vega = VegaInterface(None, problem=prob, max_batch=256, device=0)
    vega.chi2()
if 'single' in variant:
    print(variant, bench.single_point_latency(0)['us_per_eval'])
    sys.exit(0)
out = bench.monte_carlo_fits(prob, 0)
print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa' in l}))
print(variant, round(out['migrad']['fits_per_s']), out['migrad']['driver_seconds'])
