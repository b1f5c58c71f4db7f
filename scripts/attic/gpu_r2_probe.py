"""Round-2 probe: B = 1 latency chain (configs[1]) and the metals workload, per-kernel."""
import json, sys, os
sys.path.insert(0, '.')
import torch
torch.cuda.init()
import bench
what = sys.argv[1:] or ['single', 'metals']
out = {}
if 'single' in what:
    out['single'] = bench.single_point_latency(0)
if 'metals' in what:
    out['metals'] = bench.metals_throughput(0)
print(json.dumps(out, indent=1))
