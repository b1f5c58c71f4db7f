"""Monte-Carlo fit throughput of the two minimisers (development aid)."""
import json, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
torch.cuda.init()
import bench
prob = bench.build_problem('joint')
out = bench.monte_carlo_fits(prob, 0, n_mocks=int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
print(json.dumps(out, indent=1))
