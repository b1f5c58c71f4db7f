"""The COEFMOD = 2 leg of bench.py alone: Q' against the factored form on model grids finer than the data grids."""
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch
torch.cuda.init()
import bench

print(json.dumps(bench.coefmod_throughput(0), indent=1))
