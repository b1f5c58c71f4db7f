"""configs[3] share (joint + metals, B = 512): evaluations / s with one and two batches in flight (development aid)."""
import json, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
torch.cuda.init()
import bench
print(json.dumps(bench.metals_throughput(0, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 512, steps=20), indent=1))
