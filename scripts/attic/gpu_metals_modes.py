import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa
import bench  # noqa
print(bench.metals_throughput(0, batch=int(sys.argv[1]) if len(sys.argv) > 1 else 512))
