"""B = 1 distortion-matrix product, 8 distinct 2500^2 matrices round-robin (profiling target for the
HBM-traffic counters of the streaming kernel k_gemv<1>)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402
import bench  # noqa: E402
from vega_amd import VegaInterface  # noqa: E402

vega = VegaInterface('configs/auto/main.ini', search_dirs=[REPO / 'tests' / 'golden'], max_batch=1)
print(bench.distortion_microbench(vega.engine, torch))
