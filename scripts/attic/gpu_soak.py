"""Soak run: random batch sizes and paths for a few minutes; every call repeated (same bits), chi2-only against the full chain
(1e-9), host entry against device entry (same bits when the hint is the one the host derives), a few walkers against the oracle."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import numpy as np
import torch
import bench
from vega_amd import VegaInterface, synthetic
from oracle import vega_cpu as oc

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.
rng = np.random.default_rng(5)
prob = bench.build_problem('joint')
vega = VegaInterface(None, problem=prob, max_batch=640, device=0)
eng = vega.engine
eng.set_lanes(2)
pool = synthetic.walkers(eng.low.theta0, eng.names, 640, varied=bench.VARIED, seed=77)
t_end = time.time() + seconds
n = worst = 0
while time.time() < t_end:
    B = int(rng.choice([1, 2, 7, 8, 9, 16, 63, 64, 65, 128, 255, 256, 257, 511, 512, 513, 640]))
    start = int(rng.integers(0, 640 - B + 1))
    theta = pool[start:start + B]
    a = eng.eval(theta)[0]
    b = eng.eval(theta)[0]
    assert np.array_equal(a, b), ('repeat', B)
    full, status, _ = eng.eval(theta, want_model=True)
    assert not status.any()
    rel = np.abs(a - full).max() / np.abs(full).max()
    worst = max(worst, rel)
    assert rel < 1e-9, ('quad vs full', B, rel)
    if B >= 64:
        eng.set_constant_nl_hint(True, gaussian=True)
        d = vega.chi2_batch_device(torch.from_numpy(theta).to('cuda:0')).cpu().numpy()
        d2 = vega.chi2_batch_device(torch.from_numpy(theta).to('cuda:0')).cpu().numpy()
        assert np.array_equal(d, d2), ('device repeat', B)
        assert np.abs(d - a).max() / np.abs(a).max() < 1e-12, ('device vs host', B)
    if n % 50 == 0:
        j = int(rng.integers(0, B))
        ref = oc.chi2(prob, dict(zip(eng.names, theta[j])))
        assert abs(a[j] - ref) < 1e-9 * abs(ref), ('oracle', B, j)
    n += 1
print('soak ok:', n, 'rounds, worst quad-vs-full', worst)
