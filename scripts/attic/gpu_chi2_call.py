"""Latency of the reference-signature call vega.chi2(params_dict) (what a sequential minimiser pays per evaluation)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import bench  # noqa: E402
from vega_amd import VegaInterface  # noqa: E402
prob = bench.build_problem('auto')
for item in prob.items.values():
    item.core.xi.ell_max = 4
vega = VegaInterface(None, problem=prob, max_batch=1)
pars = {'ap': 1.01, 'at': 0.99, 'bias_eta_LYA': -0.2, 'beta_LYA': 1.6}
for _ in range(20):
    vega.chi2(pars)
theta = vega.engine.theta_from_params(pars)[None, :]
for label, fn in (('vega.chi2(dict)', lambda: vega.chi2(pars)), ('engine.eval(theta)', lambda: vega.engine.eval(theta))):
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(300):
            fn()
        best = min(best, (time.perf_counter() - t0) / 300)
    print(f'{label}: {best * 1e6:.1f} us', flush=True)
