"""Where the time of a batched Monte-Carlo fit goes: wall per phase for n_mocks in (128, 1024)."""
import cProfile, pstats, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402,F401
import bench  # noqa: E402
from vega_amd import VegaInterface  # noqa: E402

prob = bench.build_problem('joint')
for n_mocks in (128, 1024):
    out = bench.monte_carlo_fits(prob, 0, n_mocks=n_mocks)
    print(n_mocks, {k: out[k] for k in ('fits_per_s', 'seconds', 'chi2_evaluations', 'evals_per_fit', 'valid_fraction')}, flush=True)
pr = cProfile.Profile()
pr.enable()
bench.monte_carlo_fits(prob, 0, n_mocks=1024)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
