"""cProfile of the 1024-mock Monte-Carlo fit of the bench (host side: where the time outside the GPU goes)."""
import sys, cProfile, pstats, io, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
torch.cuda.init()
import bench
prob = bench.build_problem('joint')
bench.monte_carlo_fits(prob, 0, n_mocks=256)      # warm-up (library, quadratic form, autotune)
pr = cProfile.Profile()
pr.enable()
out = bench.monte_carlo_fits(prob, 0, n_mocks=1024)
pr.disable()
print({m: (out[m]['fits_per_s'], out[m]['seconds']) for m in ('migrad', 'bfgs')})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45)
print(s.getvalue()[:9000])
