"""Experiment: two engines (own streams, own workspaces) fed alternately, against one engine, B=256 joint."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402
import bench  # noqa: E402
from vega_amd import VegaInterface, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
prob = bench.build_problem('joint')
dev = torch.device('cuda', 0)
for lanes in (1, 2, 4):
    vegas = [VegaInterface(None, problem=prob, max_batch=B) for _ in range(lanes)]
    engs = [v.engine for v in vegas]
    for e in engs:
        e.set_constant_nl_hint(True, gaussian=True)
    pools = [torch.from_numpy(synthetic.walkers(engs[0].low.theta0, engs[0].names, B, varied=bench.VARIED, seed=100 + i)).to(dev)
             for i in range(8)]
    outs = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(lanes)]
    def step(i):
        engs[i % lanes].eval_device(pools[i % 8].data_ptr(), B, outs[i % lanes].data_ptr())
    for i in range(6):
        step(i)
    for e in engs:
        e.sync()
    t0 = time.perf_counter()
    n = 60
    for i in range(n):
        step(i)
    for e in engs:
        e.sync()
    dt = time.perf_counter() - t0
    print(f'lanes={lanes}: {dt / n * 1e3:.4f} ms/step  {B * n / dt:.0f} evals/s', flush=True)
    for v in vegas:
        v.close()
