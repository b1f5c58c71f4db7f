"""The spline bins with one ln r grid for the four multipoles (`fht_lowring = False`: `k_xi_quad_plain<2, true>`, knot index and
B-spline weights once per bin) against the default `fht_lowring = True` (`<2, false>`): B = 256 joint workload, one batch in
flight, the engine's own event pairs per kernel.  (development aid)"""
import sys
import time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch  # noqa: E402
import bench  # noqa: E402
from vega_amd import VegaInterface, synthetic  # noqa: E402

B = 256
dev = torch.device('cuda', 0)
if len(sys.argv) > 1 and sys.argv[1] == 'bits':
    # chi2 of the same walkers on a common grid (fht_lowring = False) - run once with the default library and once with a
    # -DVMX_EXP_NO_SAME_GRID build (VEGAMX_LIBRARY), the outputs must be equal bit for bit:
    #   python scripts/gpu_same_grid.py bits > a; VEGAMX_LIBRARY=build_exp/nosame.so python scripts/gpu_same_grid.py bits > b; cmp a b
    prob = bench.build_problem('joint')
    for item in prob.items.values():
        item.core.xi.fht_lowring = False
    vega = VegaInterface(None, problem=prob, max_batch=B)
    eng = vega.engine
    theta = torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=7)).to(dev)
    out = torch.zeros(B, dtype=torch.float64, device=dev)
    eng.eval_device(theta.data_ptr(), B, out.data_ptr())
    eng.sync()
    chi2_full, _, model = eng.eval(theta.cpu().numpy()[:16], want_model=True)
    print(out.cpu().numpy().tobytes().hex())
    print(chi2_full.tobytes().hex())
    print(model.tobytes().hex()[:4096])
    vega.close()
    sys.exit(0)
for lowring in (True, False, True, False):
    prob = bench.build_problem('joint')
    for item in prob.items.values():
        for pipe in [item.core] + [m.pipeline for m in getattr(item, 'metals', []) if hasattr(m, 'pipeline')]:
            pipe.xi.fht_lowring = lowring
    vega = VegaInterface(None, problem=prob, max_batch=B)
    eng = vega.engine
    eng.set_constant_nl_hint(True, gaussian=True)
    eng.set_lanes(1)
    pools = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=100 + i)).to(dev)
             for i in range(8)]
    out = torch.zeros(B, dtype=torch.float64, device=dev)
    for i in range(150):
        eng.eval_device(pools[i % 8].data_ptr(), B, out.data_ptr())
    eng.sync()
    eng.set_profiling(True)
    eng.timings(reset=True)
    for i in range(8):
        eng.eval_device(pools[i % 8].data_ptr(), B, out.data_ptr())
    eng.sync()
    prof = {k: ms / n for k, (ms, n) in eng.timings().items() if n}
    eng.set_profiling(False)
    t0 = time.perf_counter()
    n = 100
    for i in range(n):
        eng.eval_device(pools[i % 8].data_ptr(), B, out.data_ptr())
    eng.sync()
    dt = time.perf_counter() - t0
    print(f'fht_lowring={lowring}: {dt / n * 1e3:.4f} ms/step {B * n / dt:.0f} evals/s  xi_bins {prof.get("xi_bins", 0) * 1e3:.1f} us  '
          f'{ {k: round(v * 1e3, 1) for k, v in prof.items()} }', flush=True)
    vega.close()
