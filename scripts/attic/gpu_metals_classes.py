"""Per-class times of the joint + metals workload (configs[3] share, B = 512), exact pipelines and static basis."""
import sys, json
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import torch
import bench
from vega_amd import VegaInterface, synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
prob = bench.build_problem('joint_metals')
for freeze in (False, True):
    vega = VegaInterface(None, problem=prob, max_batch=B)
    if freeze:
        vega.freeze_static_metals()
    eng = vega.engine
    eng.set_constant_nl_hint(True, gaussian=True)
    theta = torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=5)).cuda()
    chi2 = torch.zeros(B, dtype=torch.float64, device='cuda')
    for _ in range(3):
        eng.eval_device(theta.data_ptr(), B, chi2.data_ptr())
    eng.sync()
    eng.set_profiling(True)
    eng.timings(reset=True)
    for _ in range(10):
        eng.eval_device(theta.data_ptr(), B, chi2.data_ptr())
    eng.sync()
    t = eng.timings(reset=True)
    print('static_basis' if freeze else 'exact', json.dumps({k: round(v[0] / 10 * 1e3, 1) for k, v in t.items() if v[1]}), 'sum', round(sum(v[0] for v in t.values()) / 10 * 1e3, 1), flush=True)
    vega.close()
