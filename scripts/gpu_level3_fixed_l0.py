"""k_pk_l3 when the batch shares its L0 (the centre node alone, no interpolation): step time with level 3 on / off."""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import numpy as np
import torch
import bench
from vega_amd import VegaInterface, synthetic
B = 256
prob = bench.build_problem('joint')
varied = [v for v in bench.VARIED if v != 'L0_hcd']
for hw in ('0', '0.125'):
    os.environ['VEGA_AMD_LEVEL3'] = hw
    vega = VegaInterface(None, problem=prob, max_batch=B)
    eng = vega.engine
    eng.set_constant_nl_hint(True, gaussian=True)
    theta = synthetic.walkers(eng.low.theta0, eng.names, B, seed=3, varied=varied)
    d = torch.from_numpy(theta).cuda(); o = torch.zeros(B, dtype=torch.float64, device='cuda')
    for _ in range(60):
        eng.eval_device(d.data_ptr(), B, o.data_ptr())
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(200):
        eng.eval_device(d.data_ptr(), B, o.data_ptr())
    eng.sync()
    dt = (time.perf_counter() - t0) / 200
    eng.set_profiling(True)
    for _ in range(20):
        eng.eval_device(d.data_ptr(), B, o.data_ptr())
    eng.sync()
    tm = eng.timings(reset=True)
    print('level3', hw, f'{dt * 1e6:.1f} us / step = {B / dt:.0f} evals/s', 'pk stage %.1f us' % (tm['pk_multipoles'][0] / tm['pk_multipoles'][1] * 1e3), 'served', eng.level3_served() if hw != '0' else 0, flush=True)
    vega.close()
