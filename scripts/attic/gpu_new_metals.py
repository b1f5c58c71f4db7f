"""Throughput of a correlation with `new_metals = True` matrices (15 Kronecker-form metal matrices on the auto-correlation,
built from the synthetic stacked-delta file): Kronecker application (two small products per walker and pair) against the
same matrices uploaded dense.  usage: python scripts/gpu_new_metals.py [B]"""
import sys
import tempfile
import time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(REPO), str(REPO / 'tests')]
import numpy as np      # noqa: E402
import torch            # noqa: E402
from conftest import new_metals_problem    # noqa: E402
from vega_amd import VegaInterface, synthetic  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
with tempfile.TemporaryDirectory() as tmp:
    prob, name = new_metals_problem(Path(tmp), 'auto')
    for kron in (True, False):
        vega = VegaInterface(None, problem=prob, max_batch=B, kron_metals=kron)
        eng = vega.engine
        theta = synthetic.walkers(eng.low.theta0, eng.names, B, varied=['bias_eta_LYA', 'beta_LYA', 'ap', 'at', 'bias_hcd'], seed=1)
        d_theta = torch.from_numpy(theta).cuda()
        d_chi2 = torch.zeros(B, dtype=torch.float64, device='cuda')
        eng.set_constant_nl_hint(True)
        for _ in range(3):
            eng.eval_device(d_theta.data_ptr(), B, d_chi2.data_ptr())
        eng.sync()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            eng.eval_device(d_theta.data_ptr(), B, d_chi2.data_ptr())
        eng.sync()
        dt = (time.perf_counter() - t0) / n
        eng.set_profiling(True)
        eng.timings(reset=True)
        for _ in range(3):
            eng.eval_device(d_theta.data_ptr(), B, d_chi2.data_ptr())
        eng.sync()
        tm = eng.timings(reset=True)
        print(f'kron={kron}: {dt * 1e3:.3f} ms / step of {B} walkers = {B / dt:,.0f} evals/s; '
              f"metal products {tm['metal_matrix_product'][0] / 3:.3f} ms / step ({tm['metal_matrix_product'][1] // 3} launches); "
              f'chi2[0] = {float(d_chi2[0]):.12g}', flush=True)
        vega.close()
