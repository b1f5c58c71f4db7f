"""Steps of the COEFMOD = 2 joint fit alone (factored chi2 form, one batch in flight, B = 256): the command whose rocprofv3
kernel stats are committed as profiles/rNN_coefmod2_kernel_stats.csv."""
import sys
import tempfile
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))

import torch
torch.cuda.init()
import bench
from vega_amd import VegaInterface, synthetic
from vega_amd.setup import build_problem

with tempfile.TemporaryDirectory() as tmp:
    main = synthetic.dmat_file_configs(tmp, REPO / 'tests' / 'golden', config='joint', coef=2)
    prob = build_problem(main, search_dirs=[tmp, REPO / 'tests' / 'golden'])
B = 256
vega = VegaInterface(None, problem=prob, max_batch=B)
eng = vega.engine
eng.set_constant_nl_hint(True, gaussian=True)
theta = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, B, varied=bench.VARIED, seed=synthetic.SEED + i)).cuda() for i in range(4)]
out = torch.zeros(B, dtype=torch.float64, device='cuda')
for i in range(60):
    eng.eval_device(theta[i % 4].data_ptr(), B, out.data_ptr())
eng.sync()
print(eng.last_form(), float(out[0]))
vega.close()
