"""Phase stamps of k_pk_l3 blocks (experiment build -DVMX_EXP_L3_STAMP): entry -> tables staged -> weights -> first interpolation -> end"""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
out = REPO / 'gpurun_out' / 'l3_trace.bin'
os.environ['VMX_PK_TRACE'] = str(out)
os.environ['VEGAMX_LIBRARY'] = str(REPO / 'build_exp' / 'libvegamx_STAMP.so')
import numpy as np
import torch
from vega_amd import VegaInterface, synthetic
import bench
B = 256
prob = bench.build_problem('joint')
vega = VegaInterface(None, problem=prob, max_batch=B)
eng = vega.engine
eng.set_constant_nl_hint(True, gaussian=True)
theta = synthetic.walkers(eng.low.theta0, eng.names, B, seed=3, varied=bench.VARIED)
d = torch.from_numpy(theta).cuda()
o = torch.zeros(B, dtype=torch.float64, device='cuda')
for _ in range(6):
    eng.eval_device(d.data_ptr(), B, o.data_ptr())
eng.sync()
t = np.fromfile(out, dtype=np.uint64)
t = t[8 * 20000:8 * 20000 + 8 * 816].reshape(-1, 8)[:, :5].astype(np.int64)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
ph = np.diff(t, axis=1) / 100.0
print('blocks', len(t), 'start spread us', (t[:, 0].max() - t0) / 100.0, 'end max us', (t[:, 4].max() - t0) / 100.0)
print('phases median us: staged %.2f  weights %.2f  to first interp %.2f  loops %.2f' % tuple(np.median(ph, axis=0)))
print('phases p90    us: staged %.2f  weights %.2f  to first interp %.2f  loops %.2f' % tuple(np.percentile(ph, 90, axis=0)))
print('block life median', np.median((t[:, 4] - t[:, 0]) / 100.0))
