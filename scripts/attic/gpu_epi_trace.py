"""Where the contraction epilogue of the quadratic-form launch spends its time.  Needs an experiment build of the library:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DVMX_EPI_TRACE -o build_exp/libvegamx_epi.so vega_amd/csrc/vegamx.hip
    VEGAMX_LIBRARY=$PWD/build_exp/libvegamx_epi.so python scripts/gpu_epi_trace.py"""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / 'tests'))
out = REPO / 'gpurun_out' / 'epi_trace.bin'
os.environ['VMX_QUAD_TRACE'] = str(out)
import numpy as np
from conftest import synth_joint_problem
from vega_amd import VegaInterface, synthetic
import bench
vega = VegaInterface(None, problem=synth_joint_problem(), max_batch=256)
eng = vega.engine
theta = synthetic.walkers(eng.low.theta0, eng.names, 256, seed=3, varied=bench.VARIED)
eng.set_profiling(True)
for _ in range(6):
    eng.eval(theta)
eng.sync()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4).astype(np.float64)
t = t[t[:, 0] > 0]
d = np.diff(t, axis=1) / 100.0
print('blocks', len(t))
for name, col in (('K loop end -> epilogue entered (set-up of the next entry / nothing)', 0), ('wait for E + barrier', 1), ('reductions + stores issued', 2)):
    print(f'{name:70s} median {np.median(d[:, col]):6.2f} us  p10 {np.percentile(d[:, col], 10):6.2f}  p90 {np.percentile(d[:, col], 90):6.2f}')
