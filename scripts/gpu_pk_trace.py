"""Block timeline of k_pk_tab2 at B=256 (VMX_PK_TRACE): when blocks start / end, per XCD and CU."""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
out = REPO / 'gpurun_out' / 'pk_trace.bin'
os.environ['VMX_PK_TRACE'] = str(out)
import numpy as np
from vega_amd import VegaInterface, synthetic
B = int(os.environ.get('PKB', '256'))
vega = VegaInterface('configs/joint/main.ini', search_dirs=[REPO / 'tests' / 'golden'], max_batch=B)
eng = vega.engine
theta = synthetic.walkers(eng.low.theta0, eng.names, B, seed=3,
                          varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO',
                                  'drp_QSO', 'bias_hcd', 'beta_hcd', 'L0_hcd'])
eng.set_profiling(True)
for _ in range(4):
    eng.eval(theta)
eng.sync()
t = np.fromfile(out, dtype=np.uint64).reshape(-1, 4)
t = t[t[:, 0] > 0]
t0 = t[:, 0].min()
start = (t[:, 0] - t0) / 100.0      # us
end = (t[:, 1] - t0) / 100.0
dur = end - start
print('blocks', len(t), 'span us', end.max(), 'median dur', np.median(dur))
live = dur > 2.0
print('live blocks', live.sum(), 'dur quantiles live', np.percentile(dur[live], [5, 25, 50, 75, 95]), 'dead dur median', np.median(dur[~live]))
hw = t[:, 2].astype(np.int64); xcc = t[:, 3].astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
print('distinct CUs', len(np.unique(cuid)), 'xcc counts', np.bincount(xcc))
# concurrency over time
edges = np.arange(0, end.max() + 5, 5.0)
for lo in edges:
    act = ((start < lo + 5) & (end > lo) & live).sum()
    st = ((start >= lo) & (start < lo + 5) & live).sum()
    print(f'{lo:6.0f} us: live blocks active {act:5d}  started {st:5d}')
# per-CU busy: last end per CU
last = np.zeros(cuid.max() + 1)
np.maximum.at(last, cuid, end)
print('per-CU last end: min', last[last > 0].min(), 'median', np.median(last[last > 0]), 'max', last.max())
# durations by (item, k tile): block index = (z * grid.y + y) * grid.x + x with x = walker pairs, y = item group, z = k tile
raw = np.fromfile(out, dtype=np.uint64).reshape(-1, 4)
nx = (B + 1) // 2
ny = 2
nz = len(raw) // (nx * ny)
r = raw[:nx * ny * nz].reshape(nz, ny, nx, 4).astype(np.float64)
for y in range(ny):
    row = []
    for z in range(nz):
        m = r[z, y, :, 0] > 0
        if m.any():
            d = (r[z, y, m, 1] - r[z, y, m, 0]) / 100.0
            s0 = (r[z, y, m, 0].min() - float(t0)) / 100.0
            row.append(f'{d.mean():5.1f}@{s0:4.0f}')
    print('item group', y, 'mean duration @ first start per k tile:', ' '.join(row))
