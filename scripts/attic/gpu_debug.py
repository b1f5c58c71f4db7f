"""Stage-by-stage comparison of the engine with the oracle on the GPU box (development aid)."""
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
import __graft_entry__ as g  # noqa: E402
g.build()
from vega_amd import VegaInterface  # noqa: E402
from oracle import vega_cpu as oc  # noqa: E402

GOLD = REPO / 'tests' / 'golden'
cfg = sys.argv[1] if len(sys.argv) > 1 else 'joint'
t0 = time.time()
vega = VegaInterface(f'configs/{cfg}/main.ini', search_dirs=[GOLD], max_batch=16)
print('engine build', time.time() - t0)
prob = vega.problem
eng = vega.engine

taps = {}
ref_model = oc.compute_model(prob, taps=taps)
ref_chi2 = oc.chi2(prob)

chi2, status, model = eng.eval(eng.theta_from_params()[None, :], want_model=True)
print('status', status, 'chi2', chi2[0], 'ref', ref_chi2, 'rel', abs(chi2[0] - ref_chi2) / abs(ref_chi2))

n_pipe = len(eng.pipe_index)
nkp = 832
pl = eng.debug_read(0, 0, 4 * 1 * n_pipe * nkp).reshape(4, n_pipe, nkp)
for (name, comp), pid in eng.pipe_index.items():
    if comp in ('peak', 'smooth'):
        for i, ell in enumerate((0, 2, 4, 6)):
            ref = taps[name][comp]['pk_ell'][ell]
            got = pl[i, pid, :814]
            print(f'pk_ell {name} {comp} ell={ell}: max abs err {np.abs(got - ref).max():.3e} scale {np.abs(ref).max():.3e}')
        n = prob.items[name].model_grid.size
        npad = (n + 31) // 32 * 32
        xi = eng.debug_read(1, pid, npad)[:n]
        ref = taps[name][comp]['xi_core']
        print(f'xi_core {name} {comp}: max abs err {np.abs(xi - ref).max():.3e} scale {np.abs(ref).max():.3e}')
for name, sl in eng.model_slices.items():
    ref = ref_model[name]
    print(f'model {name}: max abs err {np.abs(model[0, sl] - ref).max():.3e} scale {np.abs(ref).max():.3e}')

# walkers from the golden file
if not (GOLD / f'expected_{cfg}.npz').exists():
    (GOLD / 'x').parent  # no golden walkers for this config
exp = np.load(GOLD / f'expected_{cfg}.npz') if (GOLD / f'expected_{cfg}.npz').exists() else None
if exp is None:
    exp = {'param_names': np.array(eng.names), 'theta': eng.low.theta0[None, :].repeat(2, 0), 'chi2': np.array([ref_chi2, ref_chi2])}
    for i in range(2):
        for name in prob.items:
            exp[f'walker{i}/model/{name}'] = ref_model[name]
names = [str(n) for n in exp['param_names']]
theta = np.stack([eng.theta_from_params(dict(zip(names, row))) for row in exp['theta']])
chi2, status, model = eng.eval(theta, want_model=True)
print('walker status', status)
print('walker chi2 rel err', np.abs(chi2 - exp['chi2']) / np.abs(exp['chi2']))
for i in range(theta.shape[0]):
    for name, sl in eng.model_slices.items():
        ref = exp[f'walker{i}/model/{name}']
        err = np.abs(model[i, sl] - ref).max() / np.abs(ref).max()
        if i < 2:
            print(f'walker{i} model {name}: scaled max err {err:.3e}')
eng.set_profiling(True)
for _ in range(3):
    eng.eval(theta)
print({k: v for k, v in eng.timings().items() if v[1]})

# ---- timing at B = 256
from vega_amd import synthetic  # noqa: E402
vega.close()
vega = VegaInterface(f'configs/{cfg}/main.ini', search_dirs=[GOLD], max_batch=256)
eng = vega.engine
theta = synthetic.walkers(eng.low.theta0, eng.names, 256, seed=3,
                          varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO',
                                  'drp_QSO', 'bias_hcd', 'beta_hcd', 'L0_hcd'])
for _ in range(2):
    eng.eval(theta)
eng.set_profiling(True)
t0 = time.time()
for _ in range(5):
    eng.eval(theta)
dt = (time.time() - t0) / 5
print(f'B=256: {dt*1e3:.3f} ms/step, {256/dt:.0f} evals/s')
print({k: round(v[0] / v[1], 4) for k, v in eng.timings().items() if v[1]})
eng.set_profiling(False)
for B in (1, 4, 16, 64):
    for _ in range(2):
        eng.eval(theta[:B])
    t0 = time.time()
    for _ in range(20):
        eng.eval(theta[:B])
    dt = (time.time() - t0) / 20
    print(f'B={B}: {dt*1e6:.1f} us/step, {B/dt:.0f} evals/s')
