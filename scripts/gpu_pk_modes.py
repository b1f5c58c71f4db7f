"""P(k,mu) stage at B=256 on the joint problem: class time and chi2 agreement of the table levels / walkers per thread."""
import os, sys, subprocess, json
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1:
    sys.path.insert(0, str(REPO))
    import numpy as np
    from vega_amd import VegaInterface, synthetic
    vega = VegaInterface('configs/joint/main.ini', search_dirs=[REPO / 'tests' / 'golden'], max_batch=int(os.environ.get('PKB', '256')))
    eng = vega.engine
    theta = synthetic.walkers(eng.low.theta0, eng.names, int(os.environ.get('PKB', '256')), seed=3,
                              varied=['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO',
                                      'drp_QSO', 'bias_hcd', 'beta_hcd', 'L0_hcd'])
    for _ in range(3):
        chi2 = eng.eval(theta)[0]
    eng.set_profiling(True)
    eng.timings(reset=True)
    for _ in range(20):
        eng.eval(theta)
    t = eng.timings(reset=True)
    np.save(sys.argv[1], chi2)
    print(json.dumps({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in t.items() if v[1]}), list(eng.debug_read(4, 0, 7)))
    sys.exit(0)
out = REPO / 'gpurun_out'
ref = None
import numpy as np
MODES = {'level1': {'VMX_NO_TAB2': '1'}, 'level2_nw1': {'VMX_PK_NW': '1'}, 'default': {}, 'no_fused_chi2': {'VMX_NO_FUSED_CHI2': '1'}, 'no_plain_pair': {'VMX_NO_PLAIN_PAIR': '1'},
         'noload': {'VEGAMX_LIBRARY': str(REPO / 'build_exp' / 'libvegamx_noload.so')}, 'B1024': {'PKB': '1024'}, 'L32': {'VMX_QUAD_L': '32'}, 'L36': {'VMX_QUAD_L': '36'}, 'L40': {'VMX_QUAD_L': '40'}, 'L44': {'VMX_QUAD_L': '44'}, 'L48': {'VMX_QUAD_L': '48'}, 'L52': {'VMX_QUAD_L': '52'}, 'pro128': {'VMX_PROLOGUE_THREADS': '128'}, 'pro512': {'VMX_PROLOGUE_THREADS': '512'}, 'no_ring': {'VMX_NO_FFT_RING': '1'}, 'B1024_no_ring': {'PKB': '1024', 'VMX_NO_FFT_RING': '1'}, 'B64': {'PKB': '64'}, 'B64_no_ring': {'PKB': '64', 'VMX_NO_FFT_RING': '1'}, 'fft_1cu': {'VMX_FFT_LDS': '24000'}, 'fft_wide': {'VMX_FFT_WIDE': '1'}, 'fft_wide_1cu': {'VMX_FFT_WIDE': '1', 'VMX_FFT_LDS': '24000'}}
for label in (os.environ.get('PK_MODES', 'level1,default').split(',')):
    env = MODES[label]
    f = out / f'pkmode_{label}.npy'
    r = subprocess.run([sys.executable, __file__, str(f)], env={**os.environ, **env}, capture_output=True, text=True)
    print(label, r.stdout.strip().split('\n')[-1] if r.stdout else r.stderr[-2000:])
    c = np.load(f)
    if ref is None:
        ref = c
    if c.size == ref.size:
        print('   max rel chi2 diff vs level1:', float(np.abs(c / ref - 1).max()))
