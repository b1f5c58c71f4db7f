"""Benchmark of the model + chi2 hot path on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N --steps K --warmup W          (N > 1, no WORLD_SIZE in the environment: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus 2 --dist-backend gloo --ranks-share-gpu --core-only     (rehearsal of the N > 1 path on ONE GPU)

A "step" evaluates one batch of B synthetic walkers (parameter points) of BASELINE.json configs[2]:
the joint Lya x Lya + QSO x Lya fit (two correlation items, shared parameters, 4 P(k)->xi pipelines
per evaluation, ell = 0,2,4,6), with the seeded dense synthetic distortion matrices (2500^2 and
5000^2) and covariances of SURVEY.md section 8d.  Walkers are resident in HBM before the timed
region.  With N ranks each rank evaluates its own B walkers per step (weak scaling) and the chi2
vectors are exchanged with ONE all-gather per step (RCCL over xGMI).

Rank 0 prints one JSON line.  Extra objects:
  roofline      the distortion-matrix step (the kernel SURVEY.md section 8d prices: fp64 MFMA at B = 256), timed live
                with HIP events over the timed steps; roofline_other_kernels: the other heavy kernels, among them
                the P(k,mu) stage (fp64 vector ALU), from the calibration pass
  distortion    the B = 1 distortion-matrix product of configs[1] (2500^2 fp64): GB/s vs HBM
  cpu_baseline  the CPU oracle (NumPy restatement of the reference) timed on this host
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TF = 78.6       # MI355X FP64 vector peak (spec; 256 CU x 128 flop/clk x 2.4 GHz)
FP64_MFMA_PEAK_TF = 78.6       # MI355X FP64 matrix peak (spec)
# what the instructions sustain on this part (scripts/micro/fp64_peaks.hip, profiles/r01_fp64_peaks.txt):
FP64_MFMA_MEASURED_TF = 49.3   # v_mfma_f64_16x16x4_f64, 8 waves / SIMD (the C^-1 and FFTLog products)
FP64_MFMA_4X4X4_MEASURED_TF = 73.4   # v_mfma_f64_4x4x4_4b_f64 (the distortion products; profiles/r01_fp64_mfma_4x4x4.txt)
FP64_VALU_MEASURED_TF = 61.6   # v_fma_f64 at 4 waves / SIMD, the occupancy of k_pk_multipoles (66.5 at 8)

# Algorithmic flops per (k, mu) grid point of a paired peak+smooth group:
# + - x count 1, FMA 2, every transcendental / rsqrt 1.
FLOPS_PER_POINT = {'auto': 35, 'cross': 43}
# ... when the batch shares its Gaussian factors (level-2 tables, k_pk_tab2): Kaiser + HCD amplitudes 8 (cross 9), table
# products 2, mu^6 1, eight moment sums 14, the HCD progression 1; cross: + Lorentzian rsqrt 9
FLOPS_PER_POINT_TAB2 = {'auto': 26, 'cross': 36}


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that committed one (the counter passes cannot run inside this process)."""
    found = sorted((REPO / 'profiles').glob(f'r[0-9][0-9]_{suffix}'))
    return found[-1] if found else None


def build_problem(workload):
    from vega_amd.setup import build_problem as bp
    from vega_amd import synthetic
    cfg = {'joint': 'configs/joint/main.ini', 'auto': 'configs/auto/main.ini',
           'joint_metals': 'configs/joint_metals/main.ini',
           'joint_metals_fast': 'configs/joint_metals_fast/main.ini'}[workload]
    prob = bp(cfg, search_dirs=[REPO / 'tests' / 'golden'])
    for item in prob.items.values():
        item.distortion = synthetic.distortion_matrix(item.model_grid.rp, item.model_grid.rt)
        item.set_covariance(synthetic.covariance(item.data_grid.rp, item.data_grid.rt),
                            inv_masked_cov=synthetic.inverse_masked_covariance(item.data_grid.rp, item.data_grid.rt, item.data_mask))
    return prob


WORKLOAD_TEXT = {
    'joint': 'joint: Lya x Lya + QSO x Lya joint fit, B={B} walkers/GPU/step, ell=0,2,4,6, dense synthetic 2500^2 + '
             '5000^2 distortion matrices and 1590^2 + 3180^2 inverse covariances (BASELINE configs[2])',
    'joint_metals': 'joint_metals: joint fit + full metals (SiII/SiIII/CIV: 15 + 4 metal pairs with their metal '
                    'matrices), B={B} walkers/GPU/step = {total} walkers/step over the node, same dense synthetic '
                    'distortion matrices and covariances (BASELINE configs[3]: 4096 walkers over 8 GPUs = --batch 512)',
    'joint_metals_fast': 'joint_metals_fast: configs[3] with the reference\'s fast_metals switch, B={B} walkers/GPU/step',
    'auto': 'auto: Lya x Lya auto-correlation only, B={B} walkers/GPU/step, dense synthetic 2500^2 distortion matrix',
}

# `general_walkers`: the same workload with walkers that ALSO differ in Arinyo, smoothing and peak-broadening parameters -
# no per-batch table applies, every walker runs its own P(k,mu) loops (what a sampler that frees those parameters gets)
GENERAL_EXTRA = ['dnl_arinyo_q1', 'par_sigma_smooth', 'per_sigma_smooth', 'sigmaNL_par']

VARIED = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'sigma_velo_disp_lorentz_QSO', 'drp_QSO',
          'bias_hcd', 'beta_hcd', 'L0_hcd', 'bias_eta_SiII(1190)', 'bias_eta_SiII(1193)',
          'bias_eta_SiIII(1207)', 'bias_eta_SiII(1260)', 'bias_eta_CIV(eff)']


def distortion_microbench(engine, torch, n=2500, copies=8, reps=40):
    """B = 1 product y = DM x with 8 distinct matrices (400 MB > Infinity Cache) round-robin, so
    every launch streams its matrix from HBM; HIP-event timed on the engine stream."""
    ld = (n + 31) // 32 * 32
    dev = torch.device('cuda', torch.cuda.current_device())
    mats = [torch.zeros(n, ld, dtype=torch.float64, device=dev) for _ in range(copies)]
    for m in mats:
        m[:, :n] = torch.rand(n, n, dtype=torch.float64, device=dev)
    x = torch.zeros(1, ld, dtype=torch.float64, device=dev)
    x[0, :n] = torch.rand(n, dtype=torch.float64, device=dev)
    y = torch.zeros(1, ld, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    for m in mats:
        engine.matvec_device(m.data_ptr(), n, ld, x.data_ptr(), 1, y.data_ptr())
    engine.sync()
    ref = (mats[-1][:, :n] @ x[0, :n])
    err = float((y[0, :n] - ref).abs().max() / ref.abs().max())
    engine.timings(reset=True)
    engine.set_profiling(True)
    for i in range(reps):
        engine.matvec_device(mats[i % copies].data_ptr(), n, ld, x.data_ptr(), 1, y.data_ptr())
    engine.sync()
    ms, launches = engine.timings(reset=True)['matvec_api']
    engine.set_profiling(False)
    # one hot-cache variant: same matrix every launch (50 MB, Infinity-Cache resident)
    engine.set_profiling(True)
    for i in range(reps):
        engine.matvec_device(mats[0].data_ptr(), n, ld, x.data_ptr(), 1, y.data_ptr())
    engine.sync()
    ms_hot, launches_hot = engine.timings(reset=True)['matvec_api']
    engine.set_profiling(False)
    # context: a library reduction (torch.sum) over the same matrices, same round-robin, timed with events on torch's stream
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for m in mats:
        m.sum()
    torch.cuda.synchronize()
    read_ms = 0.0
    for i in range(reps):
        ev0.record()
        mats[i % copies].sum()
        ev1.record()
        ev1.synchronize()
        read_ms += ev0.elapsed_time(ev1)
    plain_read = 8.0 * n * ld / (read_ms / reps * 1e-3) / 1e9
    algo_bytes = 8.0 * (n * n + n + n)          # 8 (N_out N_in + B N_in + B N_out)
    cold = algo_bytes / (ms / launches * 1e-3) / 1e9
    hot = algo_bytes / (ms_hot / launches_hot * 1e-3) / 1e9
    # (FETCH_SIZE pass of the same product under rocprofv3, scripts/gpu_matvec_only.py: the committed summary)
    traffic = None
    traffic_file = latest_profile('distortion_gemv_traffic.json')
    if traffic_file is not None and n == 2500:
        for name, rec in json.loads(traffic_file.read_text())['kernels'].items():
            if 'k_gemv1' in name:
                traffic = rec.get('fetch_bytes_per_launch')
    return {'shape': [n, n], 'batch': 1, 'bound': 'hbm', 'algorithmic_bytes': algo_bytes, 'traffic': traffic,
            'us_per_launch': ms / launches * 1e3, 'achieved': cold, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': cold / HBM_PEAK_GBS, 'achieved_cache_resident': hot, 'max_rel_err': err,
            'torch_sum_same_bytes_GBs': plain_read,   # a library reduction over the same bytes (context)
            'note': '8 distinct 50 MB matrices round-robin (HBM); cache_resident = one matrix reused'}


def distortion_csr(device, reps=60):
    """The same synthetic 2500^2 distortion matrix in CSR form (73 % non-zeros, the reference's `csr_array`), B = 1: the
    full chain of one model evaluation with the CSR product (`k_csr_spmm<1>`) timed by HIP events."""
    from scipy import sparse
    from vega_amd import VegaInterface
    prob = build_problem('auto')
    for item in prob.items.values():
        item.distortion = sparse.csr_array(item.distortion)
    item = next(iter(prob.items.values()))
    vega = VegaInterface(None, problem=prob, max_batch=1, device=device, csr_threshold=1.1)
    eng = vega.engine
    theta = eng.theta_from_params()[None, :]
    for _ in range(5):
        eng.eval(theta, want_model=True)
    eng.set_profiling(True)
    eng.timings(reset=True)
    for _ in range(reps):
        eng.eval(theta, want_model=True)
    ms, launches = eng.timings(reset=True)['distortion_product']
    eng.set_profiling(False)
    vega.close()
    nnz, (rows, cols) = item.distortion.nnz, item.distortion.shape
    algo = 12.0 * nnz + 8.0 * (rows + 1) + 8.0 * (rows + cols)          # SURVEY 8d (int64 row pointers here)
    gbs = algo / (ms / launches * 1e-3) / 1e9
    return {'shape': [rows, cols], 'nnz': int(nnz), 'density': nnz / (rows * cols), 'batch': 1, 'bound': 'hbm',
            'algorithmic_bytes': algo, 'us_per_launch': ms / launches * 1e3, 'achieved': gbs, 'peak': HBM_PEAK_GBS,
            'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS,
            'note': 'one matrix reused (42 MB of CSR arrays: Infinity-Cache resident after the first evaluation)'}


def single_point_latency(device, reps=200):
    """BASELINE configs[1]: Lya x Lya auto only, ell = 0,2,4, dense 2500^2 distortion matrix, B = 1:
    wall-clock chi2 evaluations per second through the host interface (host theta in, host chi2 out)."""
    from vega_amd import VegaInterface
    prob = build_problem('auto')
    for item in prob.items.values():
        item.core.xi.ell_max = 4
    vega = VegaInterface(None, problem=prob, max_batch=1, device=device)
    theta = vega.engine.theta_from_params()[None, :]
    for _ in range(10):
        vega.engine.eval(theta)
    dt = float('inf')
    for _ in range(3):              # best of three: a single-point call is sensitive to host jitter
        t0 = time.perf_counter()
        for _ in range(reps):
            vega.engine.eval(theta)
        dt = min(dt, (time.perf_counter() - t0) / reps)
    vega.engine.set_profiling(True)
    for _ in range(5):
        vega.engine.eval(theta)
    vega.engine.timings(reset=True)
    for _ in range(50):
        vega.engine.eval(theta)
    kern = {k: round(v[0] / v[1] * 1e3, 2) for k, v in vega.engine.timings().items() if v[1]}
    vega.close()
    return {'workload': 'configs[1]: Lya x Lya auto, ell<=4, dense 2500^2 distortion, B=1', 'evals_per_s': 1.0 / dt,
            'us_per_eval': dt * 1e6, 'kernel_us_per_launch': kern}


def metals_throughput(device, batch=512, steps=10):
    """One GPU's share of BASELINE configs[3] (4096 walkers over 8 GPUs): joint fit + the 15 + 4 metal pairs of the
    reference's test configuration, 512 walkers per step, walkers resident in HBM.  Polynomial metal pairs run in
    their exact static form (VegaInterface.freeze_static_metals); `exact_pipelines` is the same with every pair on
    its own P(k,mu) -> xi pipeline."""
    import torch
    from vega_amd import VegaInterface, synthetic
    prob = build_problem('joint_metals')
    dev = torch.device('cuda', device)
    out = {'workload': f'configs[3] share: joint + metals (15 + 4 pairs), B={batch} walkers/step'}
    for label, freeze in (('exact_pipelines', False), ('static_basis', True)):
        vega = VegaInterface(None, problem=prob, max_batch=batch, device=device)
        if freeze:
            vega.freeze_static_metals()
        eng = vega.engine
        eng.set_constant_nl_hint(True, gaussian=True)
        pools = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, batch, varied=VARIED,
                                                    seed=synthetic.SEED + 77 + i)).to(dev) for i in range(4)]
        chi2 = [torch.zeros(batch, dtype=torch.float64, device=dev) for _ in range(4)]
        out[label] = {'pipelines_per_eval': len(eng.pipe_index)}
        for lanes, key in ((2, 'evals_per_s'), (1, 'one_batch_in_flight_evals_per_s')):
            eng.set_lanes(lanes)
            for i in range(40):
                eng.eval_device(pools[i % 4].data_ptr(), batch, chi2[i % 4].data_ptr())
            eng.sync()
            dt = float('inf')
            for _ in range(3):              # best of three blocks: a one-off stall (module load, clock ramp) is not the rate
                t0 = time.perf_counter()
                for i in range(steps):
                    eng.eval_device(pools[i % 4].data_ptr(), batch, chi2[i % 4].data_ptr())
                eng.sync()
                dt = min(dt, time.perf_counter() - t0)
            out[label][key] = batch * steps / dt
            if lanes == 2:
                out[label]['ms_per_step'] = dt / steps * 1e3
                out[label]['batches_in_flight'] = 2
        vega.close()
    return out


def coefmod_throughput(device, batch=256, steps=10, coef=2):
    """The joint fit with `distortion-file`s whose model grids are COEFMOD = 2 times finer than the data grids (what DESI
    production files look like, reference vega/data.py:441-473): 10 000 and 20 000 model bins against 1590 and 3180 fitted
    ones.  The quadratic form of chi2 in its two shapes - the half-form Q' (nq^2 flops per walker and item) and the factored
    form || U r0 - F dx ||^2 (2 n_masked nq), which the engine picks by itself on such grids - and the full chain, same
    walkers, resident in HBM."""
    import tempfile
    import torch
    from vega_amd import VegaInterface, synthetic
    from vega_amd.setup import build_problem as bp
    dev = torch.device('cuda', device)
    with tempfile.TemporaryDirectory() as tmp:
        main = synthetic.dmat_file_configs(tmp, REPO / 'tests' / 'golden', config='joint', coef=coef)
        prob = bp(main, search_dirs=[tmp, REPO / 'tests' / 'golden'])
    vega = VegaInterface(None, problem=prob, max_batch=batch, device=device)
    eng = vega.engine
    eng.set_constant_nl_hint(True, gaussian=True)
    pools = [torch.from_numpy(synthetic.walkers(eng.low.theta0, eng.names, batch, varied=VARIED, seed=synthetic.SEED + 31 + i)).to(dev)
             for i in range(4)]
    chi2 = [torch.zeros(batch, dtype=torch.float64, device=dev) for _ in range(4)]
    nq = {n: int(it.model_grid.size) + sum(len(range(t.r1[0], t.r1[1] + 1, t.r1[2])) * len(range(t.r2[0], t.r2[1] + 1, t.r2[2]))
                                           for t in it.broadband if t.pos == 'post' and t.kind == 'add') for n, it in prob.items.items()}
    nm = {n: int(it.data_size) for n, it in prob.items.items()}
    out = {'workload': f'joint fit with COEFMOD = {coef} distortion files: model grids {[int(it.model_grid.size) for it in prob.items.values()]}, '
                       f'fitted bins {list(nm.values())}, B={batch} walkers/step',
           'flops_per_walker': {'q': float(sum(v * v for v in nq.values())), 'factored': float(sum(2 * nm[n] * nq[n] for n in nq))}}

    def rate(lanes, model=None):
        eng.set_lanes(lanes)
        for i in range(12):
            eng.eval_device(pools[i % 4].data_ptr(), batch, chi2[i % 4].data_ptr(), *([model.data_ptr()] if model is not None else []))
        eng.sync()
        best = float('inf')
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(steps):
                eng.eval_device(pools[i % 4].data_ptr(), batch, chi2[i % 4].data_ptr(), *([model.data_ptr()] if model is not None else []))
            eng.sync()
            best = min(best, time.perf_counter() - t0)
        return batch * steps / best

    results = {}
    for kind in ('factored', 'q'):
        eng.set_quadratic_form_kind(kind)
        entry = {'evals_per_s': rate(2), 'one_batch_in_flight_evals_per_s': rate(1)}
        assert eng.last_form() == kind
        eng.set_profiling(True)
        eng.timings(reset=True)
        for i in range(4):
            eng.eval_device(pools[i % 4].data_ptr(), batch, chi2[i % 4].data_ptr())
        eng.sync()
        tm = eng.timings(reset=True)
        eng.set_profiling(False)
        ms = tm['quadratic_form_product'][0] / tm['quadratic_form_product'][1]
        tf = out['flops_per_walker'][kind] * batch / (ms * 1e-3) / 1e12
        entry['product'] = {'ms_per_launch': ms, 'TFLOP/s': tf, 'frac_of_fp64_mfma_peak': tf / FP64_MFMA_PEAK_TF}
        results[kind] = (entry, chi2[3].clone())
        out[kind] = entry
    eng.set_quadratic_form_kind('auto')
    d_model = torch.zeros(batch, eng.model_size, dtype=torch.float64, device=dev)
    out['full_chain'] = {'evals_per_s': rate(1, d_model)}
    full = chi2[3].clone()
    for kind in ('factored', 'q'):
        out[kind]['max_rel_chi2_diff_vs_full_chain'] = float(((results[kind][1] - full).abs() / full.abs()).max())
    eng.eval_device(pools[0].data_ptr(), batch, chi2[0].data_ptr())
    eng.sync()
    out['chosen_by_the_engine'] = eng.last_form()
    vega.close()
    return out


def monte_carlo_fits(prob, device, n_mocks=1024):
    """One GPU's share of BASELINE configs[4] (8192 mocks over 8 GPUs): n_mocks Monte-Carlo realisations of the
    bench workload, each fitted over (ap, at, bias_eta_LYA, beta_LYA, beta_QSO, bias_hcd) by MIGRAD - the bias pre-fit, then the
    full fit, HESSE (reference vega/analysis.py:224-308 -> vega/minimizer.py:66-97) - all at once; mock generation (Cholesky
    factor x normal draws) included.  `migrad`: the fits advanced on the device (vmx_fit_migrad); `migrad_numpy_driver`: the
    same fits advanced in lock-step by the NumPy driver through the host entry (round 4's path); `bfgs`: the vectorised
    variable-metric minimiser (Minuit's conventions, not its trajectory)."""
    from vega_amd import VegaInterface
    vega = VegaInterface(None, problem=prob, max_batch=4096, device=device)
    vega.chi2()
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd']
    limits = {'ap': (0.5, 1.5), 'at': (0.5, 1.5), 'bias_eta_LYA': (-2., 0.), 'beta_LYA': (0., 5.),
              'beta_QSO': (0., 1.), 'bias_hcd': (-0.5, 0.)}
    errors = {'ap': 0.01, 'at': 0.01, 'bias_eta_LYA': 0.01, 'beta_LYA': 0.1, 'beta_QSO': 0.1, 'bias_hcd': 0.01}
    sample = {'limits': limits, 'values': {n: vega.params[n] for n in names}, 'errors': errors,
              'fix': {n: False for n in names}}
    truth = np.array([vega.params[n] for n in names])
    out = {'workload': f'{n_mocks} mocks x {len(names)}-parameter fits (mock generation + minimisation + Hessian), after an untimed '
                       '128-mock run of the same call (buffers of the mock pool, clocks)'}
    saved = os.environ.get('VEGA_AMD_FIT_DRIVER')
    try:
        for label, method, driver in (('migrad', 'migrad', 'device'), ('migrad_numpy_driver', 'migrad', 'python'), ('bfgs', 'bfgs', 'python')):
            os.environ['VEGA_AMD_FIT_DRIVER'] = driver
            vega.run_monte_carlo(num_mocks=128, seed=5, sample_params=sample, method=method)
            runs = []
            for _ in range(2):          # (the same run twice, the faster one reported, both listed: one-off stalls of a shared host)
                t0 = time.perf_counter()
                res_i = vega.run_monte_carlo(num_mocks=n_mocks, seed=11, sample_params=sample, method=method)
                runs.append((time.perf_counter() - t0, res_i))
            dt, res = min(runs, key=lambda r: r[0])
            pulls = (res.values - truth) / res.errors
            out[label] = {'fits_per_s': n_mocks / dt, 'seconds': dt, 'chi2_evaluations': int(res.nfcn.sum()),
                          'evals_per_fit': float(res.nfcn.mean()), 'valid_fraction': float(res.is_valid.mean()),
                          'pull_rms': [float(v) for v in pulls.std(axis=0)], 'seconds_of_both_runs': [r[0] for r in runs]}
            st = getattr(res, 'driver_stats', None)
            if st:
                # the driver's own account: the GPU is idle while the host takes its turn of a round (HIP events around it) and
                # outside the rounds (mock generation, the quadratic form's linear terms of the new pool, results)
                idle = st['gpu_idle_seconds_between_rounds'] + (dt - st['seconds_rounds'])
                out[label].update({'rounds': st['rounds'], 'engine_calls': st['engine_calls'],
                                   'seconds_in_rounds': st['seconds_rounds'], 'gpu_idle_seconds_between_rounds': st['gpu_idle_seconds_between_rounds'],
                                   'evals_per_s_in_rounds': st['evaluations'] / st['seconds_rounds'],
                                   'engine_calls_by_batch_size': st['calls_by_batch'], 'evaluations_by_batch_size': st['evaluations_by_batch'],
                                   'host_fraction': idle / dt,
                                   'driver_seconds': {k: st[k] for k in ('seconds', 'seconds_setup', 'seconds_rounds', 'seconds_host_waiting',
                                                                        'seconds_waiting_for_draws', 'seconds_enqueuing_waves', 'seconds_enqueuing_rounds',
                                                                        'seconds_enqueuing_calls') if k in st}})
    finally:
        if saved is None:
            os.environ.pop('VEGA_AMD_FIT_DRIVER', None)
        else:
            os.environ['VEGA_AMD_FIT_DRIVER'] = saved
    out['fits_per_s'] = out['migrad']['fits_per_s']
    out['host_fraction'] = out['migrad'].get('host_fraction')
    out['host_fraction_note'] = ('share of the run during which the GPU had nothing of it to do: everything outside the rounds of the '
                                 'device-resident fits + the host\'s turn of every round (HIP events on the stream); the kernel-trace '
                                 'view of the same run: profiles/r05_mc_fits_timeline.txt')
    vega.close()
    return out


_CPU = {}


def _cpu_worker_init(workload):
    """Worker of the CPU baseline: its own copy of the problem, one BLAS thread."""
    from oracle import vega_cpu as oracle
    try:
        from threadpoolctl import threadpool_limits
        _CPU['limit'] = threadpool_limits(limits=1)
    except Exception:       # pragma: no cover
        pass
    _CPU['prob'] = build_problem(workload)
    _CPU['oracle'] = oracle
    oracle.chi2(_CPU['prob'])       # warm caches (grids, FFTLog objects)


def _cpu_worker_eval(pars):
    return _CPU['oracle'].chi2(_CPU['prob'], pars)


def cpu_baseline(workload, names, theta, repeats=7, warmups=2, extra_theta=None):
    """The oracle (CPU restatement of the reference) on the host cores, SURVEY 8d: a pool of one worker per core
    (one BLAS thread each), every repeat evaluates one walker per worker; median of `repeats` after `warmups`; plus
    the single-core figure from one worker alone.  Runs BEFORE the GPU is initialised (worker processes are started
    from a process that has not touched the device)."""
    import multiprocessing as mp
    cores = os.cpu_count() or 1
    workers = max(1, min(cores, 16, theta.shape[0]))        # the GPU box's share for one GPU is 16 processes
    saved = {k: os.environ.get(k) for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS')}
    for k in saved:
        os.environ[k] = '1'
    try:
        # ProcessPoolExecutor, not multiprocessing.Pool: a worker that dies (or whose initializer raises) breaks the pool
        # with an exception instead of being respawned for ever
        from concurrent.futures import ProcessPoolExecutor
        with ProcessPoolExecutor(workers, mp_context=mp.get_context('spawn'), initializer=_cpu_worker_init,
                                 initargs=(workload,)) as pool:
            walkers = [dict(zip(names, row)) for row in theta]
            times, vals = [], {}
            t_start = time.perf_counter()
            for r in range(warmups + repeats):
                block = [(r * workers + i) % len(walkers) for i in range(workers)]
                t0 = time.perf_counter()
                out = list(pool.map(_cpu_worker_eval, [walkers[i] for i in block], chunksize=1))
                dt = time.perf_counter() - t0
                if r >= warmups:
                    times.append(dt)
                vals.update(zip(block, out))
            # one core: one walker at a time (the other workers idle)
            n_single = 5
            t0 = time.perf_counter()
            for i in range(n_single):
                pool.submit(_cpu_worker_eval, walkers[i % len(walkers)]).result()
            single = n_single / (time.perf_counter() - t0)
            extra_vals = None
            if extra_theta is not None:     # (untimed: the oracle's chi2 of the `general_walkers` sample)
                extra_vals = list(pool.map(_cpu_worker_eval, [dict(zip(names, row)) for row in extra_theta], chunksize=1))
            total = time.perf_counter() - t_start
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    med = float(np.median(times))
    done = sorted(vals)
    return {'value': workers / med, 'unit': 'evals/s', 'cores': workers, 'host_cpu_count': cores, 'kind': 'port',
            'one_core_value': single,
            'sample': f'{workers} worker processes (1 BLAS thread each) x 1 walker per repeat, median of {repeats} repeats '
                      f'after {warmups} warm-ups ({med * 1e3:.0f} ms per repeat), {len(done)} distinct walkers of the same '
                      f'workload, oracle/vega_cpu.py chi2; {total:.1f} s in all; {workers} of the host\'s {cores} cores: a one-GPU box of the '
                      'pool allows 16 worker processes (more are killed by its process guard)'}, done, [vals[i] for i in done], extra_vals


def set_rank_affinity(pci_address):
    """Pin this rank to the CPUs of its GPU's NUMA node: sysfs `numa_node` of the PCI function HIP reports for the rank's device
    (the order of /sys/class/drm cards is not HIP's device order: on the pool's 8-GPU hosts card0 sits on node 0 while the one
    visible device may be on node 1), then that node's cpulist.  Launch latency through the far socket costs 1 - 2 % of the
    latency-bound figures (single evaluation 47.5 against 46.0 us, Monte-Carlo fits 3400 against 3475 / s); eight Python drivers
    on one socket would otherwise share whatever cores the scheduler picks.  Returns what was done, for the bench line."""
    info = {'cpus_before': len(os.sched_getaffinity(0)), 'pci': pci_address}
    try:
        node = int((Path('/sys/bus/pci/devices') / pci_address / 'numa_node').read_text().strip())
        info['numa_node'] = node
        if node < 0:
            info['set'] = False
            info['reason'] = 'the device reports no NUMA node'
            return info
        cpus = set()
        for part in Path(f'/sys/devices/system/node/node{node}/cpulist').read_text().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if cpus:
            os.sched_setaffinity(0, cpus)
        info['set'] = bool(cpus)
        info['cpus'] = len(os.sched_getaffinity(0))
    except (OSError, ValueError) as exc:
        info['set'] = False
        info['reason'] = str(exc)
    return info


def launch_ranks(args):
    """`--gpus N`, N > 1, and no WORLD_SIZE in the environment: start the N ranks here - fresh child processes of a parent
    that has made NO GPU call (no torch import, the library only compiled, never loaded), one per GPU, the environment
    torchrun would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; the reference's mpirun ranks,
    bin/run_vega_mc_mpi.py:17-25).  Rank 0's stdout carries the one JSON line, which is relayed; the other ranks' stdout and
    everybody's stderr go to this process's stderr.  Returns the exit code: non-zero when any rank failed (the ranks still
    running are then ended - the exact processes started here)."""
    import socket
    import subprocess
    import __graft_entry__ as entry
    entry.compile_library()             # once, here: the ranks find it built
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(0 if args.ranks_share_gpu else r), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # the host driver only supports dmabuf IPC (RCCL needs it)
        env.setdefault('OMP_NUM_THREADS', '1')
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env, cwd=str(REPO),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    # rank 0's stdout is drained by a reader thread while the ranks are polled: the other ranks sit in the final barrier until rank
    # 0 has written its line, and a line larger than the pipe's buffer (64 KB; it is ~15 KB today and grows) must not block it
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = None
    deadline = time.time() + float(os.environ.get('VEGA_BENCH_RANKS_TIMEOUT', 3300))
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = r
        if failed is None and time.time() > deadline:
            sys.stderr.write('bench.py: the ranks did not finish in time\n')
            failed = next(r for r, p in enumerate(procs) if p.poll() is None)
        time.sleep(0.2)
    line = b''
    if failed is None:
        reader.join(timeout=30)
        line = b''.join(chunks)
        failed = next((r for r, p in enumerate(procs) if p.wait() != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        sys.stderr.write(f'bench.py: rank {failed} exited with code {procs[failed].returncode}\n')
        return procs[failed].returncode or 1        # (a rank ended for running too long has a negative code: 1 is returned)
    sys.stdout.write(line.decode())
    sys.stdout.flush()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--workload', default='joint', choices=['joint', 'auto', 'joint_metals', 'joint_metals_fast'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--ramp-steps', type=int, default=150, help='untimed steps before the calibration pass (clock ramp); half as many again before the W warm-up steps')
    ap.add_argument('--event-stride', type=int, default=2, help='HIP-event pairs around every n-th launch of the roofline kernel in the timed region')
    ap.add_argument('--core-only', action='store_true',
                    help='calibration + warm-up + timed steps only (the command the rocprofv3 summaries in profiles/ use)')
    ap.add_argument('--no-static-metals', action='store_true', help='keep every metal pair on its own pipeline')
    ap.add_argument('--force-dist', action='store_true', help='run the collective path even with one rank')
    ap.add_argument('--dist-backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl = RCCL over xGMI (one rank per GPU); gloo: the gather is staged through pinned host memory - '
                         'the rehearsal backend, with --ranks-share-gpu the whole N > 1 path runs on a one-GPU box')
    ap.add_argument('--ranks-share-gpu', action='store_true', help='every rank on GPU 0 (gloo only: RCCL refuses two ranks on one device)')
    ap.add_argument('--lanes', type=int, default=2, choices=[1, 2],
                    help='batches in flight inside the engine (vmx_set_lanes): with 2, consecutive steps alternate between two '
                         'per-batch workspaces that share every static tensor, and overlap on the GPU')
    args = ap.parse_args()
    if args.ranks_share_gpu and args.dist_backend != 'gloo':
        ap.error('--ranks-share-gpu needs --dist-backend gloo (RCCL refuses two ranks on one device)')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(launch_ranks(args))

    # stdout carries the one JSON line and nothing else: whatever libraries print there (RCCL's version banner when a
    # communicator is first used, for one) goes to stderr until the line is written
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = 0 if args.ranks_share_gpu else int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE = {world}: start it as `python bench.py --gpus N` (it '
                         'launches its own ranks) or under torchrun with --nproc-per-node equal to --gpus')

    # torch bundles its own HIP runtime: it has to be loaded before libvegamx.so brings in the system one (importing it
    # does not touch the GPU; the first CUDA call below does, after the CPU baseline has finished)
    import torch
    import torch.distributed as dist

    # every rank makes sure the library exists before any collective is entered: one compiles (file lock), the
    # others wait on the lock - not in a barrier
    import fcntl
    import __graft_entry__ as entry
    with open(REPO / 'vega_amd' / '.build.lock', 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            entry.build()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)

    B = args.batch
    prob = build_problem(args.workload)
    from vega_amd import synthetic
    from vega_amd.engine import Lowering
    low = Lowering(prob)
    host_theta = synthetic.walkers(low.theta0, low.names, B, varied=VARIED, seed=synthetic.SEED + 1000 * rank)

    # the CPU baseline runs first, on rank 0 of the N = 1 run, before this process touches the GPU
    cpu = cpu_idx = cpu_vals = general_vals = None
    general_theta = synthetic.walkers(low.theta0, low.names, B, varied=VARIED + GENERAL_EXTRA, seed=synthetic.SEED + 555)
    if rank == 0 and world == 1 and not args.core_only and not args.no_cpu_baseline:
        cpu, cpu_idx, cpu_vals, general_vals = cpu_baseline(args.workload, low.names, host_theta,
                                                            extra_theta=general_theta[:16])

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the vegamx engine has no CPU fallback')
    torch.cuda.set_device(local_rank)
    # (after the CPU baseline, which uses every core it may: the rank's driver threads next to its GPU)
    if args.ranks_share_gpu:
        affinity = {'set': False, 'cpus_before': len(os.sched_getaffinity(0)), 'reason': 'ranks sharing a GPU'}
    else:
        try:
            props = torch.cuda.get_device_properties(local_rank)
            affinity = set_rank_affinity(f'{props.pci_domain_id:04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0')
        except (AttributeError, RuntimeError) as exc:      # (a torch without the PCI fields: no pinning, said so in the line)
            affinity = {'set': False, 'cpus_before': len(os.sched_getaffinity(0)), 'reason': f'no PCI address of the device: {exc}'}
    use_dist = world > 1 or args.force_dist
    on_rccl = args.dist_backend == 'nccl'
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        if on_rccl:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group('gloo', rank=rank, world_size=world)

    from vega_amd import VegaInterface

    dev = torch.device('cuda', local_rank)
    vegas = []
    for _ in range(1):
        v = VegaInterface(None, problem=prob, max_batch=B, device=local_rank)
        v.freeze_metals()           # fast_metals workloads: the first evaluation (fiducial point) fills the metal caches
        if not args.no_static_metals:
            v.freeze_static_metals()    # polynomial metal pairs -> their exact static Kaiser basis (no-op without metals)
        # the walkers of SURVEY 8d vary biases, betas, alphas, HCD and velocity-dispersion parameters; the Arinyo
        # non-linear parameters are shared by a batch, which the device entry point is told (violations are flagged)
        v.engine.set_constant_nl_hint(True, gaussian=True)
        vegas.append(v)
    vega = vegas[0]
    eng = vega.engine
    engines = [eng]
    L = max(args.lanes, 1)
    eng.set_lanes(L)

    # distinct walker batches per step and per rank, resident in HBM before timing
    assert eng.names == low.names
    n_pool = min(args.steps, 8)
    pools = []
    for i in range(n_pool):
        th = host_theta if i == 0 else synthetic.walkers(eng.low.theta0, eng.names, B, varied=VARIED,
                                                         seed=synthetic.SEED + 1000 * rank + i)
        pools.append(torch.from_numpy(th).to(dev))
    # two output / gather buffer pairs: the collective of step i runs on its own stream while step i + 1 computes
    nslot = 4
    chi2_bufs = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(nslot)]
    # RCCL gathers device buffers; a gloo group (the rehearsal backend) gathers pinned host buffers the chi2 vector is
    # copied into on the communication stream
    gather_dev = dev if on_rccl else torch.device('cpu')
    gathered = [torch.zeros(world * B, dtype=torch.float64, device=gather_dev) for _ in range(nslot)] if use_dist else None
    staged = [torch.zeros(B, dtype=torch.float64).pin_memory() for _ in range(nslot)] if use_dist and not on_rccl else None
    comm_stream = torch.cuda.Stream(device=dev) if use_dist else None
    comm_done = [None] * nslot
    pending = []            # gloo: (slot, event of the device -> host copy) of the steps whose gather has not run yet
    ext_streams = {}

    def stream_of_last_eval():
        h = eng.last_stream_handle()
        if h not in ext_streams:
            ext_streams[h] = torch.cuda.ExternalStream(h, device=dev)
        return ext_streams[h]

    gather_clock = {'on': False, 'events': [], 'host_s': 0.0, 'n': 0}

    def finish_gathers(keep=0):
        """gloo: run the host gathers of all but the last `keep` enqueued steps (the copy's event first)."""
        while len(pending) > keep:
            slot, copied = pending.pop(0)
            copied.synchronize()
            tg = time.perf_counter()
            dist.all_gather_into_tensor(gathered[slot], staged[slot])
            if gather_clock['on']:
                gather_clock['host_s'] += time.perf_counter() - tg
                gather_clock['n'] += 1

    def sync_all():
        for e in engines:
            e.sync()
        if use_dist and not on_rccl:
            finish_gathers()

    def step(i):
        slot = i % nslot
        if use_dist and comm_done[slot] is not None:
            comm_done[slot].synchronize()       # the gather that last read this buffer pair (four steps ago) has finished
        eng.eval_device(pools[i % n_pool].data_ptr(), B, chi2_bufs[slot].data_ptr())
        if use_dist:
            # one all_gather of chi2 per step, ordered after the evaluation by an event on the lane's stream
            comm_stream.wait_event(stream_of_last_eval().record_event())
            with torch.cuda.stream(comm_stream):
                if on_rccl:
                    if gather_clock['on']:
                        ga = torch.cuda.Event(enable_timing=True)
                        ga.record(comm_stream)
                    dist.all_gather_into_tensor(gathered[slot], chi2_bufs[slot])
                    if gather_clock['on']:
                        gb = torch.cuda.Event(enable_timing=True)
                        gb.record(comm_stream)
                        gather_clock['events'].append((ga, gb))
                    comm_done[slot] = comm_stream.record_event()
                else:
                    # the host gather of step i runs once step i + 1 is enqueued: the lanes stay busy while the host waits
                    staged[slot].copy_(chi2_bufs[slot], non_blocking=True)
                    comm_done[slot] = comm_stream.record_event()
                    pending.append((slot, comm_done[slot]))
            if not on_rccl:
                finish_gathers(keep=1)

    # Calibration pass (untimed): HIP-event pairs around every kernel class give the per-kernel breakdown and name
    # the dominant class.  Event pairs around all ~20 launches of a step cost ~7 % of throughput, so the timed
    # region below keeps them around the dominant class only (< 1 %): its launch durations are still measured live,
    # over the timed region, on the streams the kernels run on.
    # Clock ramp (untimed, before anything is measured): the part needs tens of milliseconds of sustained load to reach
    # the clocks it then holds - a 20-step region entered after 5 warm-up steps alone runs ~8 % below the steady state
    # every later region shows (`regions`).  The W warm-up steps of the contract follow further down, right before the
    # timed region.
    for i in range(max(args.warmup, 3) * L + args.ramp_steps):
        step(i)
    sync_all()
    eng.set_profiling(True)         # (every class timed: the engine keeps one batch in flight - uncontended kernel durations)
    for i in range(3):
        step(i)
    eng.sync()
    eng.timings(reset=True)
    for i in range(5):
        step(i)
    eng.sync()
    breakdown = eng.timings(reset=True)
    dominant = max((k for k, v in breakdown.items() if v[1]), key=lambda k: breakdown[k][0])
    # SURVEY.md section 8d prices the roofline on the distortion-matrix step; the class with the largest total is timed too
    # ... which chi2-only steps run as the quadratic-form product (one half-triangle MFMA product per item in place
    # of the distortion and C^-1 products, include/vegamx.h: vmx_set_quadratic_form)
    roof_class = next((k for k in ('quadratic_form_product', 'distortion_product') if breakdown.get(k, (0, 0))[1]), dominant)
    # ... and around every second launch of it: with K = 20 steps on two lanes that is still five live samples of the
    # roofline kernel inside the timed region, at half the perturbation
    eng.set_profiling_classes(sorted({roof_class, dominant}), stride=args.event_stride)

    # (the calibration pass idles the queue between its event pairs: ramp again, until two consecutive 20-step blocks agree
    # to 1 % - the clocks have settled -, at most `ramp_steps` blocks; then the W warm-up steps of the contract)
    previous, settled = 0.0, 0
    for _ in range(max(args.ramp_steps, 1)):
        tb = time.perf_counter()
        for i in range(20):
            step(i)
        sync_all()
        rate = 20 / (time.perf_counter() - tb)
        settled = settled + 1 if abs(rate - previous) < 0.01 * rate else 0
        previous = rate
        if use_dist:
            # (the steps carry a collective: every rank has to run the same number of them - eight blocks, no early exit)
            if _ >= 7:
                break
        elif settled >= 2:
            break
    for i in range(args.warmup):
        step(i)
    sync_all()
    torch.cuda.synchronize()
    eng.timings(reset=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    gather_clock['on'] = use_dist
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    sync_all()
    torch.cuda.synchronize()
    elapsed_own = time.perf_counter() - t0         # this rank's own K steps (the contract's time, below, ends behind the barrier)
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    gather_clock['on'] = False
    timings = eng.timings(reset=True)
    eng.set_profiling(False)

    single_lane = None
    single_timings = None
    if L > 1:
        # (with N > 1 ranks every rank runs this section too: the steps carry the collective, so the ranks stay in step)
        # the same steps with ONE batch in flight (vmx_set_lanes(1)): what a caller gets whose next batch depends on this one's
        # chi2.  With two lanes (`value`) consecutive batches overlap: one lane's launch tails and small kernels are filled
        # by the other's work; co-running kernels share the chip, so per-launch durations (and with them the live roofline
        # fraction) are only clean with one lane - the calibration pass above runs that way.
        eng.set_lanes(1)
        eng.set_profiling(True)
        eng.set_profiling_classes(sorted({roof_class, dominant}))       # event pairs around every launch of the roofline kernel
        for i in range(40):
            step(i)
        sync_all()
        torch.cuda.synchronize()
        eng.timings(reset=True)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        sync_all()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        single_timings = eng.timings(reset=True)
        eng.set_profiling(False)
        single_lane = {'lanes': 1, 'value': B * args.steps / dt, 'unit': 'evals/s', 'ms_per_step': dt / args.steps * 1e3,
                       'note': 'one batch in flight: step i + 1 is enqueued behind step i on the same workspace'}
        eng.set_lanes(L)
        for i in range(40):
            step(i)
        sync_all()

    pk_state = engines[0].debug_read(4, 0, 5)       # live wavenumbers, node-rule tiles and table level of the timed steps
    # the same K steps a few more times: `value` above is ONE region of K steps (the contract); the spread between regions
    # of a few milliseconds each is reported beside it
    regions = None
    if not use_dist:
        rates = []
        for _ in range(5):
            sync_all()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            sync_all()
            torch.cuda.synchronize()
            rates.append(B * args.steps / (time.perf_counter() - t0))
        regions = {'evals_per_s_median': float(np.median(rates)), 'evals_per_s_min': min(rates), 'evals_per_s_max': max(rates),
                   'regions': len(rates), 'steps_per_region': args.steps}

    def rate_of(fn, reps):
        for _ in range(20):
            fn()
        eng.sync()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.sync()
        torch.cuda.synchronize()
        return B * reps / (time.perf_counter() - t0)

    other_paths = None
    if not use_dist and not args.core_only:
        # The other ways through the same engine, same workload, same B (none of them is `value`):
        #   general_walkers  walkers that also differ in Arinyo / smoothing / peak-broadening parameters (no tables)
        #   full_chain       the model is asked for (compute_model_batch): distortion + C^-1 products instead of Q'
        #   host_entry       vmx_eval: host theta in, host chi2 out (one 80 KB copy in, one synchronisation per step)
        other_paths = {}
        reps = max(args.steps // 2, 5)
        d_general = torch.from_numpy(general_theta).to(dev)
        out = torch.zeros(B, dtype=torch.float64, device=dev)
        eng.set_constant_nl_hint(False)
        rate = rate_of(lambda: eng.eval_device(d_general.data_ptr(), B, out.data_ptr()), reps)
        level = int(eng.debug_read(4, 0, 5)[4])
        entry = {'evals_per_s': rate, 'ms_per_step': B / rate * 1e3, 'table_level': level,
                 'varied_in_addition': GENERAL_EXTRA}
        if general_vals is not None:
            got = out[:len(general_vals)].cpu().numpy()
            entry['max_rel_chi2_diff_vs_oracle'] = float(np.max(np.abs(got - general_vals) / np.abs(general_vals)))
            entry['oracle_walkers'] = len(general_vals)
        other_paths['general_walkers'] = entry
        eng.set_constant_nl_hint(True, gaussian=True)
        d_model = torch.zeros(B, eng.model_size, dtype=torch.float64, device=dev)
        rate = rate_of(lambda: eng.eval_device(pools[0].data_ptr(), B, out.data_ptr(), d_model.data_ptr()), reps)
        entry = {'evals_per_s': rate, 'ms_per_step': B / rate * 1e3, 'model_doubles_per_walker': int(eng.model_size)}
        if cpu is not None:
            got = out.cpu().numpy()[cpu_idx]
            entry['max_rel_chi2_diff_vs_oracle'] = float(np.max(np.abs(got - np.array(cpu_vals)) / np.abs(cpu_vals)))
            entry['oracle_walkers'] = len(cpu_idx)
        other_paths['full_chain'] = entry
        host_out = {}

        def host_step():
            host_out['chi2'] = eng.eval(host_theta)[0]
        rate = rate_of(host_step, reps)
        entry = {'evals_per_s': rate, 'ms_per_step': B / rate * 1e3}
        if cpu is not None:
            got = host_out['chi2'][cpu_idx]
            entry['max_rel_chi2_diff_vs_oracle'] = float(np.max(np.abs(got - np.array(cpu_vals)) / np.abs(cpu_vals)))
            entry['oracle_walkers'] = len(cpu_idx)
        other_paths['host_entry'] = entry

    exact_mu = None
    if not use_dist and not args.core_only:
        # the same steps with the reference's 1000-point mu loop itself instead of the node rule that reproduces its sums
        # (include/vegamx.h: vmx_set_mu_quadrature) - reported beside `value`, and the two must agree
        ref_chi2 = chi2_bufs[(args.steps - 1) % nslot].clone()
        if eng.set_mu_quadrature(False) is False:
            for i in range(30):
                step(i)
            sync_all()
            t0 = time.perf_counter()
            for i in range(args.steps):
                step(i)
            sync_all()
            dt = time.perf_counter() - t0
            loop_chi2 = chi2_bufs[(args.steps - 1) % nslot]
            rel = float(((loop_chi2 - ref_chi2).abs() / ref_chi2.abs()).max())
            exact_mu = {'value': B * args.steps / dt, 'unit': 'evals/s', 'ms_per_step': dt / args.steps * 1e3,
                        'max_rel_chi2_diff_vs_node_rule': rel,
                        'note': 'mu sums as the plain 1000-point loop (vmx_set_mu_quadrature(0)); `value` uses the '
                                '178-node rule that reproduces those sums to ~1e-13'}
        eng.set_mu_quadrature(True)

    t = torch.tensor([elapsed], dtype=torch.float64, device=gather_dev if use_dist else dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    communicator = None
    if use_dist:
        # the gathered vector holds every rank's chi2 of the last step, rank-major: this rank's block must be its own
        # vector, and - the ranks draw their walkers from different seeds - the blocks must differ from rank to rank
        torch.cuda.synchronize()
        last = (args.steps - 1) % nslot
        got = gathered[last].to(dev).view(world, B)
        if not torch.equal(got[rank], chi2_bufs[last]) or not bool(torch.isfinite(got).all()):
            raise SystemExit('all_gather of chi2 returned unexpected values')
        distinct = len({tuple(row[:8].tolist()) for row in got.cpu()})
        if distinct != world:
            raise SystemExit(f'all_gather of chi2: {distinct} distinct rank blocks for {world} ranks')
        # what the communicator itself reports
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, (rank, local_rank, torch.cuda.get_device_properties(dev).name))
        # what every rank measured by itself: its own time over the K steps (before the closing barrier), the gather's own time per
        # step (HIP events on the communication stream around the collective; gloo: the host call), the event-timed kernel classes
        # of the timed region, the CPUs it ran on - so that a shortfall at N > 1 can be read off the line: one slow rank, the
        # gather, or the hosts' share
        if on_rccl:
            gather_us = [a.elapsed_time(b) * 1e3 for a, b in gather_clock['events']]
        else:
            gather_us = [gather_clock['host_s'] / max(gather_clock['n'], 1) * 1e6] * max(gather_clock['n'], 1)
        mine = {'rank': rank, 'elapsed_s': elapsed_own, 'gather_us_per_step': float(np.mean(gather_us)) if gather_us else None,
                'gather_us_max': float(np.max(gather_us)) if gather_us else None,
                'kernel_ms_in_timed_region': {k: {'ms': v[0], 'launches': v[1]} for k, v in timings.items() if v[1]},
                'cpu_affinity': affinity}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
        own = sorted(r['elapsed_s'] for r in per_rank)
        slowest = max(per_rank, key=lambda r: r['elapsed_s'])
        gathers = [r['gather_us_per_step'] for r in per_rank if r['gather_us_per_step'] is not None]
        communicator = {'op': 'one all_gather_into_tensor of chi2 [B] per rank and step', 'backend': dist.get_backend(),
                        'rank_elapsed_s': {'min': own[0], 'median': own[len(own) // 2], 'max': own[-1], 'all': [r['elapsed_s'] for r in per_rank],
                                           'note': 'each rank\'s own K steps, before the closing barrier; `value` uses the time behind it'},
                        'gather_us_per_step': {'mean_over_ranks': float(np.mean(gathers)) if gathers else None,
                                               'max_over_ranks': max(gathers) if gathers else None,
                                               'how': 'HIP events on the communication stream around the collective' if on_rccl else 'host clock around the gloo call'},
                        'slowest_rank': {'rank': slowest['rank'], 'elapsed_s': slowest['elapsed_s'],
                                         'kernel_ms_in_timed_region': slowest['kernel_ms_in_timed_region'],
                                         'gather_us_per_step': slowest['gather_us_per_step']},
                        'cpu_affinity': [r['cpu_affinity'] for r in per_rank],
                        'world_size': dist.get_world_size(), 'ranks': [r[0] for r in ranks_seen],
                        'devices': [r[1] for r in ranks_seen], 'device_name': ranks_seen[0][2],
                        'distinct_rank_blocks_in_last_gather': distinct, 'ranks_share_gpu': bool(args.ranks_share_gpu),
                        'bytes_per_rank_per_step': 8 * B}
        if on_rccl:
            try:
                communicator['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
            except Exception as exc:        # pragma: no cover
                communicator['rccl_version'] = f'unavailable ({exc})'
        else:
            communicator['note'] = ('rehearsal backend: chi2 is copied to pinned host memory on the communication stream and '
                                    'gathered there, one step behind the evaluations')
    if rank == 0:
        total_evals = B * args.steps * world
        value = total_evals / elapsed
        kernels = {k: {'ms_per_launch': v[0] / v[1], 'launches_per_step': v[1] // 5, 'ms_per_step': v[0] / 5}
                   for k, v in breakdown.items() if v[1]}
        live = {k: {'ms_per_launch': timings[k][0] / timings[k][1], 'launches': timings[k][1]}
                for k in {roof_class, dominant}}
        n_items = len(prob.items)

        def roofline_for(kclass, ms_per_launch):
            if kclass == 'pk_multipoles':
                # (k, mu) points the stage evaluates per walker and item: the live wavenumbers of the step (blocks whose
                # every value underflows are skipped) x the mu nodes of each - 178 where the node rule applies, 1000 above
                k_live, k_node_max, n_nodes, k_rule, level = pk_state
                points = float(np.where(np.arange(int(k_live)) < k_rule, n_nodes, 1000).sum())
                per_point = FLOPS_PER_POINT_TAB2 if level >= 2 else FLOPS_PER_POINT
                flops = 0.0
                for item in prob.items.values():
                    kind = 'auto' if item.tracer1.name == item.tracer2.name else 'cross'
                    flops += B * points * per_point[kind]          # one paired pass per item
                bound, peak, reach = 'valu-fp64', FP64_VALU_PEAK_TF, FP64_VALU_MEASURED_TF
            elif kclass == 'quadratic_form_product':
                # x'^T Q' x' in half form: nq^2 flops per walker and item, nq = n_model + additive post-distortion coefficients
                nqs = []
                for it in prob.items.values():
                    na = sum(len(range(t.r1[0], t.r1[1] + 1, t.r1[2])) * len(range(t.r2[0], t.r2[1] + 1, t.r2[2]))
                             for t in it.broadband if t.pos == 'post' and t.kind == 'add')
                    nqs.append(it.model_grid.size + na)
                flops = float(sum(n * n for n in nqs)) * B
                if kernels[kclass]['launches_per_step'] > 1:
                    flops /= len(nqs)
                bound, peak, reach = 'mfma', FP64_MFMA_PEAK_TF, FP64_MFMA_4X4X4_MEASURED_TF
            elif kclass in ('distortion_product', 'invcov_product'):
                # (the FFTLog product is left out: its launch skips the operator rows and columns outside the batch's live
                # ranges, so a flop count of the full operator would overstate it)
                if kclass == 'distortion_product':
                    shapes = [(it.dist_grid.size, it.model_grid.size) for it in prob.items.values()]
                elif kclass == 'invcov_product':
                    shapes = [(it.data_size, it.data_size) for it in prob.items.values()]
                else:
                    shapes = [(816, 814)] * 4
                n_vec = B * (4 * n_items // 2 if kclass == 'fftlog_spline_product' else 1)
                flops = sum(2.0 * m * n * n_vec for m, n in shapes)
                if kclass == 'invcov_product':
                    flops /= 2.0            # the half (lower-triangular) form of the symmetric inverse covariance
                # the FFTLog launch covers all ell and, for B > 8, one launch covers every item's product
                if kclass != 'fftlog_spline_product' and kernels[kclass]['launches_per_step'] > 1:
                    flops /= len(shapes)
                bound, peak = 'mfma', FP64_MFMA_PEAK_TF
                reach = FP64_MFMA_4X4X4_MEASURED_TF if kclass == 'distortion_product' else FP64_MFMA_MEASURED_TF
            else:
                return None
            tf = flops / (ms_per_launch * 1e-3) / 1e12
            return {'kernel': kclass, 'bound': bound, 'achieved': tf, 'peak': peak, 'unit': 'TFLOP/s',
                    'frac': tf / peak, 'traffic': None, 'algorithmic_flops_per_launch': flops,
                    'ms_per_launch': ms_per_launch, 'instruction_issue_ceiling': reach,
                    'frac_of_issue_ceiling': tf / reach}

        # The roofline kernel is timed live (HIP events on its stream) in BOTH timed regions.  With two batches in flight a
        # launch shares the chip with the other lane's kernels, so its duration there says what the lanes do to each other, not
        # what the kernel reaches: `roofline` is the one-batch-in-flight region (`single_lane`: the same K steps, same events),
        # `roofline.two_lane_region` the same launch inside the region `value` is measured on.
        live2 = None
        if single_timings is not None and single_timings[roof_class][1]:
            live2 = live
            live = {k: {'ms_per_launch': single_timings[k][0] / single_timings[k][1], 'launches': single_timings[k][1]}
                    for k in {roof_class, dominant}}
        roofline = roofline_for(roof_class, live[roof_class]['ms_per_launch'])
        # HBM bytes per launch come from separate rocprofv3 --pmc passes of `bench.py --core-only` (a counter pass
        # cannot run inside this process): the committed summary of the latest collection is attached when present
        traffic_file = latest_profile('bench_core_traffic.json')
        traffic = json.loads(traffic_file.read_text())['kernels'] if traffic_file is not None else {}
        if roofline is not None and roof_class in traffic and args.workload == 'joint' and B == 256:
            roofline['traffic'] = traffic[roof_class]['hbm_bytes_per_launch']
            roofline['traffic_unit'] = f'bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE, profiles/{traffic_file.name})'
            roofline['algorithmic_bytes_per_launch'] = traffic[roof_class]['algorithmic_bytes_per_launch']
            roofline['traffic_over_algorithmic'] = roofline['traffic'] / roofline['algorithmic_bytes_per_launch']
            # the counter passes belong to the kernel as it was when they were collected: their launch duration travels with
            # them, and a live duration that has moved away from it says the file is due for a refresh
            collected_us = traffic[roof_class].get('avg_us_kernel_trace_run')
            if collected_us:
                roofline['traffic_collected_at_us_per_launch'] = collected_us
                roofline['traffic_stale'] = bool(abs(live[roof_class]['ms_per_launch'] * 1e3 / collected_us - 1.0) > 0.1)
        if roofline is not None:
            roofline['launches_timed'] = live[roof_class]['launches']
            roofline['timing'] = 'HIP events on the launch stream, over the timed region'
            if live2 is not None:
                roofline['timing'] = (f'HIP events on the launch stream over the {args.steps} timed steps of the one-batch-in-flight '
                                      'region (`single_lane`); `two_lane_region`: the same events inside the region `value` '
                                      'is measured on (every second launch of lane 0), where the other lane\'s kernels share the chip')
                r2 = roofline_for(roof_class, live2[roof_class]['ms_per_launch'])
                roofline['two_lane_region'] = {k: r2[k] for k in ('achieved', 'frac', 'ms_per_launch', 'frac_of_issue_ceiling')}
                roofline['two_lane_region']['launches_timed'] = live2[roof_class]['launches']
                roofline['calibration_pass'] = {k: roofline_for(roof_class, kernels[roof_class]['ms_per_launch'])[k]
                                                for k in ('achieved', 'frac', 'ms_per_launch')}
        roofline_other = []
        for k, v in kernels.items():
            if k == roof_class:
                continue
            timed = k in live
            r = roofline_for(k, live[k]['ms_per_launch'] if timed else v['ms_per_launch'])
            if r is not None:
                r['timing'] = 'HIP events over the timed region' if timed else 'calibration pass'
                r['largest_total_time'] = k == dominant
                roofline_other.append(r)
        extras = not args.core_only and world == 1        # the side sections and the CPU baseline belong to the N = 1 run
        distortion = distortion_microbench(eng, torch) if extras else None
        single = single_point_latency(local_rank) if extras else None
        dist_csr = distortion_csr(local_rank) if extras else None
        mc_fits = monte_carlo_fits(prob, local_rank) if extras and args.workload == 'joint' else None
        metals = metals_throughput(local_rank) if extras and args.workload == 'joint' else None
        if other_paths is not None and extras and args.workload == 'joint':
            other_paths['coefmod2'] = coefmod_throughput(local_rank)
        if cpu is not None:
            got = vega.chi2_batch(host_theta[cpu_idx])
            cpu['max_rel_chi2_diff_vs_gpu'] = float(np.max(np.abs(got - np.array(cpu_vals)) / np.abs(cpu_vals)))
        out = {
            'metric': 'model+chi2 evals/sec (Lya auto+cross)', 'value': value, 'unit': 'evals/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': WORKLOAD_TEXT[args.workload].format(B=B, total=B * world),
                       'batch_per_gpu': B, 'batches_in_flight': L, 'pipelines_per_eval': len(eng.pipe_index),
                       'chi2_only': True, 'walkers_share_nl_parameters': True,
                       'full_chain_evals_per_s': ((other_paths or {}).get('full_chain') or {}).get('evals_per_s'),
                       'general_walkers_evals_per_s': ((other_paths or {}).get('general_walkers') or {}).get('evals_per_s'),
                       'single_lane_evals_per_s': (single_lane or {}).get('value'),
                       'fine_print': '`value`: chi2 only (static quadratic form, no model vector written), two independent batches in flight, walkers '
                                     'sharing their Arinyo / smoothing parameters; the same workload with the model written: full_chain_evals_per_s; '
                                     'with walkers that vary those parameters too: general_walkers_evals_per_s; one batch in flight: single_lane_evals_per_s',
                       'varied_parameters': [v for v in VARIED if v in eng.low.slot],
                       'collective': communicator if use_dist else 'none',
                       'cpu_affinity': affinity,
                       'steady_state': f'untimed before the timed region: {args.ramp_steps} ramp steps, a calibration pass, 20-step '
                                       f'blocks until two agree to 1 % (clock ramp), then the {args.warmup} warm-up steps - '
                                       '`value` is a warmed steady state'},
            'roofline': roofline, 'roofline_other_kernels': roofline_other, 'distortion': distortion, 'distortion_csr': dist_csr, 'single_point': single, 'metals': metals, 'monte_carlo_fits': mc_fits, 'single_lane': single_lane, 'exact_mu_loop': exact_mu, 'regions': regions, 'other_paths': other_paths, 'cpu_baseline': cpu,
            'pk_stage': dict(zip(('k_live', 'k_node_max', 'mu_nodes', 'k_on_node_rule', 'table_level'), [float(v) for v in pk_state])),
            'kernels': kernels, 'kernels_note': 'calibration pass before the timed region, event pairs around every kernel',
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    for v in vegas:
        v.close()


if __name__ == '__main__':
    main()
