"""Timeline of the 1024-mock Monte-Carlo fit of the bench (BASELINE configs[4] share): engine calls, their batch sizes, the
host gaps between them and - under `rocprofv3 --kernel-trace` - the GPU-busy fraction of the run.

    python3 scripts/gpu_mc_timeline.py [--mocks 1024] [--method migrad] [--python-driver] [--out FILE]
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scripts/gpu_mc_timeline.py --marker
    python3 scripts/gpu_mc_timeline.py --trace DIR/.../*_kernel_trace.csv        (summarise the trace: no GPU needed)

The host-side part wraps the engine's entry points with time stamps; `--marker` makes the script sleep 0.4 s before and after
the timed run so that the trace can be cut at the silent gaps.
"""
import argparse
import csv
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def summarise_trace(path, gap_s=0.3):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    # segments separated by silent gaps of more than gap_s: the timed run is the last segment with more than 1000 kernels
    segs, cur = [], [rows[0]]
    for a, b in zip(rows[:-1], rows[1:]):
        if b[0] - max(a[1], cur[-1][1]) > gap_s * 1e9:
            segs.append(cur)
            cur = []
        cur.append(b)
    segs.append(cur)
    big = [s for s in segs if len(s) > 1000]
    seg = big[-1]
    t0, t1 = seg[0][0], max(r[1] for r in seg)
    # union of the kernel intervals
    busy, end = 0, t0
    for s, e, _ in seg:
        if e <= end:
            continue
        busy += e - max(s, end)
        end = e
    by = {}
    for s, e, name in seg:
        key = name.split('(')[0][:48]
        d = by.setdefault(key, [0, 0.])
        d[0] += 1
        d[1] += (e - s) / 1e3
    out = {'kernels': len(seg), 'span_ms': (t1 - t0) / 1e6, 'gpu_busy_ms': busy / 1e6, 'gpu_busy_fraction': busy / (t1 - t0),
           'by_kernel_us': {k: {'launches': v[0], 'total_us': round(v[1], 1)} for k, v in
                            sorted(by.items(), key=lambda kv: -kv[1][1])[:14]}}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mocks', type=int, default=1024)
    ap.add_argument('--method', default='migrad')
    ap.add_argument('--python-driver', action='store_true', help='the NumPy lock-step driver (vega_amd/migrad.py) instead of the device-resident one')
    ap.add_argument('--marker', action='store_true')
    ap.add_argument('--like-bench', type=int, default=0, help='1: a second engine with two lanes alive (as in bench.py), 2: the same, closed again')
    ap.add_argument('--trace', default=None)
    ap.add_argument('--out', default=None)
    args = ap.parse_args()
    if args.trace:
        out = summarise_trace(args.trace)
        text = json.dumps(out, indent=1)
        print(text)
        if args.out:
            Path(args.out).write_text(text + '\n')
        return

    import bench
    from vega_amd import VegaInterface
    prob = bench.build_problem('joint')
    if args.like_bench:
        # what bench.py has done before its Monte-Carlo leg: torch on the device, a second engine with two lanes that has run steps
        import torch
        from vega_amd import synthetic
        other = VegaInterface(None, problem=prob, max_batch=256, device=0)
        other.engine.set_constant_nl_hint(True, gaussian=True)
        other.engine.set_lanes(2)
        pool = torch.from_numpy(synthetic.walkers(other.engine.low.theta0, other.engine.names, 256, varied=bench.VARIED)).to('cuda:0')
        out = torch.zeros(256, dtype=torch.float64, device='cuda:0')
        for _ in range(300):
            other.engine.eval_device(pool.data_ptr(), 256, out.data_ptr())
        other.engine.sync()
        if args.like_bench > 1:
            other.close()
    vega = VegaInterface(None, problem=prob, max_batch=4096, device=0)
    vega.chi2()
    names = ['ap', 'at', 'bias_eta_LYA', 'beta_LYA', 'beta_QSO', 'bias_hcd']
    limits = {'ap': (0.5, 1.5), 'at': (0.5, 1.5), 'bias_eta_LYA': (-2., 0.), 'beta_LYA': (0., 5.),
              'beta_QSO': (0., 1.), 'bias_hcd': (-0.5, 0.)}
    errors = {'ap': 0.01, 'at': 0.01, 'bias_eta_LYA': 0.01, 'beta_LYA': 0.1, 'beta_QSO': 0.1, 'bias_hcd': 0.01}
    sample = {'limits': limits, 'values': {n: vega.params[n] for n in names}, 'errors': errors,
              'fix': {n: False for n in names}}
    kw = {}
    if args.python_driver:
        import os
        os.environ['VEGA_AMD_FIT_DRIVER'] = 'python'
    vega.run_monte_carlo(num_mocks=128, seed=5, sample_params=sample, method=args.method, **kw)

    eng = vega.engine
    calls = []
    orig_eval = eng.eval

    def eval_logged(theta, want_model=False):
        t0 = time.perf_counter()
        r = orig_eval(theta, want_model)
        calls.append((t0, time.perf_counter(), np.asarray(theta).shape[0] if np.asarray(theta).ndim > 1 else 1))
        return r
    eng.eval = eval_logged
    if args.marker:
        time.sleep(0.4)
    t0 = time.perf_counter()
    res = vega.run_monte_carlo(num_mocks=args.mocks, seed=11, sample_params=sample, method=args.method, **kw)
    t1 = time.perf_counter()
    if args.marker:
        time.sleep(0.4)
    eng.eval = orig_eval
    out = {'mocks': args.mocks, 'method': args.method, 'seconds': t1 - t0, 'fits_per_s': args.mocks / (t1 - t0),
           'chi2_evaluations': int(res.nfcn.sum()), 'evals_per_fit': float(res.nfcn.mean())}
    stats = getattr(res, 'driver_stats', None)
    if stats:
        out['device_driver'] = stats
    if len(calls) > 2:
        ts = np.array([(a, b) for a, b, _ in calls])
        B = np.array([c for _, _, c in calls])
        in_engine = float((ts[:, 1] - ts[:, 0]).sum())
        gaps = ts[1:, 0] - ts[:-1, 1]
        edges = [1, 2, 9, 65, 257, 1025, 4097]
        hist = {f'{lo}..{hi - 1}': int(((B >= lo) & (B < hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])}
        evals_hist = {f'{lo}..{hi - 1}': int(B[(B >= lo) & (B < hi)].sum()) for lo, hi in zip(edges[:-1], edges[1:])}
        out['host_entry_calls'] = {
            'calls': len(calls), 'seconds_inside_engine_calls': in_engine, 'seconds_between_calls': float(gaps.sum()),
            'seconds_before_first_and_after_last': float((ts[0, 0] - t0) + (t1 - ts[-1, 1])),
            'host_fraction': 1. - in_engine / (t1 - t0),
            'gap_us_median': float(np.median(gaps) * 1e6), 'gap_us_p90': float(np.percentile(gaps, 90) * 1e6),
            'calls_by_batch_size': hist, 'evaluations_by_batch_size': evals_hist,
            'time_in_calls_by_batch_size_ms': {f'{lo}..{hi - 1}': float(((ts[:, 1] - ts[:, 0])[(B >= lo) & (B < hi)]).sum() * 1e3)
                                               for lo, hi in zip(edges[:-1], edges[1:])}}
    text = json.dumps(out, indent=1)
    print(text)
    if args.out:
        Path(args.out).write_text(text + '\n')
    vega.close()


if __name__ == '__main__':
    main()
