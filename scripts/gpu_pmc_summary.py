"""Summarise rocprofv3 output directories (csv): kernel-trace durations and PMC counter means per kernel.
usage: python scripts/gpu_pmc_summary.py <out.json> <dir> [<dir> ...]"""
import glob, json, sys
import pandas as pd

out = {}
for d in sys.argv[2:]:
    for f in glob.glob(f'{d}/**/*_kernel_trace.csv', recursive=True):
        kt = pd.read_csv(f)
        kt['dur_us'] = (kt.End_Timestamp - kt.Start_Timestamp) / 1e3
        for name, g in kt.groupby('Kernel_Name'):
            e = out.setdefault(name, {})
            e.setdefault('runs', []).append({'dir': d, 'calls': int(len(g)), 'avg_us': float(g.dur_us.mean()),
                                             'min_us': float(g.dur_us.min()), 'max_us': float(g.dur_us.max())})
    for f in glob.glob(f'{d}/**/*_counter_collection.csv', recursive=True):
        df = pd.read_csv(f)
        for (name, counter), g in df.groupby(['Kernel_Name', 'Counter_Name']):
            out.setdefault(name, {}).setdefault('counters_mean_per_launch', {})[counter] = float(g.Counter_Value.mean())
json.dump(out, open(sys.argv[1], 'w'), indent=1, sort_keys=True)
print('wrote', sys.argv[1], len(out), 'kernels')
