"""prints the fields of a bench line one looks at first (python3 scripts/attic/gpu_bench_fields.py <file with the JSON line>)"""
import json
import sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['frac'], d['roofline'].get('ms_per_launch'))
print('config', json.dumps(d['config'])[:1500])
for key in ('other_paths', 'monte_carlo_fits', 'single_point', 'single_lane', 'metals', 'cpu_baseline'):
    print(key, json.dumps(d.get(key))[:1800])
