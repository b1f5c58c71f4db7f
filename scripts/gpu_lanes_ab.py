"""Two-batches-in-flight rate of the headline workload under environment settings, interleaved and repeated in one call
(development aid: the rate moves by 1 - 2 % from run to run).  usage: gpu_lanes_ab.py "A=1" "VMX_X=1" ..."""
import json, os, subprocess, sys
settings = sys.argv[1:] or ['A=1']
res = {s: [] for s in settings}
for rep in range(3):
    for s in settings:
        env = dict(os.environ, **dict(kv.split('=', 1) for kv in s.split()))
        out = subprocess.run([sys.executable, 'bench.py', '--core-only', '--lanes', '2', '--steps', '20', '--warmup', '5'],
                             capture_output=True, text=True, env=env).stdout
        d = json.loads(out.strip().splitlines()[-1])
        res[s].append((round(d['value']), round(d['regions']['evals_per_s_median']) if d.get('regions') else 0))
for s, v in res.items():
    print(f'{s:40s}', v)
