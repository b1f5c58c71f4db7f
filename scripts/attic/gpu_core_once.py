"""One-lane and two-lane core timing of the headline workload with the per-kernel breakdown (development aid).
Extra arguments are KEY=VALUE environment settings for the bench processes."""
import json
import os
import subprocess
import sys
env = dict(os.environ, **dict(a.split('=', 1) for a in sys.argv[1:]))
for lanes in (1, 2):
    out = subprocess.run([sys.executable, 'bench.py', '--core-only', '--lanes', str(lanes), '--steps', '20', '--warmup', '5'],
                         capture_output=True, text=True, env=env).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print('lanes', lanes, 'value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4))
    print({k: round(v['ms_per_step'] * 1e3, 1) for k, v in d.get('kernels', {}).items()}, d.get('pk_stage'))
