import os, sys, subprocess, json
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1:
    sys.path.insert(0, str(REPO))
    import torch
    torch.cuda.init()
    import bench
    print(json.dumps(bench.single_point_latency(0)))
    sys.exit(0)
for label, env in (('default', {}), ('no_host_reduce', {'VMX_NO_HOST_REDUCE': '1'}), ('no_small_tab', {'VMX_NO_SMALL_TAB': '1'})):
    r = subprocess.run([sys.executable, __file__, 'x'], env={**os.environ, **env}, capture_output=True, text=True)
    print(label, r.stdout.strip().split('\n')[-1] if r.stdout else r.stderr[-1500:])
