#!/bin/bash
# bench.py --core-only under the environment given as arguments (e.g. VMX_GEMM_44=1): value and per-kernel table
for kv in "$@"; do export "$kv"; done
python3 bench.py --core-only --steps 40 2> gpurun_out/bk.err | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', round(d['value']), 'ms/step', round(d['ms_per_step'], 4), 'roof', round(d['roofline']['achieved'], 2), round(d['roofline']['ms_per_launch'], 4))
for k, v in d['kernels'].items():
    print(f\"  {k:24s} {v['ms_per_launch']:.4f}\")
"
