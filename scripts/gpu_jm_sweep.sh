#!/bin/bash
# one-lane core timing of the configs[3] share under a few knob settings (development aid) -> gpurun_out/jm_sweep.txt
O=gpurun_out/jm_sweep.txt
: > $O
run() { echo "== $*" >> $O; env "$@" python3 bench.py --core-only --workload joint_metals --batch 512 --lanes 1 --no-static-metals --steps 20 --warmup 5 --ramp-steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d.get('value'), d.get('ms_per_step'))
print({k: round(v,1) for k,v in d.get('stage_us',{}).items()} if 'stage_us' in d else '')
" >> $O; }
run A=1
run VMX_XI_LEAN_NW=1
run VMX_XI_LEAN_NW=4
run VMX_XI_LEAN_NORADPEAK=1
run VMX_XI_LEAN_NORADPEAK=1 VMX_XI_LEAN_NW=4
run VMX_XI_STATIC_NW=2
cat $O
