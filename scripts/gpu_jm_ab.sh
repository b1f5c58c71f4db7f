#!/bin/bash
# joint + metals, B = 512, one lane: evaluations / s under a few knob settings (development aid) -> gpurun_out/jm_ab.txt
O=gpurun_out/jm_ab.txt
: > $O
run() { echo "== $*" >> $O; env "$@" python3 bench.py --core-only --workload joint_metals --batch 512 --lanes 1 --no-static-metals --steps 20 --warmup 5 --ramp-steps 40 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(d['value']), round(d['ms_per_step'],4), {k: round(v['ms_per_step']*1e3,1) for k,v in d.get('kernels',{}).items()})
" >> $O; }
run A=1
run VMX_QUAD_SKEW=0
run VMX_QUAD_SKEW=0.2
run A=1
run VMX_QUAD_SKEW=0
cat $O
