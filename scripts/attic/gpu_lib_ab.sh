#!/bin/bash
# bench.py --core-only with the default library and with every build_exp/*.so given as argument: value / single lane / kernels
R=$PWD
for lib in default "$@"; do
  if [ "$lib" != default ]; then export VEGAMX_LIBRARY=$R/$lib; fi
  for rep in 1 2; do
    python3 bench.py --core-only --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib', round(d['value']), round(d['single_lane']['value']), {k: round(v['ms_per_step']*1e3,1) for k,v in d['kernels'].items()})"
  done
done
