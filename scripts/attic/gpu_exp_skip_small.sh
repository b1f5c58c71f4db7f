set -e
mkdir -p gpurun_out/skip
for s in ${SKIPS:-0 1 2 3}; do
  VEGAMX_LIBRARY=vega_amd/libvegamx_exp.so VMX_EXP_SKIP=$s timeout -k 10 200 python3 bench.py --core-only --steps 20 --warmup 5 > gpurun_out/skip/s$s.json 2> gpurun_out/skip/s$s.err
  python3 - <<PY
import json
d=json.loads(open('gpurun_out/skip/s$s.json').read().strip().splitlines()[-1])
sl=d.get('single_lane') or {}
print('skip=$s value', round(d['value']), 'ms/step', d['ms_per_step'], 'single_lane', sl.get('evals_per_s'), sl.get('ms_per_step'), 'regions', d.get('regions'))
PY
done
