#!/bin/bash
# usage: gpu_sweep.sh TAG "ENV1=a ENV2=b" "ENV1=c" ...   -> one bench --core-only line per setting
tag=$1; shift
mkdir -p gpurun_out
for setting in "$@"; do
  name=$(echo "$setting" | tr ' =' '__')
  env $setting python bench.py --core-only --steps 30 ${BENCH_ARGS} > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err
  python - "$setting" gpurun_out/${tag}_${name}.json <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    k = {n: round(v['ms_per_step'] * 1e3, 1) for n, v in d['kernels'].items()}
    r = d.get('regions') or {}
    print(f"{sys.argv[1]:40s} value {d['value']:9.0f} median {r.get('evals_per_s_median', 0):9.0f}  roof {d['roofline']['frac']:.3f} {d['roofline']['ms_per_launch']*1e3:.1f}us  {k}")
except Exception as exc:
    print(sys.argv[1], 'FAILED', exc)
PY
done
