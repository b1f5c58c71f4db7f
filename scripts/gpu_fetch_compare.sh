#!/bin/bash
# FETCH_SIZE of the quadratic-form product under the settings given as arguments (each "ENV=val ENV=val")
R=$PWD; O=$R/gpurun_out/fetchcmp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for setting in "$@"; do
  i=$((i+1))
  env $setting rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/p$i -- python3 $R/bench.py --core-only --lanes 1 --steps 10 --warmup 3 --ramp-steps 30 ${BENCH_ARGS} > /dev/null 2> $O/p$i.err
  python3 - "$setting" $O/p$i <<'PY'
import glob, sys
import pandas as pd
d = sys.argv[2]
c = pd.read_csv(glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0])
k = pd.read_csv(glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0])
k['us'] = (k.End_Timestamp - k.Start_Timestamp) / 1e3
q = c[c.Kernel_Name.str.contains('k_gemm_nt44<12', regex=False) & (c.Counter_Name == 'FETCH_SIZE')]
kq = k[k.Kernel_Name.str.contains('k_gemm_nt44<12', regex=False)]
print(f"{sys.argv[1]:45s} FETCH {q.Counter_Value.mean() * 2048 / 1e6:8.1f} MB  avg {kq.us.mean():7.1f} us  launches {len(kq)}")
PY
  rm -rf $O/p$i
done
