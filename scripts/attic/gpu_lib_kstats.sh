#!/bin/bash
# rocprofv3 kernel-trace statistics of `bench.py --core-only --lanes 1` with the default library and with every build_exp/*.so
# given as argument: the average duration of the kernels whose name matches $KPAT (default: k_prologue|k_xtab|k_chi2_parts)
R=$PWD
KPAT=${KPAT:-k_prologue|k_xtab|k_chi2_parts}
export TMPDIR=/tmp
for lib in default "$@"; do
  if [ "$lib" != default ]; then export VEGAMX_LIBRARY=$R/$lib; fi
  out=$R/gpurun_out/kstats_$(basename $lib .so)
  rm -rf $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --core-only --lanes 1 --steps 20 --warmup 5 > /dev/null 2>&1
  f=$(find $out -name '*kernel_stats.csv' | head -1)
  echo "== $lib"
  python3 - "$f" "$KPAT" <<'PY'
import csv, re, sys
for row in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], row['Name']):
        print(f"  {row['Name'][:60]:60s} calls {row['Calls']:>6s} avg {float(row['AverageNs'])/1e3:8.2f} us  min {float(row['MinNs'])/1e3:7.2f}")
PY
done
