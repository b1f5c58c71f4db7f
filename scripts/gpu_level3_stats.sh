#!/bin/bash
# per-kernel durations of the bench step with and without the level-3 tables -> gpurun_out/l3_stats/{on,off}.csv (development aid)
R=$PWD
O=$R/gpurun_out/l3_stats
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for mode in on off; do
  if [ $mode = off ]; then export VEGA_AMD_LEVEL3=0; else export VEGA_AMD_LEVEL3=0.125; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$mode -- python3 $R/bench.py --core-only --lanes 1 --steps 20 --warmup 5 > $O/$mode.json 2> $O/$mode.err
  cp $(find $O/$mode -name '*kernel_stats.csv' | head -1) $O/$mode.csv
  rm -rf $O/$mode
  echo "== $mode"; head -12 $O/$mode.csv | cut -d, -f1-4 | cut -c1-110
done
