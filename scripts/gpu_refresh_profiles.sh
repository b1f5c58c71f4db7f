#!/bin/bash
# Refresh the bench-related summaries under profiles/ (run through gpurun from the repo root):
#   full bench line, kernel stats of `bench.py --core-only`, and its two HBM-traffic counter passes.
set -e -o pipefail
R=$PWD
O=$R/gpurun_out/refresh
rm -rf $O && mkdir -p $O
python3 bench.py > $O/bench_full.json 2> $O/bench_full.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --core-only > $O/bench_core.json 2> $O/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py --core-only > $O/fetch.out 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py --core-only > $O/write.out 2> $O/write.err
cd $R
python3 scripts/gpu_traffic_json.py $O/traffic.json $O/stats $O/fetch $O/write
cp $(find $O/stats -name '*kernel_stats.csv' | sort | tail -1) $O/kernel_stats.csv
# the raw traces are large: keep the summaries only
rm -rf $O/stats $O/fetch $O/write
