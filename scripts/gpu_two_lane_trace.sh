#!/bin/bash
# kernel trace of the two-lane timed region (bench.py --core-only --lanes 2): per-kernel durations and the timeline of the last
# steps -> gpurun_out/lanes2/timeline.txt  (development aid)
R=$PWD
O=$R/gpurun_out/lanes2
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --core-only --lanes ${1:-2} --steps 20 --warmup 5 > $O/core.json 2> $O/err.txt
cd $R
python3 scripts/trace_overlap.py $(find $O/trace -name '*kernel_trace.csv' | head -1) 36 > $O/timeline.txt
rm -rf $O/trace
cat $O/timeline.txt
