#!/bin/bash
# Timeline of the 1024-mock Monte-Carlo fit (BASELINE configs[4] share): the host-side view (engine calls, batch sizes, gaps) and
# the kernel trace's view (GPU-busy fraction of the run) -> gpurun_out/mc_timeline/{host.json,trace.json}.  Extra arguments go to
# scripts/gpu_mc_timeline.py (e.g. --python-driver).
R=$PWD
O=$R/gpurun_out/mc_timeline
rm -rf $O && mkdir -p $O
python3 scripts/gpu_mc_timeline.py --out $O/host.json "$@" > $O/host.out 2> $O/host.err || { tail -20 $O/host.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/scripts/gpu_mc_timeline.py --marker "$@" > $O/traced.out 2> $O/traced.err || { tail -20 $O/traced.err; exit 1; }
cd $R
python3 scripts/gpu_mc_timeline.py --trace $(find $O/trace -name '*kernel_trace.csv' | head -1) --out $O/trace.json > /dev/null
rm -rf $O/trace
cat $O/host.json $O/trace.json
