#!/bin/bash
# kernel statistics of the configs[3] share (joint + metals, B = 512, one lane) -> gpurun_out/jm/kernel_stats.csv
R=$PWD
O=$R/gpurun_out/jm
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --core-only --workload joint_metals --batch 512 --lanes 1 --no-static-metals --steps 20 --warmup 5 --ramp-steps 40 > $O/core.json 2> $O/err.txt
cd $R
cp $(find $O/stats -name '*kernel_stats.csv' | sort | tail -1) $O/kernel_stats.csv
rm -rf $O/stats
cut -d, -f1-4 $O/kernel_stats.csv | cut -c1-150
