#!/bin/bash
# counter passes of `python3 bench.py --core-only` (environment assignments as arguments), summarised per kernel:
#   gpurun_out/quad_pmc/summary.json  ->  profiles/r03_quad_pmc.json
for kv in "$@"; do export "$kv"; done
R=$PWD; O=$R/gpurun_out/quad_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -- python3 $R/bench.py --core-only --steps 20 > $O/p1.out 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVES --output-format csv -d $O/p2 -- python3 $R/bench.py --core-only --steps 20 > $O/p2.out 2> $O/p2.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/p3 -- python3 $R/bench.py --core-only --steps 20 > $O/p3.out 2> $O/p3.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/p4 -- python3 $R/bench.py --core-only --steps 20 > $O/p4.out 2> $O/p4.err
cd $R
python3 scripts/gpu_pmc_summary.py $O/summary.json $O/p1 $O/p2 $O/p3 $O/p4
rm -rf $O/p1 $O/p2 $O/p3 $O/p4
