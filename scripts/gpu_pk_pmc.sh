#!/bin/bash
# counter passes on the joint engine at B=256 (P(k,mu) stage); environment assignments as arguments
for kv in "$@"; do export "$kv"; done
R=$PWD; O=$R/gpurun_out/pk_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES --output-format csv -d $O/p1 -- python3 $R/scripts/gpu_pk_only.py > $O/p1.out 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $O/p2 -- python3 $R/scripts/gpu_pk_only.py > $O/p2.out 2> $O/p2.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/p3 -- python3 $R/scripts/gpu_pk_only.py > $O/p3.out 2> $O/p3.err
cd $R
python3 scripts/gpu_pmc_summary.py $O/summary.json $O/p1 $O/p2 $O/p3
rm -rf $O/p1 $O/p2 $O/p3
