#!/bin/bash
# counter passes on the stand-alone product (environment assignments as arguments, e.g. VMX_GEMM_44=1)
for kv in "$@"; do export "$kv"; done
R=$PWD; O=$R/gpurun_out/gemm_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -- python3 $R/scripts/gpu_gemm_one.py > $O/p1.out 2> $O/p1.err
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAVES --output-format csv -d $O/p2 -- python3 $R/scripts/gpu_gemm_one.py > $O/p2.out 2> $O/p2.err
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/p3 -- python3 $R/scripts/gpu_gemm_one.py > $O/p3.out 2> $O/p3.err
cd $R
python3 scripts/gpu_pmc_summary.py $O/summary.json $O/p1 $O/p2 $O/p3
rm -rf $O/p1 $O/p2 $O/p3
