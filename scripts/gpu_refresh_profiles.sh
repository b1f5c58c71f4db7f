#!/bin/bash
# Refresh of profiles/ (run through gpurun from the repo root; every pass appends to gpurun_out/refresh/log.txt; then
# `python3 scripts/install_profiles.py rNN` copies the summaries into profiles/ under the round's names):
#   rNN_bench_full.json                      plain `python3 bench.py --steps 20 --warmup 5` (what the driver runs)
#   rNN_bench_core.json + _kernel_stats.csv  `bench.py --core-only --lanes 1` under rocprofv3 --kernel-trace --stats
#   rNN_quad_pmc.json                        SQ counter passes of the same command (MFMA busy / MOPS, VALU, waits)
#   rNN_bench_core_traffic.json              FETCH_SIZE / WRITE_SIZE passes of the same command
#   rNN_joint_metals_kernel_stats.csv (+ traffic, + pmc)   the configs[3] share (joint + metals, B = 512, every pair its own pipeline)
#   rNN_coefmod2_kernel_stats.csv            the COEFMOD = 2 joint fit (factored chi2 form), scripts/gpu_coefmod_core.py
#   rNN_distortion_gemv_kernel_stats.csv, rNN_single_point_chain_kernel_stats.csv   B = 1: the streaming product, the configs[1] chain
R=$PWD
O=$R/gpurun_out/refresh
rm -rf $O && mkdir -p $O
log() { echo "[$(date +%T)] $*" | tee -a $O/log.txt; }
log "full bench"
python3 bench.py --steps 20 --warmup 5 > $O/bench_full.json 2> $O/bench_full.err
log "kernel stats (core, one lane)"
cd /tmp && export TMPDIR=/tmp
CORE="$R/bench.py --core-only --lanes 1 --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $CORE > $O/bench_core.json 2> $O/stats.err
log "SQ counters 1"
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/p1 -- python3 $CORE > $O/p1.out 2> $O/p1.err
log "SQ counters 2"
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVES --output-format csv -d $O/p2 -- python3 $CORE > $O/p2.out 2> $O/p2.err
log "SQ counters 3"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/p3 -- python3 $CORE > $O/p3.out 2> $O/p3.err
log "FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $CORE > $O/fetch.out 2> $O/fetch.err
log "WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $CORE > $O/write.out 2> $O/write.err
cd $R
python3 scripts/gpu_pmc_summary.py $O/quad_pmc.json $O/p1 $O/p2 $O/p3 >> $O/log.txt 2>&1
python3 scripts/gpu_traffic_json.py $O/traffic.json $O/stats $O/fetch $O/write >> $O/log.txt 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | sort | tail -1) $O/kernel_stats.csv
rm -rf $O/stats $O/fetch $O/write $O/p1 $O/p2 $O/p3
log "joint + metals"
cd /tmp
JM="$R/bench.py --core-only --workload joint_metals --batch 512 --lanes 1 --no-static-metals --steps 20 --warmup 5 --ramp-steps 40"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/jm_stats -- python3 $JM > $O/jm_core.json 2> $O/jm.err
log "joint + metals FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/jm_fetch -- python3 $JM > $O/jm_fetch.out 2> $O/jm_fetch.err
log "joint + metals WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/jm_write -- python3 $JM > $O/jm_write.out 2> $O/jm_write.err
cd $R
python3 scripts/gpu_pmc_summary.py $O/jm_traffic_raw.json $O/jm_fetch $O/jm_write >> $O/log.txt 2>&1
cp $(find $O/jm_stats -name '*kernel_stats.csv' | sort | tail -1) $O/jm_kernel_stats.csv
rm -rf $O/jm_stats $O/jm_fetch $O/jm_write
log "joint + metals SQ counters"
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/jm_p1 -- python3 $JM > $O/jm_p1.out 2> $O/jm_p1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_WAVES --output-format csv -d $O/jm_p2 -- python3 $JM > $O/jm_p2.out 2> $O/jm_p2.err
cd $R
python3 scripts/gpu_pmc_summary.py $O/jm_pmc.json $O/jm_p1 $O/jm_p2 >> $O/log.txt 2>&1
rm -rf $O/jm_p1 $O/jm_p2
log "COEFMOD = 2 (factored form)"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/cm_stats -- python3 $R/scripts/gpu_coefmod_core.py > $O/cm.out 2> $O/cm.err
cd $R
cp $(find $O/cm_stats -name '*kernel_stats.csv' | sort | tail -1) $O/cm_kernel_stats.csv
rm -rf $O/cm_stats
log "B = 1 streaming product"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/gemv -- python3 $R/scripts/gpu_matvec_only.py > $O/gemv.out 2> $O/gemv.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/gemv_fetch -- python3 $R/scripts/gpu_matvec_only.py > $O/gemv_fetch.out 2> $O/gemv_fetch.err
log "configs[1] chain"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b1 -- python3 $R/scripts/gpu_b1_trace.py > $O/b1.out 2> $O/b1.err
cd $R
cp $(find $O/gemv -name '*kernel_stats.csv' | sort | tail -1) $O/gemv_kernel_stats.csv
cp $(find $O/b1 -name '*kernel_stats.csv' | sort | tail -1) $O/b1_kernel_stats.csv
python3 scripts/gpu_pmc_summary.py $O/gemv_traffic_raw.json $O/gemv_fetch >> $O/log.txt 2>&1
rm -rf $O/gemv $O/gemv_fetch $O/b1
log done
