#!/bin/bash
# SQ counter passes of one training-GEMM shape:  gpurun -- 'bash tools/gpu_gemm_pmc.sh "0 1 41096 256 256" tag'
SHAPE=${1:-"0 1 41096 256 256"}; TAG=${2:-gemm}; OUT=$PWD/gpurun_out/gemm_pmc_$TAG; ROOTD=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
i=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_CYCLES_VMEM"; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -o pmc -- python $ROOTD/tools/gemm_one.py $SHAPE 10 > $OUT/sq$i.log 2>&1 )
  g=$(find $OUT/sq$i -name "*counter_collection.csv" | head -1)
  [ -n "$g" ] && python3 tools/pmc_summary.py "$g" | grep "gemm" | head -3
  rm -rf $OUT/sq$i
  i=$((i+1))
done
