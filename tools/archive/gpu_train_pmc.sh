#!/bin/bash
# SQ counter passes over forward + backward of the training path:  gpurun -- 'JODO_PROF_BATCH=2048 bash tools/gpu_train_pmc.sh tag "k_chain_a|k_bwd_a|k_chain_c"'
TAG=${1:-train}; PAT=${2:-"k_chain|k_bwd_"}; OUT=$PWD/gpurun_out/train_pmc_$TAG; ROOTD=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
i=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM"; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -o pmc -- python $ROOTD/tools/train_fwd_prof.py 1 1 bwd > $OUT/sq$i.log 2>&1 )
  g=$(find $OUT/sq$i -name "*counter_collection.csv" | head -1)
  [ -n "$g" ] && python3 tools/pmc_summary.py "$g" | grep -E "$PAT" | head -12 | tee -a $OUT/summary.txt
  rm -rf $OUT/sq$i
  i=$((i+1))
done
