#!/bin/bash
# SQ counters of a short split-bf16 A/B run (two passes, --kernel-trace + --pmc only): where the split kernels' wave cycles go
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
WL=${1:-qm9}
i=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  rm -rf $R/gpurun_out/split_pmc${i}_$WL
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/split_pmc${i}_$WL -o pmc -- python $R/tools/split_ab.py --workload $WL --steps 3 > $R/gpurun_out/split_pmc${i}_$WL.log 2>&1
  g=$(find $R/gpurun_out/split_pmc${i}_$WL -name "*counter_collection.csv" | head -1)
  [ -n "$g" ] && python3 $R/tools/pmc_summary.py "$g" | grep -E "split|k_node_post<|k_node_ab<|k_edge_update_sym<" | head -12 | tee $R/gpurun_out/split_pmc${i}_$WL.txt
  rm -rf $R/gpurun_out/split_pmc${i}_$WL
  i=$((i+1))
done
