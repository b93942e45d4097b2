#!/bin/bash
# cache counter passes of a bench workload (L2 hit rate of the weight streams, L1 traffic), --kernel-trace only:
#   gpurun --timeout 900 -- 'bash tools/gpu_cache.sh r03C qm9'
OUT=$PWD/gpurun_out/${1:-r03C}; W=${2:-qm9}; ROOTD=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
PCMD="python $ROOTD/bench.py --workload $W --steps 4 --warmup 2 --no-cpu-baseline --no-full-round"
i=1
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU"; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/c$i -o pmc -- $PCMD > $OUT/c$i.json 2> $OUT/c$i.err )
  g=$(find $OUT/c$i -name "*counter_collection.csv" | head -1)
  [ -n "$g" ] && python3 tools/pmc_summary.py "$g" | grep "jd::\|k_node_mix" | head -8 > $OUT/pmc_cache${i}_$W.txt
  find $OUT/c$i -name "*.csv" -size +8M -delete
  head -8 $OUT/pmc_cache${i}_$W.txt | cut -c1-300
  tail -2 $OUT/c$i.err | cut -c1-200
  i=$((i+1))
done
