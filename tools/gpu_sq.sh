#!/bin/bash
# SQ counter passes of the default bench command (issue mix, MFMA pipe busy, waits, LDS), --kernel-trace only:
#   gpurun --timeout 900 -- 'bash tools/gpu_sq.sh r02Q'
OUT=$PWD/gpurun_out/${1:-r02Q}; ROOTD=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
PCMD="python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-round"
i=1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq$i -o pmc -- $PCMD > $OUT/sq$i.json 2> $OUT/sq$i.err )
  g=$(find $OUT/sq$i -name "*counter_collection.csv" | head -1)
  [ -n "$g" ] && python3 tools/pmc_summary.py "$g" | grep "jd::" | head -14 > $OUT/pmc_sq$i.txt
  find $OUT/sq$i -name "*.csv" -size +8M -delete
  head -3 $OUT/pmc_sq$i.txt | cut -c1-330
  i=$((i+1))
done
