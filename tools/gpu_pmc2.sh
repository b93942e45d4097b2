#!/bin/bash
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/$TAG/pmc2
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
run() { n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o pmc -- \
     python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$n.json 2> $OUT/$n.err
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $OLDPWD/tools/pmc_summary.py "$f" | grep "jd::" | head -6 > $OUT/$n.summary.txt && cat $OUT/$n.summary.txt
  find $OUT/$n -name "*.csv" -size +8M -delete
}
run a SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_IFETCH
run b SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES
