#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the headline bench.  Usage: gpurun -- 'bash tools/gpu_pmc.sh r01'
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/$TAG/pmc
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -E "^\s*(Name|gpu|counter)|MFMA|VALU|FETCH_SIZE|WRITE_SIZE|GRBM_GUI|WAVE_CYCLES|BUSY_CYCLES|WAIT_INST|LDS_BANK|TCC_HIT|TCC_MISS|TCP_TCC" | head -80 > $OUT/counters_list.txt
run() { # name, counters...
  n=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$n -o pmc -- \
     python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$n.json 2> $OUT/$n.err
  f=$(find $OUT/$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 $OLDPWD/tools/pmc_summary.py "$f" > $OUT/$n.summary.txt && cat $OUT/$n.summary.txt
  find $OUT/$n -name "*.csv" -size +8M -delete
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run fetch FETCH_SIZE
run write WRITE_SIZE
