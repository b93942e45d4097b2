#!/bin/bash
# Profile pass of one bench workload: rocprofv3 kernel stats, FETCH_SIZE / WRITE_SIZE passes (traffic summary for bench.py)
# and, with "sq", the SQ issue-mix passes + the executed-work check.   gpurun -- 'bash tools/gpu_profile.sh qm9 r04P sq'
W=${1:-qm9}; TAG=${2:-r04P}; OUT=$PWD/gpurun_out/$TAG; ROOTD=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-full-round --no-split-leg"
PCMD="python $ROOTD/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-full-round --no-split-leg"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o trace -- $PCMD > $OUT/prof_bench_$W.json 2> $OUT/prof_$W.err )
f=$(find $OUT/prof_$W -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats_$W.csv && head -14 "$f" | cut -c1-150
find $OUT/prof_$W -name "*kernel_trace.csv" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_${c}_$W -o pmc -- $PCMD > $OUT/pmc_${c}_$W.json 2> $OUT/pmc_${c}_$W.err )
  g=$(find $OUT/pmc_${c}_$W -name "*counter_collection.csv" | head -1)
  [ -n "$g" ] && python3 tools/pmc_summary.py "$g" | grep "jd::" | head -12 > $OUT/pmc_${c}_$W.summary.txt && cp "$g" $OUT/pmc_${c}_$W.csv
  find $OUT/pmc_${c}_$W -name "*.csv" -size +8M -delete
done
python3 tools/pmc_traffic.py --fetch $OUT/pmc_FETCH_SIZE_$W.csv --write $OUT/pmc_WRITE_SIZE_$W.csv --stats $OUT/kernel_stats_$W.csv \
   --bench $OUT/prof_bench_$W.json --out $OUT/pmc_traffic_$W.json --command "$CMD"
rm -f $OUT/pmc_FETCH_SIZE_$W.csv $OUT/pmc_WRITE_SIZE_$W.csv
if [ "$3" == "sq" ]; then
  i=1
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/sq${i}_$W -o pmc -- $PCMD > $OUT/sq${i}_$W.json 2> $OUT/sq${i}_$W.err )
    g=$(find $OUT/sq${i}_$W -name "*counter_collection.csv" | head -1)
    [ -n "$g" ] && python3 tools/pmc_summary.py "$g" | grep "jd::" | head -16 > $OUT/pmc_sq${i}_$W.txt
    if [ $i == 1 ] && [ -n "$g" ]; then python3 tools/pmc_work.py --pmc "$g" --bench $OUT/sq1_$W.json --out $OUT/pmc_work_$W.json | tee $OUT/pmc_work_$W.txt; fi
    find $OUT/sq${i}_$W -name "*.csv" -size +8M -delete
    i=$((i+1))
  done
fi
ls $OUT | head -40
