#!/bin/bash
# One GPU-box round: parity tests, headline bench, rocprofv3 kernel-trace summary and the FETCH_SIZE / WRITE_SIZE PMC
# passes of the same bench command (separate runs, --kernel-trace only), HBM-traffic summary for bench.py.
# Usage (from the build container):  gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r02b [notest]'
TAG=${1:-rXX}
OUT=$PWD/gpurun_out/$TAG
ROOTD=$PWD
mkdir -p $OUT
export TMPDIR=/tmp
if [ "$2" != "notest" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
fi
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err; cat $OUT/bench.json
PCMD="python $ROOTD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-round"
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o trace -- $PCMD > $OUT/prof_bench.json 2> $OUT/prof.err )
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats.csv && head -22 "$f" | cut -c1-170
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o pmc -- $PCMD > $OUT/pmc_$c.json 2> $OUT/pmc_$c.err )
  g=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$g" ] && python3 tools/pmc_summary.py "$g" | grep "jd::" | head -12 > $OUT/pmc_$c.summary.txt
  [ -n "$g" ] && cp "$g" $OUT/pmc_$c.csv
  find $OUT/pmc_$c -name "*.csv" -size +8M -delete
done
python3 tools/pmc_traffic.py --fetch $OUT/pmc_FETCH_SIZE.csv --write $OUT/pmc_WRITE_SIZE.csv --stats $OUT/kernel_stats.csv \
   --bench $OUT/prof_bench.json --out $OUT/pmc_traffic.json --command "python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-round"
ls -la $OUT | head -30
