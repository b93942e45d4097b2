#!/bin/bash
# One GPU-box round: parity tests, headline bench, rocprofv3 kernel-trace summary of the same bench.
# Usage (from the build container):  gpurun --timeout 2400 -- 'bash tools/gpu_round.sh r01'
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 900 python bench.py --steps 10 --warmup 3 --breakdown > $OUT/bench.json 2> $OUT/bench.err
tail -2 $OUT/bench.err; cat $OUT/bench.json
# kernel-trace profile of the same command (fewer steps; no CPU baseline)
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o trace -- \
    python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OLDPWD/$OUT/prof_bench.json 2> $OLDPWD/$OUT/prof.err )
find $OUT/prof -name "*kernel_stats*" | head -3
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-220 > $OUT/kernel_stats_top.csv && cat $OUT/kernel_stats_top.csv | cut -c1-160
# keep the merged output small
find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete
