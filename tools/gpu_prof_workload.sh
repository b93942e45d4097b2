#!/bin/bash
# rocprofv3 kernel-trace summary of one bench workload:  gpurun -- 'bash tools/gpu_prof_workload.sh cond r02P'
W=${1:-cond}; OUT=$PWD/gpurun_out/${2:-r02P}; ROOTD=$PWD; mkdir -p $OUT; export TMPDIR=/tmp
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$W -o trace -- python $ROOTD/bench.py --workload $W --steps 5 --warmup 2 --no-cpu-baseline --no-full-round > $OUT/prof_bench_$W.json 2> $OUT/prof_$W.err )
f=$(find $OUT/prof_$W -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" $OUT/kernel_stats_$W.csv && head -40 "$f" | cut -c1-150
find $OUT/prof_$W -name "*kernel_trace.csv" -size +20M -delete
